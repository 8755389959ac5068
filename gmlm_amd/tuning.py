"""Library-GEMM algorithm selection for the dense projections (hipBLASLt / rocBLAS through PyTorch's TunableOp).

60 % of the headline step is plain library GEMMs (BERT projections at T ~ 9e4 tokens, K = 768 / 3072; RGCN H W_cat): the
library's default heuristic is within 0-15 % of its own best kernel per shape.  ``enable_gemm_tuning`` switches TunableOp on:
* ``tune=False`` (default): only LOOK UP a results file (``gmlm_amd/tunable/gfx950.csv`` ships the picks for the bench
  workloads; the file carries validators for the PyTorch / HIP / hipBLASLt / rocBLAS versions and the GPU arch, and is
  ignored — default algorithms — when they do not match);
* ``tune=True``: time every candidate the first time a shape is seen and write the results file (tens of seconds for a
  BERT-base step; done once, offline).
The numerics are those of the library kernels either way (bf16 operands, fp32 accumulation)."""
from __future__ import annotations

import os
from typing import Optional

import torch

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunable", "gfx950.csv")


def enable_gemm_tuning(path: Optional[str] = None, tune: bool = False, max_tuning_ms: int = 200,
                       rotating_buffer_mb: Optional[int] = None) -> bool:
    """-> True when TunableOp is active with ``path`` (a results file whose validators match this software stack, or
    tuning requested); False = the library's default algorithms are in use."""
    path = path or DEFAULT_FILE
    if not tune and not os.path.exists(path):
        return False
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(bool(tune))
    if tune:
        t.set_filename(path, insert_device_ordinal=False)
        t.set_max_tuning_duration(int(max_tuning_ms))              # the results file is written by TunableOp as shapes are tuned / at exit
        if rotating_buffer_mb is not None:                          # operands rotate through a buffer larger than the caches: candidates are timed cold, as in a real step
            t.set_rotating_buffer_size(int(rotating_buffer_mb))
    else:
        if not t.read_file(path):                                   # validators (PyTorch / HIP / hipBLASLt / rocBLAS versions, GPU arch) rejected the file
            disable_gemm_tuning()
            return False
        # look-up only: whatever TunableOp may still want to write goes to a scratch file of this process, never to the
        # shipped picks (several ranks share them)
        import tempfile
        t.set_filename(os.path.join(tempfile.gettempdir(), "gmlm_tunable_%d.csv" % os.getpid()), insert_device_ordinal=False)
    return True


def disable_gemm_tuning() -> None:
    try:
        torch.cuda.tunable.enable(False)
    except Exception:      # nothing to switch off
        pass
