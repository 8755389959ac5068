"""ctypes binding of libgmlm_hip.so (C ABI: include/gmlm_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a Python
exception is raised.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C gmlm_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgmlm_hip.so")

F32, BF16 = 0, 1
_p, _i64, _i32, _f32, _u64, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); mirrors include/gmlm_hip.h line by line
SIGNATURES = {
    "gmlm_version": (C.c_int, []),
    "gmlm_last_error": (C.c_char_p, []),
    "gmlm_device_check": (C.c_int, [_p, _p, C.c_char_p, _i32]),
    "gmlm_degree_i32": (C.c_int, [_p, _i64, _i64, _p, _p]),
    "gmlm_degree_f32": (C.c_int, [_p, _i64, _i64, _p, _p]),
    "gmlm_edge_bucket": (C.c_int, [_p, _p, _i64, _i64, _p, _p]),
    "gmlm_relation_histogram": (C.c_int, [_p, _i64, _i32, _p, _p]),
    "gmlm_segment_sort_workspace_bytes": (_sz, [_i64]),
    "gmlm_segment_sort": (C.c_int, [_p, _p, _p, _i32, _i64, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "gmlm_gather_i64_to_i32": (C.c_int, [_p, _p, _i64, _p, _p]),
    "gmlm_gather_i32": (C.c_int, [_p, _p, _i64, _p, _p]),
    "gmlm_segment_inv_count": (C.c_int, [_p, _i64, _p, _p]),
    "gmlm_split_plan_capacity": (C.c_int, [_i64, _i64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gmlm_split_plan_build": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p]),
    "gmlm_rgcn_mean_spmm": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _i32, _i64, _i64, _p, _i64, _i32, _i64, _p, _p, _p, _i64, _i64,
                                      _p, _p]),
    "gmlm_basis_compose_fwd": (C.c_int, [_p, _p, _i32, _i32, _i64, _p, _p]),
    "gmlm_basis_compose_bwd_workspace_bytes": (_sz, [_i32, _i32, _i64]),
    "gmlm_basis_compose_bwd": (C.c_int, [_p, _p, _p, _i32, _i32, _i64, _p, _p, _p, _sz, _p]),
    "gmlm_colstats_workspace_bytes": (_sz, [_i64, _i64]),
    "gmlm_colstats": (C.c_int, [_p, _i32, _p, _i64, _i64, _p, _p, _p, _sz, _p]),
    "gmlm_graphnorm_finalize": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _f32, _p, _p, _p]),
    "gmlm_graphnorm_apply": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _f32, _u64, _p, _p, _i32, _p]),
    "gmlm_graphnorm_bwd_stats": (C.c_int, [_p, _i32, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _f32, _u64, _p, _p, _p, _sz, _p]),
    "gmlm_graphnorm_bwd_apply": (C.c_int, [_p, _i32, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _f32, _u64, _p,
                                           _p, _p, _p, _p, _p]),
    "gmlm_bias_res_layernorm_fwd": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _f32, _i32, _f32, _u64, _p, _p, _p, _p, _i32, _p]),
    "gmlm_layernorm_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "gmlm_bias_res_layernorm_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _f32, _u64, _p, _p, _p, _p, _p,
                                              _p, _i32, _p, _sz, _p]),
    "gmlm_attention_fwd": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _u64, _p,
                                     _p, _p, _p, _i32, _p, _i64, _p, _i64, _p]),
    "gmlm_attention_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "gmlm_attention_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32,
                                     _f32, _u64, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _p, _i64, _p, _sz, _p, _p, _p, _p, _i64, _p]),
    "gmlm_embed_sum_fwd": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _i32, _p, _p]),
    "gmlm_meanpool_scatter_fwd": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _i32, _p, _p]),
    "gmlm_meanpool_scatter_bwd": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _i32, _p, _p]),
    "gmlm_softmask_blend_fwd": (C.c_int, [_p, _p, _p, _f32, _i64, _i64, _p, _i64, _i32, _p]),
    "gmlm_softmask_blend_bwd": (C.c_int, [_p, _i64, _p, _f32, _i64, _i64, _p, _p, _sz, _p]),
    "gmlm_bias_gelu_fwd": (C.c_int, [_p, _p, _i64, _i64, _f32, _u64, _p, _p, _i32, _p]),
    "gmlm_bias_gelu_bwd_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "gmlm_bias_gelu_bwd": (C.c_int, [_p, _p, _p, _i64, _i64, _f32, _u64, _p, _p, _p, _i32, _p, _sz, _p]),
}


class GmlmHipError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load libgmlm_hip.so once; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GmlmHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C gmlm_amd/csrc` "
            "(or __graft_entry__.build()). gmlm_amd has no CPU / eager fallback.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError here = header / library mismatch
        fn.restype, fn.argtypes = res, args
    if handle.gmlm_version() != 1:
        raise GmlmHipError(f"ABI version mismatch: library {handle.gmlm_version()} != binding 1")
    _lib = handle
    return handle


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().gmlm_last_error().decode(errors="replace")
        raise GmlmHipError(f"{what or 'gmlm call'} failed (code {rc}): {msg}")


_device_ok = {}


def require_gfx950(device_index: int) -> None:
    """One-time check per device that we are on a gfx950 part."""
    if _device_ok.get(device_index):
        return
    import torch
    with torch.cuda.device(device_index):
        arch = C.create_string_buffer(64)
        cu, wave = C.c_int(0), C.c_int(0)
        check(lib().gmlm_device_check(C.byref(cu), C.byref(wave), arch, 64), "gmlm_device_check")
    _device_ok[device_index] = (cu.value, wave.value, arch.value.decode())


def device_info(device_index: int = 0):
    require_gfx950(device_index)
    return _device_ok[device_index]
