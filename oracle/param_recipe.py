"""Deterministic per-key parameter recipe (TEST INFRASTRUCTURE).

Golden fixtures do not store model weights (the Cornell-size RGCN alone is tens of MB).  Instead
every tensor of a state dict is regenerated from ``(seed, key, shape)`` with numpy's PCG64, which is
bit-reproducible across machines, so the fixture generator (which loads the values into the real
``main.GraphTextLM``) and the tests (which load them into the oracle and into ``gmlm_amd``) see
identical weights.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def recipe_tensor(key: str, shape, seed: int) -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
    shape = tuple(shape)
    z = rng.standard_normal(shape).astype(np.float32)
    leaf = key.split(".")[-1]
    if key.endswith("scale_weights"):
        v = 0.25 + 0.2 * z
    elif len(shape) == 1:
        norm_like = ("LayerNorm" in key or "layer_norm" in key or "gnorm" in key or key.startswith("fusion_network.1."))
        if leaf in ("weight", "mean_scale") and norm_like:
            v = 1.0 + 0.1 * z
        else:
            v = 0.05 * z
    elif "embeddings" in key and len(shape) == 2:
        v = 0.3 * z
    elif key.endswith(".comp"):
        v = z / np.sqrt(shape[-1])
    elif len(shape) == 3 or key.endswith(".root"):      # rgcn weight [B,in,out] / root [in,out]
        v = z / np.sqrt(shape[-2])
    elif key == "gnn_mask_token_embed":
        v = 0.5 * z
    else:                                               # nn.Linear [out,in]
        v = z / np.sqrt(shape[-1])
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def recipe_state_dict(template: dict, seed: int) -> dict:
    """template: {key: tensor-or-shape}.  Non-float entries (buffers) are passed through."""
    out = {}
    for k, v in template.items():
        if torch.is_tensor(v) and not v.is_floating_point():
            out[k] = v.clone()
            continue
        shape = v.shape if torch.is_tensor(v) else v
        out[k] = recipe_tensor(k, shape, seed)
    return out
