"""Shared test helpers: golden loading, oracle model construction from the parameter recipe."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    for k in ("config", "grad_norms"):
        if k in d:
            d[k] = json.loads(str(d[k]))
    return d


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def bert_state_template(hidden, layers, inter, vocab, max_pos, pooler=True):
    """HF BertModel state-dict key -> shape, without instantiating transformers."""
    s = {"embeddings.word_embeddings.weight": (vocab, hidden),
         "embeddings.position_embeddings.weight": (max_pos, hidden),
         "embeddings.token_type_embeddings.weight": (2, hidden),
         "embeddings.LayerNorm.weight": (hidden,), "embeddings.LayerNorm.bias": (hidden,)}
    for i in range(layers):
        p = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (hidden, hidden)
            s[p + f"attention.self.{n}.bias"] = (hidden,)
        s[p + "attention.output.dense.weight"] = (hidden, hidden)
        s[p + "attention.output.dense.bias"] = (hidden,)
        s[p + "attention.output.LayerNorm.weight"] = (hidden,)
        s[p + "attention.output.LayerNorm.bias"] = (hidden,)
        s[p + "intermediate.dense.weight"] = (inter, hidden)
        s[p + "intermediate.dense.bias"] = (inter,)
        s[p + "output.dense.weight"] = (hidden, inter)
        s[p + "output.dense.bias"] = (hidden,)
        s[p + "output.LayerNorm.weight"] = (hidden,)
        s[p + "output.LayerNorm.bias"] = (hidden,)
    if pooler:
        s["pooler.dense.weight"] = (hidden, hidden)
        s["pooler.dense.bias"] = (hidden,)
    return s


def model_state_template(f_in, hc, c, plm, num_relations=5, num_bases=30):
    """main.GraphTextLM state-dict key -> shape."""
    p = plm["hidden"]
    dims = [f_in, hc, 2 * hc, 4 * hc, 8 * hc]
    s = {"gnn_mask_token_embed": (1, f_in)}
    for k in range(4):
        s[f"rgcn{k+1}.weight"] = (num_bases, dims[k], dims[k + 1])
        s[f"rgcn{k+1}.comp"] = (num_relations, num_bases)
        s[f"rgcn{k+1}.root"] = (dims[k], dims[k + 1])
        s[f"rgcn{k+1}.bias"] = (dims[k + 1],)
        for n in ("weight", "bias", "mean_scale"):
            s[f"gnorm{k+1}.{n}"] = (dims[k + 1],)
    for k, (i, o) in enumerate(((f_in, hc), (hc, 2 * hc), (2 * hc, 8 * hc))):
        s[f"residual_proj{k+1}.weight"] = (o, i)
        s[f"residual_proj{k+1}.bias"] = (o,)
    for k, v in bert_state_template(p, plm["layers"], plm["inter"], plm["vocab"], plm["max_pos"]).items():
        s["plm_encoder." + k] = v
    s["multi_scale_fusion.scale_weights"] = (4,)
    for k in range(4):
        s[f"multi_scale_fusion.projections.{k}.weight"] = (p, dims[k + 1])
        s[f"multi_scale_fusion.projections.{k}.bias"] = (p,)
    s["multi_scale_fusion.layer_norm.weight"] = (p,)
    s["multi_scale_fusion.layer_norm.bias"] = (p,)
    for a in ("graph_to_text_attn", "text_to_graph_attn"):
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[f"{a}.{n}.weight"] = (p, p)
            s[f"{a}.{n}.bias"] = (p,)
    s["fusion_network.0.weight"] = (p, 2 * p)
    s["fusion_network.0.bias"] = (p,)
    s["fusion_network.1.weight"] = (p,)
    s["fusion_network.1.bias"] = (p,)
    s["classifier.0.weight"] = (hc, p)
    s["classifier.0.bias"] = (hc,)
    s["classifier.3.weight"] = (c, hc)
    s["classifier.3.bias"] = (c,)
    return s


def oracle_model_from_config(cfg):
    from gmlm_oracle import OracleGraphTextLM
    from param_recipe import recipe_state_dict
    plm = cfg["plm"]
    sd = recipe_state_dict(model_state_template(cfg["f_in"], cfg["hc"], cfg["c"], plm), cfg["seed"])
    plm_sd = {k[len("plm_encoder."):]: v for k, v in sd.items() if k.startswith("plm_encoder.")}
    m = OracleGraphTextLM(cfg["f_in"], cfg["hc"], cfg["c"], plm_sd, plm["heads"])
    m.load_reference_state(sd)
    return m, sd


def attn_dropout_scale(seed: int, slab: int, nq: int, nk: int, p: float, device="cpu"):
    """The attention kernels' replayable probability-dropout mask, restated with torch integer arithmetic (checker only;
    DESIGN.md section 4 'Dropout'): one 32-bit hash word per 2 x 2 (query, key) tile,
    word = lowbias32(base + (q >> 1) * C1 + (k >> 1) * C2), base = (lo32(seed) ^ hi32(seed)) + slab * C3, the element's 8-bit
    field = bits [16 (q & 1) + 8 (k & 1), +8), kept when field >= round(256 p); kept elements are scaled by 256 / (256 - th).
    ``slab`` = (batch * heads + head) * lq for padded tensors.  Returns the [nq, nk] float64 multiplier (0 or keep scale)."""
    m32 = 0xFFFFFFFF
    c1, c2, c3 = 0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D
    th = int(round(p * 256.0))
    th = max(0, min(255, th))
    ks = 256.0 / (256.0 - th)
    base = (((seed & m32) ^ ((seed >> 32) & m32)) + (slab * c3)) & m32
    q = torch.arange(nq, dtype=torch.int64, device=device)[:, None]
    k = torch.arange(nk, dtype=torch.int64, device=device)[None, :]
    u = (base + (q >> 1) * c1 + (k >> 1) * c2) & m32
    u = u ^ (u >> 16)
    u = (u * 0x7FEB352D) & m32
    u = u ^ (u >> 15)
    field = (u >> (16 * (q & 1) + 8 * (k & 1))) & 0xFF
    return (field >= th).to(torch.float64) * ks
