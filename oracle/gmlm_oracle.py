"""CPU oracle for the GraphTextLM forward/backward hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-fp32 / numpy restatement of the algorithm the reference runs on the
hot path named by BASELINE.json (``GraphTextLM.forward`` + backward).  It is NOT part of the
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / timed CPU baseline.  ``gmlm_amd`` never imports it.

What it follows (file:line into /root/reference/main.py unless prefixed):

* ``degree``, ``edge_types_from_degree``      main.py:253-267 (+ PyG ``degree`` semantics)
* ``rgcn_conv`` / ``OracleRGCNConv``          call sites main.py:189-203, 272-308; arithmetic from
                                              PyG ``RGCNConv`` public documentation  [PyG, unpinned]
* ``graph_norm`` / ``OracleGraphNorm``        call sites main.py:190-202, 273-309; arithmetic from
                                              PyG ``GraphNorm`` public documentation [PyG, unpinned]
* ``soft_masking_gnn_input``                  main.py:92-99
* ``cross_attention``                         main.py:139-165
* ``multi_scale_fusion``                      main.py:167-180
* ``bert_encoder``                            hf:modeling_bert.py:53-108 (embeddings), 111-136 (attention),
                                              282-293 / 325-351 (output blocks), 354-416 (layer)
                                              (hf = transformers 5.15.0 in this image)
* ``masked_mean_pool``                        main.py:351-356
* ``OracleGraphTextLM``                       main.py:182-372

Parity pinning (see DESIGN.md "Oracle"):
* everything reference-owned (main.py) and the HF BERT block is PINNED: ``oracle/make_golden.py``
  imports /root/reference/main.py in the build container, runs ``main.GraphTextLM`` and writes
  ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against those vectors.
* the two PyG operators are "parity unpinned": PyG is not installed in this image and not
  vendored in the reference, so ``RGCNConv``/``GraphNorm``/``degree`` are restated from PyG's
  published semantics; the reference holds no test or fixture for them.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# integer / index part (bit-exact)
# ----------------------------------------------------------------------------------------------
def degree(index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """PyG ``degree``: float32 count of occurrences (main.py:65, 256)."""
    out = torch.zeros(num_nodes, dtype=torch.float32, device=index.device)
    return out.scatter_add_(0, index.to(torch.long), torch.ones(index.numel(), dtype=torch.float32))


def edge_types_loop(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Faithful per-edge Python loop of main.py:253-267 (slow; used for small cases + cpu_baseline)."""
    edge_index = edge_index.to(torch.long)
    num_edges = edge_index.size(1)
    edge_type = torch.zeros(num_edges, dtype=torch.long)
    deg = degree(edge_index[0], num_nodes)
    for i in range(num_edges):
        d = deg[edge_index[0, i]]
        if d <= 2:
            edge_type[i] = 0
        elif d <= 5:
            edge_type[i] = 1
        elif d <= 10:
            edge_type[i] = 2
        else:
            edge_type[i] = 3
    return edge_type


def edge_types_from_degree(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Vectorised numpy form of main.py:253-267; identical integers to ``edge_types_loop``."""
    src = edge_index[0].cpu().numpy().astype(np.int64)
    deg = np.bincount(src, minlength=num_nodes)
    d = deg[src]
    et = np.full(src.shape, 3, dtype=np.int64)
    et[d <= 10] = 2
    et[d <= 5] = 1
    et[d <= 2] = 0
    return torch.from_numpy(et)


def relation_csr(edge_index: torch.Tensor, edge_type: torch.Tensor, num_nodes: int, num_relations: int):
    """Target-sorted, relation-segmented CSR used to check the HIP K1 build.

    Segment s = dst * R + rel.  Within a segment edges keep their original order (stable).
    Returns (rowptr int32 [N*R+1], col int32 [E] (source ids), eid int32 [E] (original edge id)).
    """
    src = edge_index[0].cpu().numpy().astype(np.int64)
    dst = edge_index[1].cpu().numpy().astype(np.int64)
    rel = edge_type.cpu().numpy().astype(np.int64)
    key = dst * num_relations + rel
    order = np.argsort(key, kind="stable")
    counts = np.bincount(key, minlength=num_nodes * num_relations)
    rowptr = np.zeros(num_nodes * num_relations + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr.astype(np.int32), src[order].astype(np.int32), order.astype(np.int32)


# ----------------------------------------------------------------------------------------------
# PyG operators restated  [PyG, parity unpinned]
# ----------------------------------------------------------------------------------------------
def rgcn_mean_aggregate(x: torch.Tensor, edge_index: torch.Tensor, edge_type: torch.Tensor,
                        num_relations: int) -> torch.Tensor:
    """h[r, i, :] = mean over edges (j -> i) of type r of x[j]; 0 where a node has none.

    flow = source_to_target: edge_index[0] = j (message source), edge_index[1] = i (target).
    """
    n, f = x.shape
    out = x.new_zeros(num_relations, n, f)
    for r in range(num_relations):
        m = edge_type == r
        src, dst = edge_index[0, m], edge_index[1, m]
        s = x.new_zeros(n, f).index_add_(0, dst, x.index_select(0, src))
        cnt = x.new_zeros(n).index_add_(0, dst, torch.ones(dst.numel(), dtype=x.dtype))
        out[r] = s / cnt.clamp(min=1).unsqueeze(1)
    return out


def rgcn_conv(x, edge_index, edge_type, weight, comp, root, bias):
    """PyG RGCNConv(aggr='mean', num_bases=B, root_weight=True, bias=True) forward."""
    num_relations, num_bases = comp.shape
    in_c, out_c = weight.shape[1], weight.shape[2]
    w = (comp @ weight.view(num_bases, -1)).view(num_relations, in_c, out_c)
    h = rgcn_mean_aggregate(x, edge_index, edge_type, num_relations)
    out = x.new_zeros(x.size(0), out_c)
    for r in range(num_relations):
        out = out + h[r] @ w[r]
    out = out + x @ root
    return out + bias


def graph_norm(x, weight, bias, mean_scale, eps: float = 1e-5):
    """PyG GraphNorm with a single graph (batch = all rows)."""
    mean = x.mean(dim=0, keepdim=True)
    out = x - mean * mean_scale
    var = out.pow(2).mean(dim=0, keepdim=True)
    return weight * out / (var + eps).sqrt() + bias


def _glorot(t: torch.Tensor) -> None:
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


class OracleRGCNConv(nn.Module):
    """Parameter names / shapes / init of PyG ``RGCNConv`` (weight, comp, root, bias)."""

    def __init__(self, in_channels, out_channels, num_relations, num_bases=None, **_):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_relations, self.num_bases = num_relations, num_bases
        self.weight = nn.Parameter(torch.empty(num_bases, in_channels, out_channels))
        self.comp = nn.Parameter(torch.empty(num_relations, num_bases))
        self.root = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        _glorot(self.weight), _glorot(self.comp), _glorot(self.root)

    def forward(self, x, edge_index, edge_type):
        return rgcn_conv(x, edge_index, edge_type, self.weight, self.comp, self.root, self.bias)


class OracleGraphNorm(nn.Module):
    def __init__(self, in_channels, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = nn.Parameter(torch.ones(in_channels))

    def forward(self, x):
        return graph_norm(x, self.weight, self.bias, self.mean_scale, self.eps)


# ----------------------------------------------------------------------------------------------
# reference-owned pieces
# ----------------------------------------------------------------------------------------------
def soft_masking_gnn_input(x, mask, mask_token_embed, beta=0.7):
    """main.py:92-99."""
    xm = x.clone()
    if mask.any():
        xm[mask] = (1 - beta) * x[mask] + beta * mask_token_embed
    return xm


def cross_attention(x, y, wq, bq, wk, bk, wv, bv, wo, bo, num_heads=8):
    """main.py:151-165 with dropout off.  x, y: [N, C] (the batch-1 dim dropped)."""
    n, c = x.shape
    d = c // num_heads
    scale = d ** -0.5
    q = F.linear(x, wq, bq).view(n, num_heads, d).transpose(0, 1)
    k = F.linear(y, wk, bk).view(-1, num_heads, d).transpose(0, 1)
    v = F.linear(y, wv, bv).view(-1, num_heads, d).transpose(0, 1)
    attn = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1)
    o = (attn @ v).transpose(0, 1).reshape(n, c)
    return F.linear(o, wo, bo)


def multi_scale_fusion(embs: Sequence[torch.Tensor], scale_weights, proj_w, proj_b, ln_w, ln_b):
    """main.py:176-180."""
    w = F.softmax(scale_weights, dim=0)
    acc = sum(w[k] * F.linear(e, proj_w[k], proj_b[k]) for k, e in enumerate(embs))
    return F.layer_norm(acc, (acc.size(-1),), ln_w, ln_b, 1e-5)


def masked_mean_pool(last_hidden, attention_mask):
    """main.py:351-356."""
    m = attention_mask.unsqueeze(-1).expand(last_hidden.size()).float()
    return (last_hidden * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)


def bert_encoder(sd: dict, prefix: str, input_ids, attention_mask, num_heads: int, eps: float = 1e-12):
    """HF BertModel (encoder-only, eval/dropout 0, absolute positions, token type 0) -> last_hidden_state.

    ``sd`` maps HF parameter names (``embeddings.word_embeddings.weight`` ...) under ``prefix``.
    hf:modeling_bert.py:98-107 (embeddings), 125-133 (softmax(QK^T*d^-1/2 + mask) V),
    289-293 and 347-351 (dense -> LN(x + residual)), 333-336 (dense + GELU(erf)).
    """
    g = lambda k: sd[prefix + k]
    b, l = input_ids.shape
    h = g("embeddings.word_embeddings.weight")[input_ids]
    h = h + g("embeddings.token_type_embeddings.weight")[0]
    h = h + g("embeddings.position_embeddings.weight")[:l].unsqueeze(0)
    hid = h.size(-1)
    h = F.layer_norm(h, (hid,), g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
    d = hid // num_heads
    add_mask = torch.zeros(b, 1, 1, l, dtype=h.dtype)
    add_mask.masked_fill_(attention_mask[:, None, None, :] == 0, torch.finfo(h.dtype).min)
    i = 0
    while f"{prefix}encoder.layer.{i}.attention.self.query.weight" in sd:
        p = f"encoder.layer.{i}."
        q = F.linear(h, g(p + "attention.self.query.weight"), g(p + "attention.self.query.bias"))
        k = F.linear(h, g(p + "attention.self.key.weight"), g(p + "attention.self.key.bias"))
        v = F.linear(h, g(p + "attention.self.value.weight"), g(p + "attention.self.value.bias"))
        q, k, v = (t.view(b, l, num_heads, d).transpose(1, 2) for t in (q, k, v))
        s = (q @ k.transpose(2, 3)) * d ** -0.5 + add_mask
        ctx = (s.softmax(-1) @ v).transpose(1, 2).reshape(b, l, hid)
        a = F.linear(ctx, g(p + "attention.output.dense.weight"), g(p + "attention.output.dense.bias"))
        a = F.layer_norm(a + h, (hid,), g(p + "attention.output.LayerNorm.weight"),
                         g(p + "attention.output.LayerNorm.bias"), eps)
        m = F.gelu(F.linear(a, g(p + "intermediate.dense.weight"), g(p + "intermediate.dense.bias")))
        o = F.linear(m, g(p + "output.dense.weight"), g(p + "output.dense.bias"))
        h = F.layer_norm(o + a, (hid,), g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), eps)
        i += 1
    return h


# ----------------------------------------------------------------------------------------------
# whole model (state-dict-key compatible with main.GraphTextLM; PLM given as a plain tensor dict)
# ----------------------------------------------------------------------------------------------
class OracleGraphTextLM(nn.Module):
    """main.py:182-372 restated; dropout is a no-op (parity runs use dropout 0 / eval).

    The text encoder's weights are ``nn.Parameter``s registered under the HF names with the
    ``plm_encoder.`` prefix so ``load_state_dict`` from a reference state dict works.  Texts are
    replaced by pre-tokenised ``input_ids`` / ``attention_mask`` rows (one row per node), which is
    what ``main.py:342-345`` produces per micro-batch (padding to the batch max is emulated by
    trimming each micro-batch to its longest row).
    """

    def __init__(self, gnn_in_channels, hidden_channels, num_classes, plm_state: dict, plm_heads: int,
                 num_relations=5, num_bases=30, plm_eps=1e-12):
        super().__init__()
        hc = hidden_channels
        self.gnn_mask_token_embed = nn.Parameter(torch.zeros(1, gnn_in_channels))
        nn.init.xavier_uniform_(self.gnn_mask_token_embed)
        dims = [gnn_in_channels, hc, hc * 2, hc * 4, hc * 8]
        for k in range(4):
            setattr(self, f"rgcn{k+1}", OracleRGCNConv(dims[k], dims[k + 1], num_relations, num_bases))
            setattr(self, f"gnorm{k+1}", OracleGraphNorm(dims[k + 1]))
        self.residual_proj1 = nn.Linear(gnn_in_channels, hc)
        self.residual_proj2 = nn.Linear(hc, hc * 2)
        self.residual_proj3 = nn.Linear(hc * 2, hc * 8)
        self.plm_heads, self.plm_eps = plm_heads, plm_eps
        self.plm_names = list(plm_state.keys())
        self.plm_params = nn.ParameterDict({k.replace(".", "/"): nn.Parameter(v.clone().float())
                                            for k, v in plm_state.items() if v.is_floating_point()})
        p = plm_state["embeddings.word_embeddings.weight"].shape[1]
        self.hidden_size = p

        class _MSF(nn.Module):
            def __init__(s):
                super().__init__()
                s.scale_weights = nn.Parameter(torch.ones(4) / 4)
                s.projections = nn.ModuleList([nn.Linear(d, p) for d in dims[1:]])
                s.layer_norm = nn.LayerNorm(p)

        class _CA(nn.Module):
            def __init__(s):
                super().__init__()
                s.q_proj, s.k_proj, s.v_proj, s.out_proj = (nn.Linear(p, p) for _ in range(4))

        self.multi_scale_fusion = _MSF()
        self.graph_to_text_attn = _CA()
        self.text_to_graph_attn = _CA()
        self.fusion_network = nn.Sequential(nn.Linear(2 * p, p), nn.LayerNorm(p), nn.GELU(), nn.Dropout(0.0))
        self.classifier = nn.Sequential(nn.Linear(p, hc), nn.GELU(), nn.Dropout(0.0), nn.Linear(hc, num_classes))

    # -- state-dict bridge: reference keys 'plm_encoder.<hf name>' <-> our ParameterDict -----------
    def load_reference_state(self, sd: dict) -> None:
        own = {}
        for k, v in sd.items():
            if k.startswith("plm_encoder."):
                own["plm_params." + k[len("plm_encoder."):].replace(".", "/")] = v
            else:
                own[k] = v
        missing, unexpected = self.load_state_dict(own, strict=False)
        assert not unexpected, unexpected
        assert not missing, missing

    def _plm_sd(self):
        return {k.replace("/", "."): v for k, v in self.plm_params.items()}

    def get_graph_embeddings(self, x, edge_index, edge_type=None, return_layers=False):
        edge_index = edge_index.to(torch.long)
        if edge_type is None:
            edge_type = edge_types_from_degree(edge_index, x.size(0))
        embs = []
        h = x
        for k in range(1, 5):
            y = getattr(self, f"rgcn{k}")(h, edge_index, edge_type)
            if y.size(0) > 1:
                y = getattr(self, f"gnorm{k}")(y)
            y = F.gelu(y)
            embs.append(y)
            if k == 1:
                h = y + self.residual_proj1(x)
            elif k == 2:
                h = y + self.residual_proj2(h)
            else:
                h = y  # main.py:297-318: layer 4 takes raw x3; x4 + residual_proj3(x2) is discarded
        m = self.multi_scale_fusion
        out = multi_scale_fusion(embs, m.scale_weights, [q.weight for q in m.projections],
                                 [q.bias for q in m.projections], m.layer_norm.weight, m.layer_norm.bias)
        return (out, embs) if return_layers else out

    def encode_texts(self, input_ids, attention_mask, node_mask, plm_batch_size=8):
        """main.py:328-358 on pre-tokenised rows; returns plm_embeds [N, P]."""
        n = node_mask.numel()
        plm = torch.zeros(n, self.hidden_size)
        idx = node_mask.nonzero(as_tuple=True)[0]
        sd = self._plm_sd()
        for s in range(0, idx.numel(), plm_batch_size):
            bi = idx[s:s + plm_batch_size]
            am = attention_mask[bi]
            lmax = int(am.sum(1).max().item())
            ids, am = input_ids[bi, :lmax], am[:, :lmax]
            hs = bert_encoder(sd, "", ids, am, self.plm_heads, self.plm_eps)
            plm = plm.index_put((bi,), masked_mean_pool(hs, am))
        return plm

    def forward(self, x, edge_index, input_ids, attention_mask, node_mask, edge_type=None, plm_batch_size=8,
                return_parts=False):
        gnn = self.get_graph_embeddings(x, edge_index, edge_type)
        plm = self.encode_texts(input_ids, attention_mask, node_mask, plm_batch_size)

        def ca(mod, a, b):
            return cross_attention(a, b, mod.q_proj.weight, mod.q_proj.bias, mod.k_proj.weight, mod.k_proj.bias,
                                   mod.v_proj.weight, mod.v_proj.bias, mod.out_proj.weight, mod.out_proj.bias, 8)

        g_att = ca(self.graph_to_text_attn, gnn, plm)
        t_att = ca(self.text_to_graph_attn, plm, gnn)
        fused = self.fusion_network(torch.cat([g_att, t_att], dim=-1))
        logits = self.classifier(fused)
        if return_parts:
            return logits, dict(gnn_embeds=gnn, plm_embeds=plm, gnn_attended=g_att, text_attended=t_att)
        return logits


# ----------------------------------------------------------------------------------------------
# synthetic workloads of SURVEY.md §8d (shared by tests and bench so both sides see the same bytes)
# ----------------------------------------------------------------------------------------------
WORKLOADS = {
    # name: N, E, F_in, C
    "toy": (64, 256, 32, 5),
    "cornell": (183, 298, 1703, 5),
    "chameleon": (2277, 36101, 2325, 5),
    "squirrel": (5201, 217073, 2089, 5),
    "arxiv": (169343, 1166243, 128, 40),
}


def synthetic_graph(name: str, seed_offset: int = 0):
    n, e, f_in, c = WORKLOADS[name]
    g = torch.Generator().manual_seed(1000 + list(WORKLOADS).index(name) + seed_offset)
    x = torch.randn(n, f_in, generator=g)
    edge_index = torch.randint(0, n, (2, e), generator=g, dtype=torch.long)
    y = torch.randint(0, c, (n,), generator=g)
    perm = torch.randperm(n, generator=g)
    train_mask = torch.zeros(n, dtype=torch.bool)
    train_mask[perm[: int(0.48 * n)]] = True
    active = train_mask & (torch.rand(n, generator=g) < 0.5)
    return dict(x=x, edge_index=edge_index, y=y, train_mask=train_mask, active_mask=active, num_classes=c)


def synthetic_tokens(n: int, max_len: int, vocab: int, seed: int, min_len: int = 16):
    g = torch.Generator().manual_seed(seed)
    min_len = min(min_len, max_len)
    lens = torch.randint(min_len, max_len + 1, (n,), generator=g)
    ids = torch.randint(5, vocab, (n, max_len), generator=g)
    am = (torch.arange(max_len)[None, :] < lens[:, None]).long()
    return ids * am, am


def split_plan(rowptr: np.ndarray, thresh: int):
    """Numpy restatement of gmlm_amd.graph.make_split_plan (chunking of long CSR segments): checker only."""
    lens = np.diff(rowptr.astype(np.int64))
    long_seg = np.nonzero(lens > thresh)[0]
    nch = (lens[long_seg] + thresh - 1) // thresh
    chunk_ptr = np.concatenate([[0], np.cumsum(nch)])
    owner = np.repeat(np.arange(long_seg.size), nch)
    return long_seg.astype(np.int32), chunk_ptr.astype(np.int32), owner.astype(np.int32)
