"""bf16-EMULATING mode of the CPU oracle.  TEST INFRASTRUCTURE ONLY (same rules as gmlm_oracle.py).

The reference runs its GPU path under ``torch.amp.autocast('cuda')`` (main.py:348, 446, 543): reduced-precision GEMM /
attention operands, fp32 accumulation, fp32 normalisation statistics.  The HIP path's reduced precision is bf16.  This
module restates the SAME algorithm as ``gmlm_oracle.OracleGraphTextLM`` (main.py:182-372, file:line citations there) in
fp32 on the CPU, but rounds to bf16 at every point where the bf16 HIP path holds a tensor in bf16:

  * every operand of a GEMM and of the attention products (Q, K, V, dO; the probabilities P before P.V; dS before dS.K
    and dS^T.Q), products and sums in fp32;
  * every activation and every activation-gradient the path STORES between two kernels;
  * parameters are fp32 masters, rounded where they enter a GEMM; weight / bias gradients of every dense layer stay fp32
    (fp32 GEMM outputs / split-K fp32 partials / fixed-order fp32 column sums straight to the masters); the RGCN relation
    weights' gradient is a bf16 GEMM output (it feeds the basis-composition backward).

It does NOT call, import or share code with ``gmlm_amd``; it follows the order of operations that DESIGN.md sections 3-4
document for the kernels (K2-K10), including the kernel-specific details that decide which values get rounded:
the short-sequence attention kernels keep Q unscaled and form exp2(fma(s, scale*log2e, -m)); the pipelined forward and the
long backward kernels pre-multiply Q (resp. K) by scale*log2e and round it once more, subtract a bf16-representable
reference max that is raised per 32-query wave and 32-key block with a deferral of 2^6, and split lse / delta into two bf16
terms.  With the rounding points aligned, the two bf16 computations agree far more tightly than either agrees with fp32
(values about to be rounded agree to ~1e-6, so the same bf16 number comes out except for rare boundary flips): that is what
lets a test tell a kernel defect (large deviation) from reduced precision (the emulation shows the same deviation).

Dropout is off in this mode (parity runs use p = 0)."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

import gmlm_oracle as O

LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453
K_DEFER = 6.0


def r(x: torch.Tensor) -> torch.Tensor:
    """Round to the nearest bf16 value (ties to even), returned as fp32."""
    return x.to(torch.bfloat16).to(torch.float32)


class _Store(torch.autograd.Function):
    """A tensor the path holds in bf16: the value is rounded going forward, its gradient is rounded coming back."""

    @staticmethod
    def forward(ctx, x):
        return r(x)

    @staticmethod
    def backward(ctx, g):
        return r(g)


def st(x):
    return _Store.apply(x)


class _Lin(torch.autograd.Function):
    """y = bf16( x~ w~^T + b~ ) with fp32 accumulation.  Backward (dy arrives as a stored bf16 tensor):
    dx = bf16(dy w~ [+ dres]);  dw = dy^T x~ and db = colsum(dy), each rounded to bf16 when ``round_wgrad`` (a bf16 GEMM /
    reduction output), fp32 otherwise (split-K fp32 partials).  ``residual``: x is also returned as a second output whose
    gradient is added inside the data-gradient GEMM's epilogue (ONE rounding of the sum)."""

    @staticmethod
    def forward(ctx, x, w, b, round_wgrad, residual):
        wt = r(w)
        ctx.save_for_backward(x, wt)
        ctx.cfg = (round_wgrad, b is not None, w.shape)
        y = x @ wt.t()
        if b is not None:
            y = y + r(b)
        y = r(y)
        return (y, x.clone()) if residual else y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, wt = ctx.saved_tensors
        round_wgrad, has_b, _ = ctx.cfg
        dy = r(dy)
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = dy2 @ wt
        if dres is not None:
            dx = dx + r(dres).reshape(dx.shape)
        dx = r(dx).view(x.shape)
        dw = dy2.t() @ x2
        db = dy2.sum(0) if has_b else None
        if round_wgrad:
            dw = r(dw)
            db = r(db) if has_b else None
        return dx, dw, db, None, None


def lin(x, w, b=None, round_wgrad=True, residual=False):
    return _Lin.apply(x, w, b, round_wgrad, residual)


# ------------------------------------------------------------------------------------------------------------------
# GNN
# ------------------------------------------------------------------------------------------------------------------
def _aggregate(x, edge_index, edge_type, rels):
    """[n, R_a * f]: per-(target, relation) mean of the source rows (fp32 sum, * 1/count, stored bf16), relation blocks side
    by side; only the relations that occur get a block."""
    n = x.shape[0]
    hs = O.rgcn_mean_aggregate(x, edge_index, edge_type, 5)
    return st(torch.cat([hs[rel] for rel in rels], 1))


def _rgcn_block(conv, norm, x, edge_index, edge_type, rels):
    """z = bf16(bf16(bias~ + x~ root~) + H~ W~cat); y = bf16(gelu(GraphNorm_fp32(z)))."""
    nb = conv.num_bases
    w_rel = (conv.comp[rels] @ conv.weight.view(nb, -1)).view(len(rels), conv.in_channels, conv.out_channels)   # fp32 (K10)
    pad = x.shape[1] - conv.in_channels
    root = conv.root
    if pad:
        w_rel = F.pad(w_rel, (0, 0, 0, pad))
        root = F.pad(root, (0, 0, 0, pad))
    out1 = lin(x, root.t(), conv.bias, round_wgrad=False)                 # addmm(bias, x, root) -> bf16; root / bias gradients fp32 (nn._RootAddmm)
    h = _aggregate(x, edge_index, edge_type, rels)
    z = st(out1 + lin_noround(h, w_rel.reshape(-1, conv.out_channels).t()))   # out.addmm_(h, w): one rounding of the sum
    if z.shape[0] > 1:
        z = O.graph_norm(z, norm.weight, norm.bias, norm.mean_scale, norm.eps)
    return st(F.gelu(z))


class _LinNoRound(torch.autograd.Function):
    """x~ w~^T in fp32 WITHOUT rounding the result (it is added to an accumulator operand inside the GEMM epilogue and the SUM
    is what gets rounded); backward as _Lin with bf16 GEMM outputs."""

    @staticmethod
    def forward(ctx, x, w):
        wt = r(w)
        ctx.save_for_backward(x, wt)
        return x @ wt.t()

    @staticmethod
    def backward(ctx, dy):
        x, wt = ctx.saved_tensors
        dy = r(dy)
        return r(dy @ wt), r(dy.t() @ x)


def lin_noround(x, w):
    return _LinNoRound.apply(x, w)


class _ScaledProjectionSum(torch.autograd.Function):
    """MultiScaleFusion ahead of its LayerNorm (main.py:176-179): acc = sum_k e_k~ bf16(w_k s_k)^T + sum_k b_k s_k as ONE GEMM
    over the concatenated operands (fp32 accumulation and fp32 output: nn._ScaledProjectionCat); backward with the gradient
    rounded once to bf16 for the GEMMs, weight gradients fp32."""

    @staticmethod
    def forward(ctx, weights, *args):
        k = len(args) // 3
        embs, ws, bs = args[:k], args[k:2 * k], args[2 * k:]
        acc = None
        for i in range(k):                                   # ONE GEMM over the concatenated operands: fp32 out, no rounding per scale
            t = embs[i] @ r(ws[i] * weights[i]).t()
            acc = t if acc is None else acc + t
        acc = acc + sum(bs[i] * weights[i] for i in range(k))
        ctx.save_for_backward(weights, *embs, *ws, *bs)
        ctx.k = k
        return acc

    @staticmethod
    def backward(ctx, g):
        k = ctx.k
        weights = ctx.saved_tensors[0]
        embs, ws, bs = ctx.saved_tensors[1:1 + k], ctx.saved_tensors[1 + k:1 + 2 * k], ctx.saved_tensors[1 + 2 * k:]
        gc, gsum = r(g), g.sum(0)
        d_weights = torch.zeros_like(weights)
        d_embs, d_ws, d_bs = [], [], []
        for i in range(k):
            d_embs.append(r(gc @ r(ws[i] * weights[i])))
            dws = gc.t() @ embs[i]                            # fp32 weight-gradient GEMM output
            d_ws.append(dws * weights[i])
            d_bs.append(gsum * weights[i])
            d_weights[i] = (dws * ws[i]).sum() + (gsum * bs[i]).sum()
        return (d_weights, *d_embs, *d_ws, *d_bs)


def graph_embeddings(om, x0, edge_index, edge_type):
    """get_graph_embeddings (main.py:250-320) on the bf16 path; x0 = the stored (bf16, column-padded) input."""
    rels = sorted(set(edge_type.tolist())) or [0]
    f_in = om.rgcn1.in_channels
    e1 = _rgcn_block(om.rgcn1, om.gnorm1, x0, edge_index, edge_type, rels)
    x1 = st(e1 + lin(x0[:, :f_in], om.residual_proj1.weight, om.residual_proj1.bias, round_wgrad=False))
    e2 = _rgcn_block(om.rgcn2, om.gnorm2, x1, edge_index, edge_type, rels)
    x2 = st(e2 + lin(x1, om.residual_proj2.weight, om.residual_proj2.bias, round_wgrad=False))
    e3 = _rgcn_block(om.rgcn3, om.gnorm3, x2, edge_index, edge_type, rels)
    e4 = _rgcn_block(om.rgcn4, om.gnorm4, e3, edge_index, edge_type, rels)
    m = om.multi_scale_fusion
    w = F.softmax(m.scale_weights, 0)
    acc = _ScaledProjectionSum.apply(w, e1, e2, e3, e4, *[p.weight for p in m.projections], *[p.bias for p in m.projections])
    return F.layer_norm(acc, (acc.shape[-1],), m.layer_norm.weight, m.layer_norm.bias, 1e-5)      # fp32 in, fp32 out


# ------------------------------------------------------------------------------------------------------------------
# attention: short-sequence kernels (<= 128 tokens, d = 64) on a padded [B, h, L, d] batch with key / query lengths
# ------------------------------------------------------------------------------------------------------------------
class _ShortAttention(torch.autograd.Function):
    """Forward: s = q~ k~^T (fp32); m = rowmax(s) * c, c = scale*log2e; e = exp2(fma(s, c, -m)); l = sum e (fp32, not rounded);
    o = bf16((bf16(e) v~) * (1/l)); lse = (m + log2 l) ln2.
    Backward: p = exp2(fma(s, c, -lse*log2e)); dp = do~ v~^T; delta = sum_k p dp IN FP32 (the kernel keeps P and dP of all key
    blocks of a query in registers; it does not read the rounded forward output); ds = p (dp - delta);
    dq = bf16(scale * bf16(ds) k~); dk = bf16(scale * bf16(ds)^T q~); dv = bf16(bf16(p)^T do~)."""

    @staticmethod
    def forward(ctx, q, k, v, lens, scale):
        # q, k, v: [B, h, L, d] (bf16-valued fp32); lens [B]
        b, h, l, d = q.shape
        valid = (torch.arange(l)[None, :] < lens[:, None])                      # [B, L]
        kmask = valid[:, None, None, :]
        c = scale * LOG2E
        s = q @ k.transpose(-1, -2)
        s = s.masked_fill(~kmask, float("-inf"))
        m = s.amax(-1, keepdim=True) * c
        e = torch.exp2(s * c - m)
        lsum = e.sum(-1, keepdim=True)
        o = r((r(e) @ v) * (1.0 / lsum))
        lse = (m + torch.log2(lsum)) * LN2
        ctx.save_for_backward(q, k, v, o, lse, valid)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, valid = ctx.saved_tensors
        scale = ctx.scale
        c = scale * LOG2E
        do = r(do)
        kmask = valid[:, None, None, :]
        qmask = valid[:, None, :, None]
        s = q @ k.transpose(-1, -2)
        p = torch.exp2(s * c - lse * LOG2E).masked_fill(~kmask, 0.0).masked_fill(~qmask, 0.0)
        dp = do @ v.transpose(-1, -2)
        delta = (p * dp).sum(-1, keepdim=True)
        ds = r(p * (dp - delta))
        dq = r((ds @ k) * scale)
        dk = r((ds.transpose(-1, -2) @ q) * scale)
        dv = r(r(p).transpose(-1, -2) @ do)
        return dq, dk, dv, None, None


# ------------------------------------------------------------------------------------------------------------------
# attention: pipelined forward + long backward kernels (CrossAttention geometry, one sequence of N rows, no mask)
# ------------------------------------------------------------------------------------------------------------------
def _pipe_forward(q, k, v, scale):
    """q, k, v: [h, N, d] bf16-valued.  Returns o (bf16-valued), lse.  32-key blocks; the reference max m (log2 domain) of a
    query starts at bf16(max of block 0) and is re-referenced for ALL 32 queries of a wave when any of them sees a score more
    than 2^6 above its reference (then m <- max(m, bf16(block max))); P~ = bf16(exp2(s' - m)); l and O carry alpha = exp2(m_old - m_new)."""
    h, n, d = q.shape
    c = scale * LOG2E
    qs = r(q * c)
    nq_pad = (n + 31) // 32 * 32
    m = torch.zeros(h, n)
    l = torch.zeros(h, n)
    o = torch.zeros(h, n, d)
    for u, k0 in enumerate(range(0, n, 32)):
        kb, vb = k[:, k0:k0 + 32], v[:, k0:k0 + 32]
        sc = qs @ kb.transpose(-1, -2) - m[..., None]                          # [h, N, <=32]
        g = sc.amax(-1)                                                        # block max relative to the reference
        if u == 0:
            rare = torch.ones(h, n, dtype=torch.bool)
        else:
            over = F.pad(g > K_DEFER, (0, nq_pad - n)).view(h, nq_pad // 32, 32).any(-1)          # wave vote
            rare = over[..., None].expand(-1, -1, 32).reshape(h, nq_pad)[:, :n]
        m_new = r(m + g)
        if u != 0:
            m_new = torch.maximum(m, m_new)
        m_new = torch.where(rare, m_new, m)
        delta = m_new - m
        alpha = torch.where(rare & (u != 0), torch.exp2(-delta), torch.ones_like(delta))
        sc = sc - delta[..., None]
        e = torch.exp2(sc)
        l = l * alpha + e.sum(-1)
        o = o * alpha[..., None] + r(e) @ vb
        m = m_new
    of = o * (1.0 / l)[..., None]
    out = r(of)
    lse = (m + torch.log2(l)) * LN2
    return out, r(of - out), lse                                              # output, its bf16 rounding residual, log-sum-exp


def _split2(x):
    hi = r(x)
    return hi, r(x - hi)


class _LongAttention(torch.autograd.Function):
    """Forward = _pipe_forward (which also keeps the bf16 residual of its output).  Backward = the delta / dQ / dK-dV kernels
    with the folded chains; delta = rowsum(do~ * (o + o_lo)):
       dQ kernel : s = -lse_hi - lse_lo + bf16(q~ c) k~^T, p = exp2(s), dp = -d_hi - d_lo + do~ v~^T, dq = bf16(scale bf16(p dp) k~)
       dKV kernel: s = -lse_hi - lse_lo + q~ bf16(k~ c)^T, p = exp2(s), dv = bf16(bf16(p)^T do~), dk = bf16(scale bf16(p dp)^T q~)"""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        o, o_lo, lse = _pipe_forward(q, k, v, scale)
        ctx.save_for_backward(q, k, v, o, o_lo, lse)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, o_lo, lse = ctx.saved_tensors
        scale = ctx.scale
        c = scale * LOG2E
        do = r(do)
        delta = ((o + o_lo) * do).sum(-1)                                     # delta kernel: O = out + out_lo (2^-17)
        lh, ll = _split2(lse * LOG2E)
        dh, dl = _split2(delta)
        dp = (do @ v.transpose(-1, -2) - dh[..., None]) - dl[..., None]
        # dQ kernel (query pre-scaled)
        p1 = torch.exp2((r(q * c) @ k.transpose(-1, -2) - lh[..., None]) - ll[..., None])
        dq = r((r(p1 * dp) @ k) * scale)
        del p1
        # dK / dV kernel (key pre-scaled)
        p2 = torch.exp2((q @ r(k * c).transpose(-1, -2) - lh[..., None]) - ll[..., None])
        dv = r(r(p2).transpose(-1, -2) @ do)
        dk = r((r(p2 * dp).transpose(-1, -2) @ q) * scale)
        return dq, dk, dv, None


def _heads(t, h):
    n, cdim = t.shape
    return t.view(n, h, cdim // h).transpose(0, 1)


def cross_attention(mod, x, y, num_heads=8):
    """main.py:151-165 on the bf16 path: fused K|V projection, streaming attention, out projection."""
    n, cdim = x.shape
    xq, yk = st(x), st(y)                                                      # .to(bf16) of the fp32 embeddings
    q = lin(xq, mod.q_proj.weight, mod.q_proj.bias, round_wgrad=False)
    kv = lin(yk, torch.cat([mod.k_proj.weight, mod.v_proj.weight], 0), torch.cat([mod.k_proj.bias, mod.v_proj.bias], 0),
             round_wgrad=False)
    d = cdim // num_heads
    o = _LongAttention.apply(_heads(q, num_heads), _heads(kv[:, :cdim], num_heads), _heads(kv[:, cdim:], num_heads), d ** -0.5)
    o = o.transpose(0, 1).reshape(n, cdim)
    return lin(o, mod.out_proj.weight, mod.out_proj.bias, round_wgrad=False)


# ------------------------------------------------------------------------------------------------------------------
# text encoder (hf:modeling_bert.py:53-416) on one PACKED batch of all active nodes
# ------------------------------------------------------------------------------------------------------------------
def _ln(x, bias, res, w, b, eps):
    """K6: bf16(LN_fp32(x~ + bias + res~))"""
    z = x if bias is None else x + bias
    if res is not None:
        z = z + res
    return st(F.layer_norm(z, (z.shape[-1],), w, b, eps))


class _QKVBiasGrad(torch.autograd.Function):
    """The fused projection adds the Q|K|V biases; their gradient comes out of the attention backward kernel: column sums of
    the stored dq and dv rows (fp32), exactly 0 for dk (the rows of dS sum to zero)."""

    @staticmethod
    def forward(ctx, qkv, bq, bk, bv):
        return qkv.clone()

    @staticmethod
    def backward(ctx, g):
        p = g.shape[-1] // 3
        s = g.sum(0)
        return g, s[:p], torch.zeros(p), s[2 * p:]


def bert_packed(om, ids, lens, heads, eps):
    """ids [B, Lmax] (rows sorted as the caller likes), lens [B] -> pooled [B, P] fp32 (masked mean, main.py:351-356)."""
    sd = om._plm_sd()
    g = lambda kk: sd[kk]
    b, lmax = ids.shape
    valid = torch.arange(lmax)[None, :] < lens[:, None]
    tok = ids[valid]                                                           # packed [T]
    pos = torch.arange(lmax)[None, :].expand(b, -1)[valid]
    x = st(g("embeddings.word_embeddings.weight")[tok] + g("embeddings.token_type_embeddings.weight")[0]
           + g("embeddings.position_embeddings.weight")[pos])
    hdim = x.shape[-1]
    d = hdim // heads
    hcur = _ln(x, None, None, g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
    i = 0

    def to_padded(t):                                                          # [T, h*d] -> [B, h, L, d]
        out = t.new_zeros(b, lmax, t.shape[-1])
        out[valid] = t
        return out.view(b, lmax, heads, d).transpose(1, 2)

    while f"encoder.layer.{i}.attention.self.query.weight" in sd:
        p = f"encoder.layer.{i}."
        wq, wk, wv = (g(p + f"attention.self.{n_}.weight") for n_ in ("query", "key", "value"))
        bq, bk, bv = (g(p + f"attention.self.{n_}.bias") for n_ in ("query", "key", "value"))
        # fused projection: bias added in the GEMM, weight gradient fp32; the bias gradients come from the attention backward
        qkv, h_res = lin(hcur, torch.cat([wq, wk, wv], 0), torch.cat([bq.detach(), bk.detach(), bv.detach()], 0),
                         round_wgrad=False, residual=True)
        qkv = _QKVBiasGrad.apply(qkv, bq, bk, bv)
        o = _ShortAttention.apply(to_padded(qkv[:, :hdim]), to_padded(qkv[:, hdim:2 * hdim]), to_padded(qkv[:, 2 * hdim:]),
                                  lens, d ** -0.5)
        ctxv = o.transpose(1, 2).reshape(b, lmax, hdim)[valid]
        a = _ln(lin(ctxv, g(p + "attention.output.dense.weight"), None, round_wgrad=False),
                g(p + "attention.output.dense.bias"), h_res, g(p + "attention.output.LayerNorm.weight"),
                g(p + "attention.output.LayerNorm.bias"), eps)
        m_pre, a_res = lin(a, g(p + "intermediate.dense.weight"), None, round_wgrad=False, residual=True)
        mm = st(F.gelu(m_pre + g(p + "intermediate.dense.bias")))
        hcur = _ln(lin(mm, g(p + "output.dense.weight"), None, round_wgrad=False), g(p + "output.dense.bias"), a_res,
                   g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), eps)
        i += 1
    # K8: fp32 sum of the stored rows / length
    seq = torch.arange(b)[:, None].expand(-1, lmax)[valid]
    pooled = torch.zeros(b, hdim).index_add_(0, seq, hcur) / lens[:, None].clamp(min=1e-9)
    return pooled


# ------------------------------------------------------------------------------------------------------------------
# whole model
# ------------------------------------------------------------------------------------------------------------------
def forward(om: "O.OracleGraphTextLM", x, edge_index, input_ids, attention_mask, node_mask, beta=0.7, return_parts=False):
    """GraphTextLM.forward (main.py:322-372) + the soft mask (main.py:92-99) on the bf16 path, autograd to ``om``'s parameters.
    ``x``: RAW features (the soft mask is applied here because its output is the first stored bf16 tensor)."""
    n, f_in = x.shape
    edge_index = edge_index.to(torch.long)
    edge_type = O.edge_types_from_degree(edge_index, n)
    xm = O.soft_masking_gnn_input(x, node_mask, om.gnn_mask_token_embed, beta)
    pad = (-f_in) % 8
    x0 = st(F.pad(xm, (0, pad)) if pad else xm)
    gnn = graph_embeddings(om, x0, edge_index, edge_type)
    idx = node_mask.nonzero(as_tuple=True)[0]
    plm = torch.zeros(n, om.hidden_size)
    if idx.numel():
        lens = attention_mask[idx].sum(1)
        lmax = int(lens.max())
        pooled = bert_packed(om, input_ids[idx, :lmax], lens, om.plm_heads, om.plm_eps)
        plm = plm.index_put((idx,), pooled)
    g_att = cross_attention(om.graph_to_text_attn, gnn, plm)
    t_att = cross_attention(om.text_to_graph_attn, plm, gnn)
    fn, cl = om.fusion_network, om.classifier
    p = g_att.shape[-1]
    fused = lin(torch.cat([g_att, t_att], -1), fn[0].weight, round_wgrad=False)   # ONE GEMM over [g | t], K = 2P
    fused = st(F.gelu(F.layer_norm(fused + fn[0].bias, (p,), fn[1].weight, fn[1].bias, fn[1].eps)))
    hcls = st(F.gelu(lin(fused, cl[0].weight, round_wgrad=False) + cl[0].bias))
    logits = lin(hcls, cl[3].weight, cl[3].bias, round_wgrad=False)
    if return_parts:
        return logits, dict(gnn_embeds=gnn, plm_embeds=plm, gnn_attended=g_att, text_attended=t_att)
    return logits
