"""BERT-geometry text encoder forward on the gfx950 kernels.

Replaces the forward of HuggingFace ``BertModel`` as the reference runs it (main.py:349; classes at
hf:modeling_bert.py:53-108 embeddings, 139-203 self-attention, 282-293 / 325-351 dense epilogues,
354-416 layer) while keeping the HF module as the *parameter container*, so state-dict keys
(``plm_encoder.*``), ``config.hidden_size`` and ``base_model_prefix`` stay what callers expect
(main.py:222, 328, 380, 409).

Per layer:  fused QKV GEMM (hipBLASLt) -> K5 masked attention (MFMA, per-sequence key length)
            -> out-proj GEMM -> K6 bias+dropout+residual+LayerNorm
            -> FFN-in GEMM -> bias+GELU(+dropout) kernel -> FFN-out GEMM -> K6.
Also registers the attention core under HF's AttentionInterface as ``"gmlm_hip"`` so an unmodified
HF BertModel can call K5 through ``config._attn_implementation``.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.utils.checkpoint import checkpoint

from . import ops
from .nn import _linear, attention_any_dim


def check_supported(plm) -> None:
    cfg = plm.config
    problems = []
    if getattr(cfg, "model_type", None) != "bert":
        problems.append(f"model_type={getattr(cfg, 'model_type', None)!r} (only BERT-geometry encoders are on the kernel path)")
    if getattr(cfg, "position_embedding_type", "absolute") not in ("absolute", None):
        problems.append("non-absolute position embeddings")
    if getattr(cfg, "hidden_act", "gelu") != "gelu":
        problems.append(f"hidden_act={cfg.hidden_act!r}")
    if getattr(cfg, "is_decoder", False) or getattr(cfg, "add_cross_attention", False):
        problems.append("decoder / cross-attention BERT")
    if problems:
        raise NotImplementedError("gmlm_amd text-encoder kernels do not cover this PLM: " + "; ".join(problems))


class LayerWeights:
    """Per-forward view of one BertLayer's parameters: the GEMM operands in the compute dtype (``w*`` / ``bqkv``,
    no autograd history) next to the fp32 master parameters they were cast from.  Built ONCE per forward and
    shared by all micro-batches.  The GEMM wrappers below compute with the cached copy and hand the weight /
    bias gradients straight to the MASTERS in fp32 (the split-K partial sums are fp32 already), so neither the
    per-weight cast kernels nor their backward casts exist."""
    __slots__ = ("wqkv", "bqkv", "wo", "wi", "wo2", "m_qkv", "m_wo", "m_wi", "m_wo2", "bo", "bi", "bo2", "ln1w", "ln1b",
                 "ln2w", "ln2b")


def _layer_masters(layer):
    sa, so = layer.attention.self, layer.attention.output
    return ([sa.query.weight, sa.key.weight, sa.value.weight], [sa.query.bias, sa.key.bias, sa.value.bias],
            so.dense.weight, layer.intermediate.dense.weight, layer.output.dense.weight)


def prepare_weights(plm, cd):
    """One flat buffer in the compute dtype for all layers' GEMM weights, filled by ONE multi-tensor copy
    (Q/K/V rows land directly in the fused [3P, P] operand, so there is no concatenation either)."""
    layers = list(plm.encoder.layer)
    out, srcs, dsts = [], [], []
    with torch.no_grad():
        dev = layers[0].attention.self.query.weight.device if layers else None

        def take(shape):
            nonlocal off
            n = 1
            for d in shape:
                n *= d
            v = flat[off:off + n].view(shape)
            off += padded(n)
            return v

        def padded(n):                                        # keep every operand 256-byte aligned
            return (n + 127) // 128 * 128

        total = 0
        for layer in layers:
            wq, bq, wo, wi, wo2 = _layer_masters(layer)
            total += padded(sum(t.numel() for t in wq)) + padded(sum(t.numel() for t in bq)) + padded(wo.numel()) \
                + padded(wi.numel()) + padded(wo2.numel())
        flat = torch.empty(total, dtype=cd, device=dev)
        off = 0
        for layer in layers:
            sa, so = layer.attention.self, layer.attention.output
            wq, bq, wo, wi, wo2 = _layer_masters(layer)
            lw = LayerWeights()
            p = wq[0].shape[1]
            lw.wqkv = take((sum(t.shape[0] for t in wq), p))
            lw.bqkv = take((sum(t.shape[0] for t in bq),))
            lw.wo, lw.wi, lw.wo2 = take(tuple(wo.shape)), take(tuple(wi.shape)), take(tuple(wo2.shape))
            r = 0
            for w_, b_ in zip(wq, bq):
                dsts += [lw.wqkv[r:r + w_.shape[0]], lw.bqkv[r:r + w_.shape[0]]]
                srcs += [w_.detach(), b_.detach()]
                r += w_.shape[0]
            dsts += [lw.wo, lw.wi, lw.wo2]
            srcs += [wo.detach(), wi.detach(), wo2.detach()]
            lw.m_qkv, lw.m_wo, lw.m_wi, lw.m_wo2 = tuple(wq) + tuple(bq), (wo,), (wi,), (wo2,)
            lw.bo, lw.bi, lw.bo2 = so.dense.bias, layer.intermediate.dense.bias, layer.output.dense.bias
            lw.ln1w, lw.ln1b = so.LayerNorm.weight, so.LayerNorm.bias
            lw.ln2w, lw.ln2b = layer.output.LayerNorm.weight, layer.output.LayerNorm.bias
            out.append(lw)
        if dsts:
            torch._foreach_copy_(dsts, srcs)
    return out


_F32_OUT = [True]          # torch.bmm(..., out_dtype=float32) available (checked on first use)


def _splitk_wgrad(dy2: torch.Tensor, x2: torch.Tensor, keep_fp32: bool = False) -> torch.Tensor:
    """dW [N, K] = dy2^T [N, T] @ x2 [T, K] (returned in fp32 when ``keep_fp32``, else in dy2's dtype).  The reduction dim is the token count (10^4..10^5) while the
    output is only a few 256x256 tiles, so one hipBLASLt call leaves most CUs idle; cutting T into S
    slices (batched GEMM, fp32 partials) and adding them fills the chip (measured 1.5-2.7x on MI355X)."""
    t, n = dy2.shape
    k = x2.shape[1]
    tiles = ((n + 255) // 256) * ((k + 255) // 256)
    s = 1
    if tiles < 128 and t >= 4096:
        import math
        # measured on MI355X: 16 slices is within 5 % of the best split for every BERT shape once T >= 64k
        s = 16 if t >= 65536 else min(16, max(1, 2 ** round(math.log2(200.0 / tiles))))
    if s == 1:
        dw = dy2.t() @ x2
        return dw.float() if keep_fp32 else dw
    # a packed batch has an arbitrary token count: S equal slices of floor(T/S) rows + a tail of < S rows
    q = t // s
    a = dy2[:s * q].view(s, q, n).transpose(1, 2)
    b = x2[:s * q].view(s, q, k)
    if _F32_OUT[0] and dy2.dtype != torch.float32:
        try:
            dw = torch.bmm(a, b, out_dtype=torch.float32).sum(0)
        except (RuntimeError, TypeError):
            _F32_OUT[0] = False
            dw = torch.bmm(a, b).float().sum(0)
    else:
        dw = torch.bmm(a, b).float().sum(0)
    if s * q < t:
        dw += dy2[s * q:].t() @ x2[s * q:]         # < S rows: one tiny GEMM, added in fp32
    return dw if keep_fp32 else dw.to(dy2.dtype)


class _Linear(torch.autograd.Function):
    """y = x @ wc^T (+ bc) on hipBLASLt with the cached compute-dtype operands ``wc`` / ``bc``; the gradients go
    to the fp32 MASTER parameters (``masters`` = the weight(s) whose rows stack up to ``wc``, then the bias(es)
    stacking up to ``bc``): split-K weight gradient and the bias column sum stay in fp32 end to end.
    With ``residual`` the input is also returned as a second output for the residual branch, so that the two
    gradients of ``x`` meet HERE and the data-gradient GEMM accumulates onto the residual one (beta = 1 in the
    GEMM epilogue) instead of autograd running a separate add kernel."""

    @staticmethod
    def forward(ctx, x, wc, bc, residual, bias_grad, *masters):
        """``bias_grad`` False: ``bc`` is added but its masters are not among ``masters`` (their gradient is produced
        elsewhere: ops.AttentionQKV returns the fused QKV bias gradient itself, from inside its backward kernel)."""
        ctx.save_for_backward(x, wc)
        ctx.masters = masters
        ctx.has_bias = bc is not None and bias_grad
        ctx.n_w = len(masters) // 2 if ctx.has_bias else len(masters)
        y = torch.nn.functional.linear(x, wc, bc)
        return (y, x.view_as(x)) if residual else y

    @staticmethod
    def backward(ctx, dy, dxres=None):
        x, wc = ctx.saved_tensors
        masters, n_w = ctx.masters, ctx.n_w
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        dx = None
        if ctx.needs_input_grad[0]:
            if dxres is None:
                dx = (dy2 @ wc).view(x.shape)
            else:
                # the residual gradient buffer is ours alone (it was produced for this node): accumulate in place
                dres2 = dxres.reshape(-1, x.shape[-1])
                dx = (dres2.addmm_(dy2, wc) if dres2.is_contiguous() else torch.addmm(dres2, dy2, wc)).view(x.shape)
        grads = [None] * len(masters)
        need = ctx.needs_input_grad[5:]
        if any(need[:n_w]):
            dw = _splitk_wgrad(dy2, x2, keep_fp32=True)
            for i, (m, g) in enumerate(zip(masters[:n_w], dw.split([m.shape[0] for m in masters[:n_w]], 0))):
                if need[i]:
                    grads[i] = g if g.dtype == m.dtype else g.to(m.dtype)
        if ctx.has_bias and any(need[n_w:]):
            db = ops.column_sum(dy2) if dy2.is_cuda and dy2.dtype in (torch.float32, torch.bfloat16) else dy2.sum(0, dtype=torch.float32)
            for i, (m, g) in enumerate(zip(masters[n_w:], db.split([m.shape[0] for m in masters[n_w:]], 0))):
                if need[n_w + i]:
                    grads[n_w + i] = g if g.dtype == m.dtype else g.to(m.dtype)
        return (dx, None, None, None, None, *grads)


def _mm(x, wc, bc, masters, residual=False, bias_grad=True):
    with torch.autocast("cuda", enabled=False):
        return _Linear.apply(x, wc, bc, residual, bias_grad, *masters)


def _layer(lw, h, lens, heads, training, p_hidden, p_attn, eps, cu=None, max_len=0, pair_count=None, groups=None):
    pdim = h.shape[-1]
    d = pdim // heads
    scale = d ** -0.5
    if d in (64, 96):
        # the fused projection adds the Q|K|V biases; their GRADIENT (the column sums of dqkv) is returned by the attention
        # operator, which can form it inside its backward kernel (ops.AttentionQKV): an explicit autograd edge
        # bias master -> attention op, no side channel between the two backward functions
        qkv, h_res = _mm(h, lw.wqkv, lw.bqkv, lw.m_qkv[:3], residual=True, bias_grad=False)   # [B, L, 3P] or packed [T, 3P]
        ctx = ops.attention_qkv(qkv, None if cu is not None else lens, heads, scale, p_attn, training, cu, max_len, pair_count,
                                bias_masters=lw.m_qkv[3:], groups=groups)
    else:
        qkv, h_res = _mm(h, lw.wqkv, lw.bqkv, lw.m_qkv, residual=True)
        ctx = attention_any_dim(qkv[..., :pdim], qkv[..., pdim:2 * pdim], qkv[..., 2 * pdim:], lens, heads, scale, p_attn,
                                training)
    a = ops.bias_res_layernorm(_mm(ctx, lw.wo, None, lw.m_wo), lw.bo, h_res, lw.ln1w, lw.ln1b, eps, False, p_hidden, training)
    m_pre, a_res = _mm(a, lw.wi, None, lw.m_wi, residual=True)
    m = ops.bias_gelu(m_pre, lw.bi)
    return ops.bias_res_layernorm(_mm(m, lw.wo2, None, lw.m_wo2), lw.bo2, a_res, lw.ln2w, lw.ln2b, eps, False, p_hidden, training)


def bert_encode(plm, input_ids: torch.Tensor, lens: torch.Tensor, cd: torch.dtype, training: bool = False,
                gradient_checkpointing: bool = False, weights=None) -> torch.Tensor:
    """input_ids int [B, L] (padding anywhere past ``lens[b]`` is ignored), lens int32 [B] -> [B, L, P] in ``cd``.

    Token / position / type gathers are plain index gathers (bit-exact); padded key positions get
    probability 0 in every layer, padded query rows are computed but never read by the pooling.
    ``weights``: optional ``prepare_weights(plm, cd)`` result shared across micro-batches.
    """
    cfg = plm.config
    emb = plm.embeddings
    b, l = input_ids.shape
    x = torch.nn.functional.embedding(input_ids.long(), emb.word_embeddings.weight)   # bit-exact row gather
    x = x + (emb.token_type_embeddings.weight[0] + emb.position_embeddings.weight[:l]).unsqueeze(0)
    eps = cfg.layer_norm_eps
    p_hidden = cfg.hidden_dropout_prob
    p_attn = cfg.attention_probs_dropout_prob
    h = ops.bias_res_layernorm(x.to(cd), None, None, emb.LayerNorm.weight, emb.LayerNorm.bias, eps, False, p_hidden, training)
    heads = cfg.num_attention_heads
    if weights is None:
        weights = prepare_weights(plm, cd)
    for lw in weights:
        if gradient_checkpointing and training and torch.is_grad_enabled():
            h = checkpoint(_layer, lw, h, lens, heads, training, p_hidden, p_attn, eps, use_reentrant=False)
        else:
            h = _layer(lw, h, lens, heads, training, p_hidden, p_attn, eps)
    return h


def bert_encode_packed(plm, token_ids: torch.Tensor, pos_ids: torch.Tensor, cu_seqlens: torch.Tensor, max_len: int,
                       cd: torch.dtype, training: bool = False, gradient_checkpointing: bool = False, weights=None,
                       pair_count=None, groups=None):
    """Variable-length (packed) encoder pass: ``token_ids`` int [T] = the valid tokens of all sequences back to
    back, ``pos_ids`` int [T] = position of each token inside its sequence, ``cu_seqlens`` int32 [B+1].
    Returns [T, P].  Every GEMM / LayerNorm / GELU row is a real token and attention never sees padding
    (same result per token as the padded form: padded keys have probability exactly 0 there).
    ``groups``: optional ``ops.pack_sequence_groups`` boundaries (device int32) for the short-sequence attention kernels."""
    cfg = plm.config
    emb = plm.embeddings
    # word + type-0 + position rows in one pass (fp32 sums in torch's order, stored as cd); backward = segment sums by id
    x = ops.embed_sum(token_ids, pos_ids, emb.word_embeddings.weight, emb.position_embeddings.weight,
                      emb.token_type_embeddings.weight, cd)
    eps, p_hidden, p_attn = cfg.layer_norm_eps, cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob
    h = ops.bias_res_layernorm(x, None, None, emb.LayerNorm.weight, emb.LayerNorm.bias, eps, False, p_hidden, training)
    heads = cfg.num_attention_heads
    d = cfg.hidden_size // heads
    if d not in (64, 96):
        raise NotImplementedError("packed encoder pass needs head dim 64 or 96 (use the padded path otherwise)")
    if weights is None:
        weights = prepare_weights(plm, cd)
    for lw in weights:
        if gradient_checkpointing and training and torch.is_grad_enabled():
            h = checkpoint(_layer, lw, h, None, heads, training, p_hidden, p_attn, eps, cu_seqlens, max_len, pair_count, groups,
                           use_reentrant=False)
        else:
            h = _layer(lw, h, None, heads, training, p_hidden, p_attn, eps, cu_seqlens, max_len, pair_count, groups)
    return h


# ---------------------------------------------------------------------------------------------
# HF AttentionInterface registration (operator-level drop-in for K5)
# ---------------------------------------------------------------------------------------------
def gmlm_hip_attention_forward(module, query, key, value, attention_mask, dropout: float = 0.0,
                               scaling: Optional[float] = None, **kwargs):
    """Signature of transformers' attention functions (hf:modeling_bert.py:111-136,
    integrations/sdpa_attention.py:79).  query/key/value: [B, h, L, d].  ``attention_mask`` must be a
    key-padding mask (bool or additive [B,1,1|Lq,Lk]) whose valid keys form a prefix, which is what
    tokenizer right-padding produces; returns ([B, Lq, h, d], None)."""
    b, h, lq, d = query.shape
    lk = key.shape[2]
    if scaling is None:
        scaling = d ** -0.5
    kv_len = None
    if attention_mask is not None:
        m = attention_mask
        keep = m if m.dtype == torch.bool else (m > -1.0)
        keep = keep.reshape(b, -1, lk)[:, 0, :]
        kv_len = keep.sum(-1).to(torch.int32)
        # prefix check is cheap and guards against silently mis-handling arbitrary masks
        if not bool((keep == (torch.arange(lk, device=keep.device)[None] < kv_len[:, None])).all()):
            raise NotImplementedError("gmlm_hip attention supports right-padded key masks only")
    to_rows = lambda t: t.transpose(1, 2).reshape(b, t.shape[2], h * d).contiguous()
    q2, k2, v2 = to_rows(query), to_rows(key), to_rows(value)
    if q2.dtype not in (torch.float32, torch.bfloat16):
        q2, k2, v2 = q2.bfloat16(), k2.bfloat16(), v2.bfloat16()
    out = attention_any_dim(q2, k2, v2, kv_len, h, scaling, dropout, dropout > 0)
    return out.reshape(b, lq, h, d).to(query.dtype), None


def register_hf_attention(name: str = "gmlm_hip") -> str:
    from transformers import AttentionInterface
    AttentionInterface.register(name, gmlm_hip_attention_forward)
    try:  # same mask builder as sdpa (boolean / additive key-padding mask)
        from transformers.masking_utils import AttentionMaskInterface, sdpa_mask
        AttentionMaskInterface.register(name, sdpa_mask)
    except Exception:  # older/newer transformers without the mask registry: eager mask is used
        pass
    return name
