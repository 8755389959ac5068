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
from .nn import _Linear, _linear, _mm, _splitk_wgrad, attention_any_dim  # noqa: F401  (dense-layer functions live in nn.py)


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
