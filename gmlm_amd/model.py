"""``GraphTextLM`` — drop-in for main.py:182-372 on the gfx950 kernel path.

Same constructor arguments, attribute names (load-bearing for ``setup_optimizer`` /
``pretrain_contrastive_gnn``, main.py:379-390, 409-427), state-dict keys, ``get_graph_embeddings`` and
``forward`` signatures.  What differs is how it is computed:

* edge types + relation CSR are built once per graph on the GPU (K1) instead of a per-edge Python loop;
* each RGCN block = K2 aggregation -> one hipBLASLt GEMM per operand -> K4 GraphNorm+GELU+dropout;
* the PLM runs on K5/K6 (gmlm_amd.bert) from pre-tokenised ids cached on the device; pooling +
  scatter is K8;
* both CrossAttention modules use the streaming attention kernel K7 (no N x N score matrix);
* fusion LayerNorm+GELU and classifier GELU are fused epilogues (K6 / bias_gelu).
"""
from __future__ import annotations

import logging
from typing import Optional, Sequence

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from . import bert, ops
from .graph import GraphCache, RelCSR
from .nn import (SPLITK_ROW_QUANTUM, CrossAttention, GraphNorm, MultiScaleFusion, RGCNConv, _linear, compute_dtype,
                 cross_attention_specs, linear_specs, shadow_params, use_shadow)

logger = logging.getLogger(__name__)

# static sizes of a captured text-encoder batch: sequences / tokens / attention work items are padded up to multiples of these
ENCODER_BUCKET = (64, 2048, 32)


def bucketed_layout(lens: Sequence[int], cap: int, short_groups: bool, quanta=None):
    """Pad a packed batch of ``lens`` (host ints, every one <= cap) to STATIC sizes, so that batches with different active sets
    share one hipGraph per size bucket (gmlm_amd/graphs.py): dummy sequences of [PAD] tokens are appended until the sequence
    count is a multiple of 64 AND the token count a multiple of 2,048.  A dummy is an ordinary sequence of 1..cap rows to every
    kernel; it is pooled into a sink row behind the node table, so it reaches no output and gets a zero gradient.
    Returns (all lengths, S_b, T_b, attention work-item boundaries int32 [G_b + 1] or None)."""
    sq, tq, gq = quanta or ENCODER_BUCKET
    cap = max(int(cap), 16)
    n_real, total = len(lens), int(sum(lens))
    k_min = -(-(tq - 1) // (cap - 1)) + 1              # dummies needed in the worst case: k + tq - 1 pad tokens in k sequences of <= cap
    s_b = -(-(n_real + k_min) // sq) * sq
    k = s_b - n_real
    t_b = -(-(total + k) // tq) * tq                   # every dummy holds at least one token
    pad = t_b - total
    base, rem = divmod(pad, k)
    assert base + (1 if rem else 0) <= cap and base >= 1
    lens_all = [int(v) for v in lens] + [base + 1] * rem + [base] * (k - rem)
    if not short_groups:
        return lens_all, s_b, t_b, None
    b = ops.pack_sequence_groups(lens_all).tolist()
    g_b = min(-(-(len(b) - 1) // gq) * gq, s_b)
    need = g_b - (len(b) - 1)
    while need > 0:                                    # split work items until their count is the bucket's (a single sequence is the floor)
        out = [b[0]]
        for lo, hi in zip(b[:-1], b[1:]):
            if need > 0 and hi - lo >= 2:
                out.append(lo + (hi - lo) // 2)
                need -= 1
            out.append(hi)
        b = out
    return lens_all, s_b, t_b, torch.tensor(b, dtype=torch.int32)


class TokenizedTexts:
    """Device-resident tokenisation of all node texts (replaces per-step host tokenisation and the
    per-node ``.item()`` of main.py:338-345).  ``input_ids`` int32 [N, Lmax], ``lens`` int32 [N]."""

    def __init__(self, input_ids: torch.Tensor, lens: torch.Tensor):
        self.input_ids = input_ids.to(torch.int32).contiguous()
        self.lens = lens.to(torch.int32).contiguous()
        self.lens_host = self.lens.cpu()
        self._ids_checked = None          # (vocab, max positions) the ids were validated against (ops.check_embed_ids)
        self._source = None               # the text list these ids were made from (GraphTextLM.tokenize)

    def check_ids(self, vocab: int, npos: int) -> None:
        """Raise like F.embedding / HF BERT would on an out-of-range token id or an over-long sequence; once per table size."""
        if self._ids_checked != (vocab, npos):
            ops.check_embed_ids(self.input_ids, self.lens, vocab, npos)
            self._ids_checked = (vocab, npos)

    @classmethod
    def from_mask(cls, input_ids: torch.Tensor, attention_mask: torch.Tensor):
        return cls(input_ids, attention_mask.sum(-1))

    def to(self, device):
        return TokenizedTexts(self.input_ids.to(device), self.lens.to(device))


class GraphTextLM(nn.Module):
    def __init__(self, gnn_in_channels, hidden_channels, num_classes, num_relations=5, num_bases=30, dropout_rate=0.3,
                 model_name='thenlper/gte-base', plm_max_length=256, *, plm_encoder=None, plm_tokenizer=None,
                 compute_dtype: Optional[torch.dtype] = None, activation_checkpointing: bool = False,
                 plm_gradient_checkpointing: bool = False):
        super().__init__()
        self.gnn_mask_token_embed = nn.Parameter(torch.zeros(1, gnn_in_channels))
        nn.init.xavier_uniform_(self.gnn_mask_token_embed)
        hc = hidden_channels
        self.rgcn1 = RGCNConv(gnn_in_channels, hc, num_relations=num_relations, num_bases=num_bases)
        self.gnorm1 = GraphNorm(hc)
        self.dropout1 = nn.Dropout(dropout_rate)
        self.rgcn2 = RGCNConv(hc, hc * 2, num_relations=num_relations, num_bases=num_bases)
        self.gnorm2 = GraphNorm(hc * 2)
        self.dropout2 = nn.Dropout(dropout_rate)
        self.rgcn3 = RGCNConv(hc * 2, hc * 4, num_relations=num_relations, num_bases=num_bases)
        self.gnorm3 = GraphNorm(hc * 4)
        self.dropout3 = nn.Dropout(dropout_rate)
        self.rgcn4 = RGCNConv(hc * 4, hc * 8, num_relations=num_relations, num_bases=num_bases)
        self.gnorm4 = GraphNorm(hc * 8)
        self.dropout4 = nn.Dropout(dropout_rate)
        self.residual_proj1 = nn.Linear(gnn_in_channels, hc)
        self.residual_proj2 = nn.Linear(hc, hc * 2)
        self.residual_proj3 = nn.Linear(hc * 2, hc * 8)   # dead branch in the reference (main.py:317-318); kept for the state dict

        if plm_encoder is None:
            from transformers import AutoModel, AutoTokenizer
            logger.info("Loading HuggingFace model: %s", model_name)
            plm_encoder = AutoModel.from_pretrained(model_name, trust_remote_code=True)
            if plm_tokenizer is None:
                plm_tokenizer = AutoTokenizer.from_pretrained(model_name, trust_remote_code=True)
        bert.check_supported(plm_encoder)
        self.plm_encoder = plm_encoder
        self.plm_tokenizer = plm_tokenizer
        self.plm_max_length = plm_max_length
        p = self.plm_encoder.config.hidden_size

        self.multi_scale_fusion = MultiScaleFusion([hc, hc * 2, hc * 4, hc * 8], p)
        self.graph_to_text_attn = CrossAttention(p, num_heads=8, dropout=dropout_rate)
        self.text_to_graph_attn = CrossAttention(p, num_heads=8, dropout=dropout_rate)
        self.fusion_network = nn.Sequential(nn.Linear(p * 2, p), nn.LayerNorm(p), nn.GELU(), nn.Dropout(dropout_rate))
        self.classifier = nn.Sequential(nn.Linear(p, hc), nn.GELU(), nn.Dropout(dropout_rate), nn.Linear(hc, num_classes))

        self.num_relations = num_relations
        self.compute_dtype = compute_dtype
        self.activation_checkpointing = activation_checkpointing
        self.plm_gradient_checkpointing = plm_gradient_checkpointing
        self.plm_packed = True
        # pad every packed text batch to the static sizes of ``bucketed_layout`` (what a captured encoder replays; set it on an
        # eager model to run the very same padded batch without hipGraphs)
        self.plm_bucketed = False
        # variable-length token packing in the text encoder (head dim 64 / 96)
        self.active_index = None   # set by encode_texts: ascending device index of the active nodes of the last call
        self._graphs = GraphCache(capacity=4)
        self._tokens = {}
        self.dist = None          # gmlm_amd.dist.PartitionContext for the 1-D node partition (None = single GPU)
        # run the text encoder on a second HIP stream beside the GNN (single GPU, eager path); results are identical, only
        # the order in which independent kernels reach the device changes.  Opt-in: two streams of library GEMMs are a
        # configuration the GEMM library is rarely run in ("streams")
        self.overlap_streams = False
        self._side_stream = None
        self._gnn_shadow = None   # nn.ParamShadow of the last get_graph_embeddings call (compute-dtype operands of its dense layers)
        self._branch_stream = None  # side stream for the second cross-attention while a whole-step hipGraph is recorded (graphs.py)
        self._active_seen = None  # host copy / index tables of the last active-node mask TENSOR (reused while it is not written to)
        self._graphed = None      # gmlm_amd.graphs.GraphedStep: hipGraph recording of the GNN + head regions (capture_hip_graphs)

    # ------------------------------------------------------------------------------------------
    def _cd(self) -> torch.dtype:
        return compute_dtype(self.compute_dtype)

    def graph(self, edge_index: torch.Tensor, num_nodes: int, edge_type=None) -> RelCSR:
        if self.dist is not None:
            return self.dist.csr
        return self._graphs.get(edge_index, num_nodes, self.num_relations, edge_type)

    def soft_mask_input(self, x, mask, beta=0.7):
        """Fused soft masking (K9) straight into the compute dtype with 16-byte aligned rows."""
        cd = self._cd()
        align = 8 if cd == torch.bfloat16 else 4
        cols = (x.shape[1] + align - 1) // align * align
        return ops.soft_masking_gnn_input(x, mask, self.gnn_mask_token_embed, beta, cd, cols)

    def _block(self, k: int, x: torch.Tensor, csr: RelCSR) -> torch.Tensor:
        conv, norm, drop = getattr(self, f"rgcn{k}"), getattr(self, f"gnorm{k}"), getattr(self, f"dropout{k}")
        if self.dist is not None:
            x = self.dist.with_halo(x, defer=True)                    # [n_local + n_halo, F]; halo rows land under the root GEMM
        with use_shadow(self._gnn_shadow):                            # (no-op inside get_graph_embeddings; a checkpoint recompute needs it)
            z = conv.forward_csr(x, csr, self.dist.all_reduce_sum if self.dist is not None else None,
                                 self.dist.halo_ready if self.dist is not None else None)            # [n, out] in cd
        cd = x.dtype
        n_total = self.dist.n_total if self.dist is not None else z.size(0)
        if n_total > 1:                                               # main.py:273 guard
            reducer = self.dist.all_reduce_sum if self.dist is not None else None
            return norm(z, act=True, dropout_p=drop.p, out_dtype=cd, reducer=reducer, n_total=n_total)
        return ops.bias_gelu(z, None, drop.p, self.training)

    def get_graph_embeddings(self, x_feat, edge_index, edge_type=None):
        csr = self.graph(edge_index, x_feat.size(0), edge_type)      # cache keyed by the caller's tensor; the int64 cast (main.py:323) happens inside the build
        cd = self._cd()
        align = 8 if cd == torch.bfloat16 else 4
        f_in = self.rgcn1.in_channels
        if x_feat.shape[1] == f_in:
            pad = (-f_in) % align
            x0 = x_feat.to(cd)
            if pad:
                x0 = torch.nn.functional.pad(x0, (0, pad))
        else:                                                          # already padded by soft_mask_input
            x0 = x_feat.to(cd)
        x0 = x0.contiguous()
        run = (lambda k, x: checkpoint(self._block, k, x, csr, use_reentrant=False)) \
            if (self.activation_checkpointing and self.training and torch.is_grad_enabled()) else \
            (lambda k, x: self._block(k, x, csr))
        # compute-dtype operands of this region's dense layers: one flat buffer, one multi-tensor copy (nn.ParamShadow); kept on
        # the module so that a checkpoint recompute of a block (which runs in backward, outside this call) finds the same operands
        specs = linear_specs(self.residual_proj1, self.residual_proj2)
        for k in (1, 2, 3, 4):
            conv = getattr(self, f"rgcn{k}")
            specs += [(id(conv.root), [conv.root], x0.shape[1] - f_in if k == 1 else 0), (id(conv.bias), [conv.bias], 0)]
        with shadow_params(specs, cd, x0.device) as sh:
            self._gnn_shadow = sh
            return self._graph_embeddings_body(x0, f_in, cd, run)

    def _graph_embeddings_body(self, x0, f_in, cd, run):
        e1 = run(1, x0)
        # (a sum of two compute-dtype tensors is formed in fp32 and rounded once - the same value as adding their fp32 copies
        # and casting, without the three cast passes)
        x1 = e1 + _linear(x0[:, :f_in] if x0.shape[1] != f_in else x0, self.residual_proj1.weight, self.residual_proj1.bias)
        e2 = run(2, x1)
        x2 = e2 + _linear(x1, self.residual_proj2.weight, self.residual_proj2.bias)
        e3 = run(3, x2)
        e4 = run(4, e3)
        # main.py:317-318 computes x4 + residual_proj3(x2) and discards it: skipped (no effect on outputs or grads)
        self.multi_scale_fusion.compute_dtype = cd
        return self.multi_scale_fusion([e1, e2, e3, e4])

    # ------------------------------------------------------------------------------------------
    def tokenize(self, all_node_texts) -> TokenizedTexts:
        """Tokenise every node text once on the host (same tokenizer call as main.py:342-345) and keep
        the ids on the device.  The cache is keyed by the identity of the text list, its length and a 65-string sample of
        its content, so a list that is appended to, truncated or rewritten is re-tokenised; a caller that edits single
        strings of the same list in place between steps calls ``clear_caches()`` (the reference re-tokenises every step,
        main.py:342: its texts are read-only during training)."""
        if isinstance(all_node_texts, TokenizedTexts):
            return all_node_texts
        # Per step the check is O(1) + a fixed-size sample: same list object, same length, same strings at 64 evenly spaced
        # positions (compared by identity first: an unchanged list holds the very same str objects).  The full content hash
        # (O(N): ~17 ms at arxiv size, ~1 s at 10M nodes) is taken only when the list is first seen or the sample moved.
        n = len(all_node_texts)
        hit = self._tokens.get(id(all_node_texts))
        if hit is not None and hit[1] is all_node_texts and hit[2] == n:
            probe = hit[3]
            if all(all_node_texts[i] is t or all_node_texts[i] == t for i, t in probe):
                return hit[0]
        enc = self.plm_tokenizer(list(all_node_texts), padding=True, truncation=True, max_length=self.plm_max_length,
                                 return_tensors="pt")
        dev = self.gnn_mask_token_embed.device
        tt = TokenizedTexts.from_mask(enc["input_ids"].to(dev), enc["attention_mask"].to(dev))
        step = max(1, n // 64)
        probe = [(i, all_node_texts[i]) for i in range(0, n, step)][:64] + ([(n - 1, all_node_texts[n - 1])] if n else [])
        self._tokens = {id(all_node_texts): (tt, all_node_texts, n, probe)}
        return tt

    def clear_caches(self):
        """Drop the cached graph preprocessing (CSR) and tokenisation."""
        self._graphs.clear()
        self._tokens = {}
        self._active_seen = None

    def start_mask_copy(self, node_mask: torch.Tensor):
        """Begin the device -> pinned-host copy of the active-node mask and mark its completion with an event.
        The host needs the active set (count, token lengths) to lay out the packed PLM batch; waiting for THIS
        event instead of reading the mask later means the host never waits for the GNN kernels enqueued in
        between, and the PLM launches follow the GNN's without an idle gap (1 ms per step at Squirrel size)."""
        if not node_mask.is_cuda:
            return node_mask, None
        seen = self._active_seen
        if seen is not None and seen["mask"] is node_mask and seen["version"] == node_mask._version:
            # the very tensor of the last call, not written since (a fixed train / validation mask): its host copy is still
            # right, so the host neither copies nor waits and keeps running ahead of the device
            return seen["mask_h"], None
        buf = torch.empty(node_mask.numel(), dtype=torch.bool, pin_memory=True)
        buf.copy_(node_mask.reshape(-1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return buf, ev

    def _active_set(self, node_mask: torch.Tensor, mask_copy=None) -> dict:
        """Host view of the active-node mask: waits for the mask copy (``start_mask_copy``) unless this very tensor was seen
        last call and has not been written since; caches what is derived from it (index tables of the packed text batch)."""
        mask_h, ev = mask_copy if mask_copy is not None else self.start_mask_copy(node_mask)
        if ev is not None:
            ev.synchronize()                                          # waits for the mask copy only
        seen = self._active_seen
        if not (node_mask.is_cuda and seen is not None and seen["mask_h"] is mask_h):
            seen = dict(mask=node_mask, version=node_mask._version, mask_h=mask_h, idx_h=mask_h.nonzero(as_tuple=True)[0],
                        active_index=None, bucket=None)
            self._active_seen = seen if node_mask.is_cuda else None
        return seen

    def _bucket_tables(self, tokens: TokenizedTexts, seen: dict, cd):
        """Index tables of the ONE bucket-padded packed batch that holds every active node (``bucketed_layout``):
        (key, device tables, short-sequence attention used)."""
        hit = seen["bucket"]
        if hit is not None and hit[0] is tokens and hit[1] == cd:
            return hit[2:]                                            # same mask tensor, same token set: same tables
        idx_h = seen["idx_h"]
        a, n = idx_h.numel(), seen["mask_h"].numel()
        dev = tokens.input_ids.device
        cfg = self.plm_encoder.config
        heads, p = cfg.num_attention_heads, cfg.hidden_size
        lens_h = tokens.lens_host[idx_h]
        order = torch.argsort(lens_h, descending=True, stable=True)   # long sequences first: fuller attention work items
        cap = int(tokens.lens_host.max())
        short = cd == torch.bfloat16 and p // heads == 64 and cap <= 128
        lens_all, s_b, t_b, groups_h = bucketed_layout(lens_h[order].tolist(), cap, short)
        short = short and s_b * heads >= 512
        bi_h = torch.full((s_b,), n, dtype=torch.long)
        bi_h[:a] = idx_h[order]
        la_h = torch.tensor(lens_all, dtype=torch.int32)
        cu_h = torch.zeros(s_b + 1, dtype=torch.int32)
        cu_h[1:] = torch.cumsum(la_h, 0)

        def to_dev(t):
            return t.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else t.to(dev)

        args = (to_dev(la_h), to_dev(bi_h), to_dev(cu_h)) + ((to_dev(groups_h),) if short else ())
        key = (id(tokens), s_b, t_b, groups_h.numel() if short else 0, max(cap, 16), n)
        seen["bucket"] = (tokens, cd, key, args, short)
        return key, args, short

    def encode_texts(self, tokens: TokenizedTexts, node_mask: torch.Tensor, plm_batch_size: int = 8,
                     _mask_copy=None, _weights=None) -> torch.Tensor:
        """main.py:328-358: PLM over the active nodes in micro-batches, masked mean pool, row scatter."""
        n = node_mask.numel()
        dev = node_mask.device
        p = self.plm_encoder.config.hidden_size
        plm_embeds = torch.zeros(n, p, device=dev)
        seen = self._active_set(node_mask, _mask_copy)
        idx_h = seen["idx_h"]                                         # host: ascending active node ids
        a = idx_h.numel()
        self.active_index = None
        if a == 0:
            return plm_embeds

        def to_dev(t):                                                # pinned staging: H2D copies that do not stall the host
            return t.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else t.to(dev)

        if seen["active_index"] is None:
            seen["active_index"] = to_dev(idx_h)
        self.active_index = seen["active_index"]                      # for the caller's loss gather (no mask indexing sync)
        cd = self._cd()
        ecfg = self.plm_encoder.config
        tokens.check_ids(ecfg.vocab_size, ecfg.max_position_embeddings)      # one sync per token set, not per step
        grad = self.plm_encoder.training or self.training
        heads = self.plm_encoder.config.num_attention_heads
        packed = (p // heads) in (64, 96) and self.plm_packed
        plm_batch_size = max(1, min(int(plm_batch_size), 65535 // heads))   # attention grids index (sequence, head) in 16 bits
        with torch.set_grad_enabled(grad and torch.is_grad_enabled()):
            # cast / fuse once, share across micro-batches (forward() has done it ahead of the GNN launches)
            weights = _weights
            g = self._graphed if (grad and torch.is_grad_enabled() and self.training and self.dist is None) else None
            if g is not None and not g.encoder_enabled:
                g = None
            if packed and a <= plm_batch_size and (self.plm_bucketed or g is not None):
                # ONE micro-batch padded to bucket sizes: static shapes, replayable (graphs.py); the same function runs eagerly
                key, args, short = self._bucket_tables(tokens, seen, cd)
                if g is not None:
                    return g.encoder(self, key, tokens, args)
                return self.encode_packed_static(tokens, key, *args, *(() if short else (None,)), weights)
            if weights is None:
                weights = bert.prepare_weights(self.plm_encoder, cd)
            lens_h = tokens.lens_host[idx_h]
            # length-bucketed micro-batches: every active node is encoded independently and scattered by its own
            # index, so the order is free; sorting by token count keeps the padding of each micro-batch small
            order = torch.argsort(lens_h, descending=True, stable=True)
            lens_h = lens_h[order]
            idx = to_dev(idx_h[order])
            for s in range(0, a, plm_batch_size):
                bi = idx[s:s + plm_batch_size]
                lh = lens_h[s:s + plm_batch_size]
                lmax = max(int(lh.max()), 1)                              # host-side: no sync
                lens = tokens.lens[bi]
                if packed:
                    # variable-length packing: only real tokens exist; index arithmetic from the HOST copy of the
                    # lengths (sizes known without a device sync), token gather on the device
                    total = int(lh.sum())
                    # large batches: pad the token count to a multiple of 16 (of 112 = lcm(14, 16) once the weight-gradient GEMMs
                    # slice by 14 or 16: nn._splitk_wgrad) with ONE dummy sequence of [PAD] tokens behind the real ones (never
                    # pooled, zero gradient): the split-K weight-gradient GEMMs then cut T into equal slices with no tail rows
                    # (a tail is one more tiny GEMM and a full pass over dW per weight: 2 x 48 launches per BERT-base step),
                    # and every GEMM sees an aligned row count
                    quantum = SPLITK_ROW_QUANTUM if (total >= 65536 and lmax >= SPLITK_ROW_QUANTUM) else 16
                    pad = (-total) % quantum if (total >= 4096 and lmax >= 16) else 0
                    cu_h = torch.zeros(bi.numel() + 1 + (1 if pad else 0), dtype=torch.int32)
                    cu_h[1:bi.numel() + 1] = torch.cumsum(lh, 0)
                    if pad:
                        cu_h[-1] = total + pad
                    cu_all = to_dev(cu_h)
                    cu = cu_all[:bi.numel() + 1]
                    seq = torch.repeat_interleave(torch.arange(bi.numel(), device=dev), lens.long(), output_size=total)
                    pos = torch.arange(total, device=dev) - cu[seq].long()
                    tok = tokens.input_ids[bi[seq], pos]
                    if pad:
                        tok = torch.cat([tok, tok.new_zeros(pad)])
                        pos = torch.cat([pos, torch.arange(pad, device=dev)])
                    # short sequences (the reference tokenises to <= 128 tokens, main.py:340): consecutive sequences are packed
                    # into attention work items of <= 128 rows (host arithmetic on the host copy of the lengths, ~0.2 ms per
                    # 1,000 sequences, once per batch: all layers, forward and backward, share it)
                    groups = None
                    n_seq = bi.numel() + (1 if pad else 0)
                    if cd == torch.bfloat16 and p // heads == 64 and lmax <= 128 and n_seq * heads >= 512:
                        groups = to_dev(ops.pack_sequence_groups(lh.tolist() + ([pad] if pad else [])))
                    hs = bert.bert_encode_packed(self.plm_encoder, tok, pos, cu_all, lmax, cd, self.plm_encoder.training,
                                                 self.plm_gradient_checkpointing, weights,
                                                 pair_count=float((lh.double() ** 2).sum()) + float(pad * pad), groups=groups)
                    plm_embeds = ops.MeanPoolScatter.apply(plm_embeds, hs, None, bi, cu, total)
                else:
                    ids = tokens.input_ids[bi, :lmax]
                    hs = bert.bert_encode(self.plm_encoder, ids, lens, cd, self.plm_encoder.training,
                                          self.plm_gradient_checkpointing, weights)
                    plm_embeds = ops.MeanPoolScatter.apply(plm_embeds, hs, lens, bi)
        return plm_embeds

    def encode_packed_static(self, tokens: TokenizedTexts, key, lens_all, node_of_seq, cu_all, groups=None, weights=None) -> torch.Tensor:
        """The text-encoder pass over one bucket-padded packed batch: every shape below follows from ``key`` = (token set,
        S_b, T_b, G_b + 1, max_len, N), every index is computed on the device from the three small tables, so the function
        can be recorded once per bucket and replayed with other tables.  ``node_of_seq`` = N marks a dummy sequence
        (``bucketed_layout``): its tokens are [PAD], its pooled row is the sink row N that is cut off."""
        _, s_b, t_b, _, lmax, n = key
        dev = lens_all.device
        p = self.plm_encoder.config.hidden_size
        seq = torch.repeat_interleave(torch.arange(s_b, device=dev), lens_all.long(), output_size=t_b)
        pos = torch.arange(t_b, device=dev) - cu_all[seq].long()
        node = node_of_seq[seq]
        ids = tokens.input_ids
        tok = torch.where(node < n, ids[node.clamp(max=n - 1), pos.clamp(max=ids.shape[1] - 1)], 0)
        hs = bert.bert_encode_packed(self.plm_encoder, tok, pos, cu_all, lmax, self._cd(), self.plm_encoder.training,
                                     self.plm_gradient_checkpointing, weights, pair_count=float(t_b) * lmax / 2, groups=groups)
        out = torch.zeros(n + 1, p, device=dev)
        return ops.MeanPoolScatter.apply(out, hs, None, node_of_seq, cu_all, t_b)[:n]

    def forward(self, gnn_input_features, edge_index, all_node_texts, text_processing_node_mask, edge_type=None,
                plm_batch_size=8):
        mask_copy = self.start_mask_copy(text_processing_node_mask)                              # async; consumed below
        # the per-step compute-dtype copy of the PLM weights does not depend on the mask: its host-side set-up
        # (0.6 ms) runs here, under device work that is still queued, not between the GNN and the PLM launches
        g = self._graphed
        replay = (g is not None and self.training and torch.is_grad_enabled() and edge_type is None and self.dist is None
                  and g.matches(gnn_input_features, edge_index))
        if replay and g.whole_step:
            # ONE recording of the whole forward (and of its backward) per size bucket of the text batch, with the text encoder
            # and the GNN on two branches of the graph (graphs.py).  The host needs the active set first: it waits for the mask
            # copy here, ahead of every launch of the step - unless this mask tensor is the one of the last call, unwritten.
            tokens = self.tokenize(all_node_texts)
            seen = self._active_set(text_processing_node_mask, mask_copy)
            heads = self.plm_encoder.config.num_attention_heads
            a = seen["idx_h"].numel()
            if (0 < a <= max(1, min(int(plm_batch_size), 65535 // heads)) and self.plm_packed
                    and (self.plm_encoder.config.hidden_size // heads) in (64, 96)):
                if seen["active_index"] is None:
                    seen["active_index"] = seen["idx_h"].pin_memory().to(gnn_input_features.device, non_blocking=True)
                self.active_index = seen["active_index"]
                ecfg = self.plm_encoder.config
                tokens.check_ids(ecfg.vocab_size, ecfg.max_position_embeddings)
                key, args, _ = self._bucket_tables(tokens, seen, self._cd())
                return g.step(self, key, tokens, gnn_input_features, args)
            mask_copy = (seen["mask_h"], None)                        # already on the host
        if self.overlap_streams and not replay and self.dist is None and gnn_input_features.is_cuda:
            # GNN and text encoder are independent until the head: the encoder goes to a second HIP stream, so the GNN's
            # HBM-bound kernels (basis composition, GraphNorm, aggregation) run BESIDE the encoder's GEMMs instead of before
            # them; autograd replays every backward node on its forward stream, so the backward overlaps the same way.  The
            # side stream is issued last: its backward is then queued first (the long one), the GNN's next to it.
            cur = torch.cuda.current_stream()
            side = self._second_stream(gnn_input_features.device)
            side.wait_stream(cur)                                     # fork
            gnn_embeds = self.get_graph_embeddings(gnn_input_features, edge_index, edge_type)
            tokens = self.tokenize(all_node_texts)
            with torch.cuda.stream(side):
                weights = bert.prepare_weights(self.plm_encoder, self._cd())
                plm_embeds = self.encode_texts(tokens, text_processing_node_mask, plm_batch_size, mask_copy, weights)
            cur.wait_stream(side)                                     # join
            plm_embeds.record_stream(cur)                             # allocated on `side`, read (and saved for backward) on `cur`
            if self.active_index is not None:
                self.active_index.record_stream(cur)
            return self.head(gnn_embeds, plm_embeds)
        if replay and g.concurrent:
            # the GNN recording and the encoder recording (or the eager encoder, when the batch needs several micro-batches)
            # replay on two streams side by side; each recording is the linear one, the head's follows the join
            cur = torch.cuda.current_stream()
            side = g._side
            g.counter.add_(1)                                         # the step's dropout seed, ahead of the fork
            side.wait_stream(cur)                                     # fork
            gnn_embeds = g.gnn(gnn_input_features)
            tokens = self.tokenize(all_node_texts)
            with torch.cuda.stream(side):                             # issued last: its backward is queued first
                plm_embeds = self.encode_texts(tokens, text_processing_node_mask, plm_batch_size, mask_copy, None)
            cur.wait_stream(side)                                     # join
            plm_embeds.record_stream(cur)
            if self.active_index is not None:
                self.active_index.record_stream(cur)
            return g.head(gnn_embeds, plm_embeds)
        # (a recorded encoder casts the weights inside its own graph)
        weights = None if (replay and g.encoder_enabled) else bert.prepare_weights(self.plm_encoder, self._cd())
        gnn_embeds = g.gnn(gnn_input_features) if replay else \
            self.get_graph_embeddings(gnn_input_features, edge_index, edge_type)               # fp32 [N, P]
        tokens = self.tokenize(all_node_texts)
        plm_embeds = self.encode_texts(tokens, text_processing_node_mask, plm_batch_size, mask_copy, weights)   # fp32 [N, P]
        return g.head(gnn_embeds, plm_embeds) if replay else self.head(gnn_embeds, plm_embeds)

    def _second_stream(self, device):
        if self._side_stream is None or self._side_stream.device != device:
            self._side_stream = torch.cuda.Stream(device=device)
        return self._side_stream

    def capture_hip_graphs(self, gnn_input_sample: torch.Tensor, edge_index: torch.Tensor, encoder: bool = True,
                           whole_step: bool = True, concurrent: bool = False):
        """Record the static-shape regions of the training step (GNN blocks + fusion; cross-attention + head) as hipGraphs
        for THIS input shape and ``edge_index`` tensor; ``forward`` then replays them (training mode, same shape, same
        edge tensor) and runs eagerly otherwise.  For the launch-bound small configurations (gmlm_amd/graphs.py)."""
        from . import graphs
        return graphs.capture(self, gnn_input_sample, edge_index, encoder=encoder, whole_step=whole_step, concurrent=concurrent)

    def release_hip_graphs(self):
        self._graphed = None

    def head(self, gnn_embeds, plm_embeds):
        """main.py:360-372."""
        cd = self._cd()
        fn, c = self.fusion_network, self.classifier
        specs = (cross_attention_specs(self.graph_to_text_attn) + cross_attention_specs(self.text_to_graph_attn)
                 + linear_specs(c[3]) + [(id(fn[0].weight), [fn[0].weight], 0), (id(c[0].weight), [c[0].weight], 0)])
        with shadow_params(specs, cd, gnn_embeds.device) as sh:
            if sh is not None and self._branch_stream is not None and self.dist is None:
                sh.record_stream(self._branch_stream)              # the second cross-attention reads its operands on the branch
            return self._head_body(gnn_embeds, plm_embeds, cd)

    def _head_body(self, gnn_embeds, plm_embeds, cd):
        g = gnn_embeds.unsqueeze(0)
        t = plm_embeds.unsqueeze(0)
        ring = self.dist.ring_attention if (self.dist is not None and self.dist.use_ring) else None
        gather = self.dist.all_gather_rows if (self.dist is not None and ring is None) else None
        self.graph_to_text_attn.compute_dtype = self.text_to_graph_attn.compute_dtype = cd
        bs = self._branch_stream if self.dist is None else None
        if bs is not None:
            # whole-step recording (graphs.py): the two cross-attentions are independent - the second one goes to a branch of
            # the graph.  The fork is taken BEFORE the first module's launches; the branch is issued last, so that its backward
            # is queued first (see GraphedStep.step)
            cur = torch.cuda.current_stream()
            bs.wait_stream(cur)
        gnn_attended = self.graph_to_text_attn(g, t, gather, ring)
        if bs is not None:
            with torch.cuda.stream(bs):
                text_attended = self.text_to_graph_attn(t, g, gather, ring)
            cur.wait_stream(bs)
            # blocks read or written on a stream other than the one that allocated them (the module's saved inputs are read by
            # its backward on `bs`): keep the allocator from recycling them on their own stream while the other still runs
            gnn_embeds.record_stream(bs)
            plm_embeds.record_stream(bs)
            text_attended.record_stream(cur)
        else:
            text_attended = self.text_to_graph_attn(t, g, gather, ring)
        fn = self.fusion_network
        # Linear(2P -> P) over [gnn_attended | text_attended] (main.py:366-367): one concatenation of the two [N, P] activations,
        # ONE GEMM with K = 2P (one rounding of the sum; two half GEMMs + an add rounded three times)
        fused = _linear(torch.cat([gnn_attended, text_attended], -1), fn[0].weight)
        fused = ops.bias_res_layernorm(fused, fn[0].bias, None, fn[1].weight, fn[1].bias, fn[1].eps, True, fn[3].p,
                                       self.training)
        fused = fused.squeeze(0)
        c = self.classifier
        hcls = ops.bias_gelu(_linear(fused, c[0].weight), c[0].bias, c[2].p, self.training)
        return _linear(hcls, c[3].weight, c[3].bias).float()
