"""Autograd operators over libgmlm_hip.so.  Every op is the HIP path; there is no eager fallback.

Tensors are PyTorch-owned device memory handed to the C ABI as raw pointers together with
``torch.cuda.current_stream()``; the library never allocates, frees or synchronises.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import BF16, F32, check, lib


# ---------------------------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------------------------
def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"gmlm_amd kernels take float32 or bfloat16 tensors, got {t.dtype}")


def _cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.GmlmHipError("gmlm_amd ops run on the GPU only (tensor on %s); there is no CPU path" % t.device)
    dev = next(t for t in ts if t is not None).device
    _lib.require_gfx950(dev.index if dev.index is not None else torch.cuda.current_device())


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the launch stream (bench.py roofline).
    Off by default (TIMER is None): the product path records nothing."""

    def __init__(self):
        self.records = []            # (name, start_event, end_event, meta)

    def span(self, name, **meta):
        return _Span(self, name, meta)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, meta in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, bytes=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["bytes"] += meta.get("bytes", 0.0)
            d["flops"] += meta.get("flops", 0.0)
        return out


class _Span:
    def __init__(self, timer, name, meta):
        self.timer, self.name, self.meta = timer, name, meta

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record()
        return self

    def __exit__(self, *exc):
        self.e1.record()
        self.timer.records.append((self.name, self.e0, self.e1, self.meta))
        return False


class _NoSpan:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


TIMER: Optional[KernelTimer] = None
_NOSPAN = _NoSpan()


def _span(name, **meta):
    return TIMER.span(name, **meta) if TIMER is not None else _NOSPAN


def spmm_algorithmic_bytes(num_edges, num_segments, out_rows, f, elem, edge_w=False):
    """every edge reads its source row once (no cache credit) + a 4-byte index,
    per-segment row pointers, one output row per segment (+ a 4-byte 1/cnt lookup per edge backward)."""
    return float(num_edges * (f * elem + 4 + (4 if edge_w else 0)) + (num_segments + 1) * 4 + out_rows * f * elem)


SEED_DEVICE: Optional[torch.Tensor] = None   # optional int64[1] device counter added to every dropout seed (see graphs.py)


def _sd():
    """Device pointer of the seed counter (None = plain host seeds).  Kernels add *ptr to the host seed when they run, so a
    captured hipGraph draws fresh dropout masks on every replay (the counter is bumped inside the graph)."""
    return None if SEED_DEVICE is None else SEED_DEVICE.data_ptr()


def draw_seed() -> int:
    """Dropout seed from torch's CPU generator: replayed by torch.utils.checkpoint (RNG state is
    preserved there), costs no device sync."""
    return int(torch.randint(0, 2 ** 62, (1,), device="cpu").item())


# ---------------------------------------------------------------------------------------------
# K1: degree / edge types (integer)
# ---------------------------------------------------------------------------------------------
def degree(index: torch.Tensor, num_nodes: Optional[int] = None, dtype=torch.float32) -> torch.Tensor:
    """Drop-in for ``torch_geometric.utils.degree`` (main.py:65, 256): float32 occurrence counts."""
    _cuda(index)
    index = index.to(torch.long).contiguous()
    if num_nodes is None:
        num_nodes = int(index.max().item()) + 1 if index.numel() else 0
    if dtype == torch.float32:
        out = torch.empty(num_nodes, dtype=torch.float32, device=index.device)
        check(lib().gmlm_degree_f32(_ptr(index), index.numel(), num_nodes, _ptr(out), _stream()), "gmlm_degree_f32")
        return out
    out = torch.empty(num_nodes, dtype=torch.int32, device=index.device)
    check(lib().gmlm_degree_i32(_ptr(index), index.numel(), num_nodes, _ptr(out), _stream()), "gmlm_degree_i32")
    return out.to(dtype)


def edge_types_from_degree(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """main.py:253-267 as two kernels (degree histogram + bucketing); int64 [E], bit-exact."""
    _cuda(edge_index)
    src = edge_index[0].to(torch.long).contiguous()
    e = src.numel()
    deg = torch.empty(num_nodes, dtype=torch.int32, device=src.device)
    check(lib().gmlm_degree_i32(_ptr(src), e, num_nodes, _ptr(deg), _stream()), "gmlm_degree_i32")
    et = torch.empty(e, dtype=torch.long, device=src.device)
    check(lib().gmlm_edge_bucket(_ptr(src), _ptr(deg), e, num_nodes, _ptr(et), _stream()), "gmlm_edge_bucket")
    return et


# ---------------------------------------------------------------------------------------------
# K2/K3: relation-segmented mean aggregation
# ---------------------------------------------------------------------------------------------
def _spmm(src, rowptr, idx, edge_w, mean, num_segments, f, out, split=None):
    """split: optional graph.SplitPlan (long segments of a skewed graph are reduced chunk-wise)."""
    if split is not None and split.n_long > 0:
        partial = torch.empty(split.n_chunks, f, dtype=torch.float32, device=src.device)
        extra = (split.thresh, _ptr(split.long_seg), _ptr(split.chunk_ptr), _ptr(split.chunk_owner), split.n_long,
                 split.n_chunks, _ptr(partial))
    else:
        extra = (0, None, None, None, 0, 0, None)
    check(lib().gmlm_rgcn_mean_spmm(_ptr(src), src.shape[0], src.stride(0), _ptr(rowptr), _ptr(idx), _ptr(edge_w),
                                    1 if mean else 0, num_segments, f, _ptr(out), f, _dt(src), *extra, _stream()),
          "gmlm_rgcn_mean_spmm")


def column_sum(x: torch.Tensor) -> torch.Tensor:
    """fp32 column sums of a 2-D fp32 / bf16 tensor on the K4 statistics kernel (fixed-order partials, deterministic): the bias
    gradients of the dense layers.  (``x.sum(0, dtype=float32)`` - ATen's reduction, on shapes that take its cross-block path
    with a semaphore buffer - goes wrong from the second replay on inside a recorded hipGraph on this PyTorch / ROCm stack:
    tools/dev/aten_sum_graph_repro.py; this kernel needs no zero-initialised scratch.)"""
    _cuda(x)
    x = x.contiguous()
    n, f = x.shape
    out = torch.empty(2, f, dtype=torch.float32, device=x.device)
    if n == 0:
        return out[0].zero_()
    ws = _ws(lib().gmlm_colstats_workspace_bytes(n, f), x.device)
    check(lib().gmlm_colstats(_ptr(x), _dt(x), None, n, f, _ptr(out[0]), _ptr(out[1]), _ptr(ws), ws.numel(), _stream()), "gmlm_colstats")
    return out[0]


def device_split_plan(rowptr: torch.Tensor, num_items: int, thresh: int = 64):
    """``graph.SplitPlan`` for segments known only on the device (they change every step): capacity-sized arrays filled by ONE
    kernel, no host round trip; unused slots are -1 and skipped by the aggregation kernels (gmlm_split_plan_build)."""
    import ctypes
    from .graph import SplitPlan
    cl, cc = ctypes.c_int64(), ctypes.c_int64()
    check(lib().gmlm_split_plan_capacity(int(num_items), int(thresh), ctypes.byref(cl), ctypes.byref(cc)), "gmlm_split_plan_capacity")
    dev = rowptr.device
    long_seg = torch.empty(cl.value, dtype=torch.int32, device=dev)
    chunk_ptr = torch.empty(cl.value + 1, dtype=torch.int32, device=dev)
    chunk_owner = torch.empty(cc.value, dtype=torch.int32, device=dev)
    check(lib().gmlm_split_plan_build(_ptr(rowptr), rowptr.numel() - 1, int(num_items), int(thresh), _ptr(long_seg), _ptr(chunk_ptr),
                                      _ptr(chunk_owner), _stream()), "gmlm_split_plan_build")
    return SplitPlan(int(thresh), long_seg, chunk_ptr, chunk_owner, int(cl.value), int(cc.value))


class RGCNAggregate(torch.autograd.Function):
    """H[i, r*F:(r+1)*F] = mean_{j->i, type r} x[j]  (K2); backward = same kernel on the transposed CSR (K3)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, csr) -> torch.Tensor:
        _cuda(x)
        x = x.contiguous()
        n_src, f = x.shape
        n = csr.num_nodes
        if n_src != csr.num_src:
            raise ValueError(f"x has {n_src} rows but the graph expects {csr.num_src} source rows")
        out = torch.empty(n, csr.r_active * f, dtype=x.dtype, device=x.device)
        with _span("spmm_fwd", bytes=spmm_algorithmic_bytes(csr.num_edges, n * csr.r_active, n * csr.r_active, f,
                                                           x.element_size()), f=f):
            _spmm(x, csr.rowptr, csr.col, None, True, n * csr.r_active, f, out, csr.split)
        ctx.csr, ctx.n_src = csr, x.shape[0]
        return out

    @staticmethod
    def backward(ctx, gh: torch.Tensor):
        csr = ctx.csr
        gh = gh.contiguous()
        n, rf = gh.shape
        f = rf // csr.r_active
        gx = torch.empty(csr.num_src, f, dtype=gh.dtype, device=gh.device)
        with _span("spmm_bwd", bytes=spmm_algorithmic_bytes(csr.num_edges, csr.num_src, csr.num_src, f, gh.element_size(),
                                                           True), f=f):
            _spmm(gh.view(n * csr.r_active, f), csr.t_rowptr, csr.t_seg, csr.inv_cnt, False, csr.num_src, f, gx, csr.t_split)
        return gx, None


# ---------------------------------------------------------------------------------------------
# K4: GraphNorm + GELU + dropout
# ---------------------------------------------------------------------------------------------
class GraphNormAct(torch.autograd.Function):
    """y = dropout(act(GraphNorm(z))).  z [n, f] in the compute dtype (the GEMM output feeds the kernels
    directly: no fp32 staging copy); y and dz in the same dtype; statistics fp32.

    ``reducer``: optional callable(tensor) summing a small fp32 tensor over all ranks in place (the
    1-D node partition makes GraphNorm's column statistics global: SURVEY.md §8e); ``n_total`` is the
    global row count.
    """

    @staticmethod
    def forward(ctx, z, weight, bias, mean_scale, eps, act, p, seed, out_dtype, reducer, n_total):
        _cuda(z)
        z = z.to(out_dtype).contiguous()
        n, f = z.shape
        n_total = int(n_total) if n_total else n
        dev = z.device
        w, b, ms = _f32c(weight), _f32c(bias), _f32c(mean_scale)
        st = _stream()
        dt = _dt(z)
        stats = torch.empty(2, f, dtype=torch.float32, device=dev)
        shift = None
        if reducer is None and n > 0:
            shift = z[0].float().contiguous()          # shifted sums: no cancellation in E[o^2]
        ws = _ws(lib().gmlm_colstats_workspace_bytes(n, f), dev)
        check(lib().gmlm_colstats(_ptr(z), dt, _ptr(shift), n, f, _ptr(stats[0]), _ptr(stats[1]), _ptr(ws), ws.numel(), st),
              "gmlm_colstats")
        if reducer is not None:
            # exact two-pass in the distributed case: all-reduce sum(x) -> mean, then sum((x - mean*ms)^2)
            reducer(stats)
            mu = stats[0] / n_total
            shift = (mu * ms).contiguous()
            check(lib().gmlm_colstats(_ptr(z), dt, _ptr(shift), n, f, _ptr(stats[0]), _ptr(stats[1]), _ptr(ws), ws.numel(),
                                      st), "gmlm_colstats")
            reducer(stats)
            mean = mu.contiguous()
            rstd = torch.rsqrt(stats[1] / n_total + eps).contiguous()
        else:
            mean = torch.empty(f, dtype=torch.float32, device=dev)
            rstd = torch.empty(f, dtype=torch.float32, device=dev)
            check(lib().gmlm_graphnorm_finalize(_ptr(stats[0]), _ptr(stats[1]), _ptr(shift), _ptr(ms), n_total, f, eps,
                                                _ptr(mean), _ptr(rstd), st), "gmlm_graphnorm_finalize")
        y = torch.empty(n, f, dtype=out_dtype, device=dev)
        sd = ctx.sd = _sd()
        check(lib().gmlm_graphnorm_apply(_ptr(z), _ptr(mean), _ptr(rstd), _ptr(w), _ptr(b), _ptr(ms), n, f, int(act),
                                         float(p), seed, sd, _ptr(y), dt, st), "gmlm_graphnorm_apply")
        ctx.save_for_backward(z, mean, rstd, w, b, ms)
        ctx.cfg = (int(act), float(p), seed, reducer, n_total)
        return y

    @staticmethod
    def backward(ctx, gy):
        z, mean, rstd, w, b, ms = ctx.saved_tensors
        act, p, seed, reducer, n_total = ctx.cfg
        gy = gy.contiguous().to(z.dtype)
        n, f = z.shape
        dev = z.device
        st = _stream()
        dt = _dt(z)
        gs = torch.empty(2, f, dtype=torch.float32, device=dev)
        ws = _ws(lib().gmlm_colstats_workspace_bytes(n, f), dev)
        sd = ctx.sd
        check(lib().gmlm_graphnorm_bwd_stats(_ptr(gy), dt, _ptr(z), _ptr(mean), _ptr(rstd), _ptr(w), _ptr(b), _ptr(ms),
                                             n, f, act, p, seed, sd, _ptr(gs), _ptr(ws), ws.numel(), st),
              "gmlm_graphnorm_bwd_stats")
        gs_local = gs
        if reducer is not None:
            gs_local = gs.clone()
            reducer(gs)
        dz = torch.empty(n, f, dtype=z.dtype, device=dev)
        if reducer is None:
            dw = torch.empty(f, dtype=torch.float32, device=dev)
            db = torch.empty_like(dw)
            dms = torch.empty_like(dw)
            check(lib().gmlm_graphnorm_bwd_apply(_ptr(gy), dt, _ptr(z), _ptr(mean), _ptr(rstd), _ptr(w), _ptr(b),
                                                 _ptr(ms), _ptr(gs), n, n_total, f, act, p, seed, sd, _ptr(dz), _ptr(dw),
                                                 _ptr(db), _ptr(dms), st), "gmlm_graphnorm_bwd_apply")
        else:
            check(lib().gmlm_graphnorm_bwd_apply(_ptr(gy), dt, _ptr(z), _ptr(mean), _ptr(rstd), _ptr(w), _ptr(b),
                                                 _ptr(ms), _ptr(gs), n, n_total, f, act, p, seed, sd, _ptr(dz), None, None,
                                                 None, st), "gmlm_graphnorm_bwd_apply")
            # parameter grads: LOCAL contributions only (the gradient all-reduce sums them over ranks)
            dw, db = gs_local[1].clone(), gs_local[0].clone()
            m2 = gs[1] / n_total
            # sum over LOCAL rows of do_j = w*rstd*(sum_local gz - m2 * sum_local ohat)
            sum_oh_local = (z.float().sum(0) - n * mean * ms) * rstd
            dms = -mean * (w * rstd * (gs_local[0] - m2 * sum_oh_local))
        return dz, dw, db, dms, None, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------
# K6: bias + dropout + residual + LayerNorm (+GELU)
# ---------------------------------------------------------------------------------------------
class BiasResLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, residual, gamma, beta, eps, act, p, seed):
        _cuda(x)
        shape = x.shape
        f = shape[-1]
        x2 = x.contiguous().view(-1, f)
        rows = x2.shape[0]
        res2 = None if residual is None else residual.to(x.dtype).contiguous().view(-1, f)
        bf = None if bias is None else _f32c(bias)
        g, b = _f32c(gamma), _f32c(beta)
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        sd = ctx.sd = _sd()
        check(lib().gmlm_bias_res_layernorm_fwd(_ptr(x2), _ptr(bf), _ptr(res2), _ptr(g), _ptr(b), rows, f, float(eps),
                                                int(act), float(p), seed, sd, _ptr(y), _ptr(mean), _ptr(rstd), _dt(x2), _stream()),
              "gmlm_bias_res_layernorm_fwd")
        ctx.save_for_backward(x2, res2, bf, g, b, mean, rstd)
        ctx.cfg = (int(act), float(p), seed, shape, bias is not None, residual is not None)
        return y.view(shape)

    @staticmethod
    def backward(ctx, gy):
        x2, res2, bf, g, b, mean, rstd = ctx.saved_tensors
        act, p, seed, shape, has_bias, has_res = ctx.cfg
        rows, f = x2.shape
        gy2 = gy.contiguous().view(rows, f).to(x2.dtype)
        dx = torch.empty_like(x2)
        dres = torch.empty_like(x2) if has_res else None
        dg = torch.empty(f, dtype=torch.float32, device=x2.device)
        dbeta = torch.empty_like(dg)
        dbias = torch.empty_like(dg) if has_bias else None
        ws = _ws(lib().gmlm_layernorm_bwd_workspace_bytes(rows, f), x2.device)
        sd = ctx.sd
        check(lib().gmlm_bias_res_layernorm_bwd(_ptr(gy2), _ptr(x2), _ptr(bf), _ptr(res2), _ptr(g), _ptr(b), _ptr(mean),
                                                _ptr(rstd), rows, f, act, p, seed, sd, _ptr(dx), _ptr(dres), _ptr(dg),
                                                _ptr(dbeta), _ptr(dbias), _dt(x2), _ptr(ws), ws.numel(), _stream()),
              "gmlm_bias_res_layernorm_bwd")
        return (dx.view(shape), dbias, None if dres is None else dres.view(shape), dg, dbeta, None, None, None, None)


def bias_res_layernorm(x, bias, residual, gamma, beta, eps=1e-5, act=False, p=0.0, training=False):
    p = float(p) if training else 0.0
    return BiasResLayerNorm.apply(x, bias, residual, gamma, beta, eps, act, p, draw_seed() if p > 0 else 0)


# ---------------------------------------------------------------------------------------------
# K5/K7: attention core
# ---------------------------------------------------------------------------------------------
def _rows_view(t: torch.Tensor, h: int, d: int):
    """[b, l, h*d-wide slice] view -> (tensor, row stride in elements); last dim must be contiguous."""
    if t.stride(-1) != 1 or t.shape[-1] != h * d:
        raise ValueError("attention operands must be [b, l, h*d] with a contiguous last dim")
    if t.stride(0) != t.shape[1] * t.stride(1) and t.shape[0] > 1:
        raise ValueError("attention operands must have batch stride = l * row stride")
    return t.stride(1)


def _short_path(dtype, d: int, rows: int, pairs: int) -> bool:
    """Same condition as the C entries: bf16, d = 64, every sequence <= 128 rows, >= 512 (sequence, head) pairs -> the
    LDS-resident short-sequence kernels (their backward forms delta itself and needs neither ``out`` nor ``out_lo``)."""
    return dtype == torch.bfloat16 and d == 64 and rows <= 128 and pairs >= 512


class Attention(torch.autograd.Function):
    """out[b, lq, h*d] = softmax(q k^T * scale + key-padding mask) v with streaming softmax.

    q: [b, lq, h*d], k/v: [b, lk, h*d] (may be strided slices of a fused QKV buffer);
    kv_len: int32 [b] or None.
    """

    @staticmethod
    def forward(ctx, q, k, v, kv_len, h, scale, p, seed):
        _cuda(q, k, v)
        b, lq, hd = q.shape
        lk = k.shape[1]
        d = hd // h
        qs, ks, vs = _rows_view(q, h, d), _rows_view(k, h, d), _rows_view(v, h, d)
        out = torch.empty(b, lq, hd, dtype=q.dtype, device=q.device)
        lse = torch.empty(b, h, lq, dtype=torch.float32, device=q.device)
        # bf16 streaming path: keep the rounding residual of the output for the backward's delta (gmlm_hip.h, out_lo)
        want_lo = q.dtype == torch.bfloat16 and not _short_path(q.dtype, d, max(lq, lk), b * h) and \
            any(t.requires_grad for t in (q, k, v))
        out_lo = torch.empty_like(out) if want_lo else None
        with _span("attn_fwd_d%d" % d, flops=4.0 * b * h * lq * lk * d):
            sd = ctx.sd = _sd()
            check(lib().gmlm_attention_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(kv_len), b, h, lq, lk, d, qs, ks, vs, float(scale),
                                           float(p), seed, sd, _ptr(out), _ptr(out_lo), _ptr(lse), _dt(q), None, 0, None, 0, _stream()),
                  "gmlm_attention_fwd")
        ctx.save_for_backward(q, k, v, out, lse, kv_len, out_lo)
        ctx.cfg = (h, float(scale), float(p), seed)
        return out

    @staticmethod
    def backward(ctx, gout):
        q, k, v, out, lse, kv_len, out_lo = ctx.saved_tensors
        h, scale, p, seed = ctx.cfg
        b, lq, hd = q.shape
        lk = k.shape[1]
        d = hd // h
        gout = gout.contiguous().to(q.dtype)
        # one fused [.., 3*h*d] gradient buffer when q/k/v came from a fused QKV GEMM, else three
        dq = torch.empty(b, lq, hd, dtype=q.dtype, device=q.device)
        dk = torch.empty(b, lk, hd, dtype=q.dtype, device=q.device)
        dv = torch.empty(b, lk, hd, dtype=q.dtype, device=q.device)
        ws = _ws(lib().gmlm_attention_bwd_workspace_bytes(b, h, lq, lk, d), q.device)
        with _span("attn_bwd_d%d" % d, flops=10.0 * b * h * lq * lk * d):
            sd = ctx.sd
            check(lib().gmlm_attention_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(gout), _ptr(lse), _ptr(kv_len), b, h,
                                           lq, lk, d, q.stride(1), k.stride(1), v.stride(1), scale, p, seed, sd, _ptr(dq),
                                           _ptr(dk), _ptr(dv), hd, hd, hd, _dt(q), None, 0, _ptr(ws), ws.numel(), None, None,
                                           _ptr(out_lo), None, 0, _stream()),
                  "gmlm_attention_bwd")
        return dq, dk, dv, None, None, None, None, None


class AttentionQKV(torch.autograd.Function):
    """Self-attention on a fused QKV buffer (BERT layout).  Forward reads q/k/v in place through row
    strides; backward writes dq|dk|dv straight into ONE gradient buffer of the same layout, so no
    slice-gradient accumulation passes are needed.

    padded mode: qkv [b, l, 3*h*d], kv_len int32 [b] or None.
    packed mode: qkv [total_rows, 3*h*d], cu_seqlens int32 [b+1], max_len = longest sequence: sequence i owns
    rows [cu[i], cu[i+1]) and attends only to itself; no padded token exists anywhere.
    """

    @staticmethod
    def forward(ctx, qkv, kv_len, h, scale, p, seed, cu_seqlens, max_len, pair_count=None, groups=None, *bias_masters):
        """``bias_masters``: optional (q, k, v) bias parameters of the fused projection that produced ``qkv``.  Their values
        are not read (the projection has added them); they are inputs so that backward can RETURN their gradient, the
        column sums of dqkv, which the short-sequence kernel forms in-kernel (otherwise one reduction pass here).
        ``groups`` (packed mode): int32 [G + 1] device tensor; work item g of the short-sequence kernels = sequences
        [groups[g], groups[g+1]) - at most 13 sequences / 128 rows (``pack_sequence_groups``)."""
        _cuda(qkv)
        qkv = qkv.contiguous()
        packed = cu_seqlens is not None
        if packed:
            l, hd3 = qkv.shape
            b = cu_seqlens.numel() - 1
        else:
            b, l, hd3 = qkv.shape
        hd = hd3 // 3
        d = hd // h
        q, k, v = qkv[..., :hd], qkv[..., hd:2 * hd], qkv[..., 2 * hd:]
        out = torch.empty(qkv.shape[:-1] + (hd,), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((h, l) if packed else (b, h, l), dtype=torch.float32, device=qkv.device)
        # packed: exact sum of len^2 when the caller knows it (host copy of the lengths), else the bound max_len * rows
        flops = 4.0 * h * d * ((float(pair_count) if pair_count else float(max_len) * l) if packed else float(b) * l * l)
        rows = int(max_len) if packed else l
        short = _short_path(qkv.dtype, d, rows, b * h)
        if groups is not None and not (packed and short):
            groups = None                                       # only the short-sequence kernels pack sequences into work items
        n_groups = 0 if groups is None else groups.numel() - 1
        out_lo = torch.empty_like(out) if (qkv.dtype == torch.bfloat16 and not short and qkv.requires_grad) else None
        nb = out.numel() * out.element_size()                   # one of q, k, v, o
        with _span("attn_fwd_d%d" % d, flops=flops, bytes=(4 + (1 if out_lo is not None else 0)) * nb):
            sd = ctx.sd = _sd()
            check(lib().gmlm_attention_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(kv_len), b, h, l, l, d, hd3, hd3, hd3, float(scale),
                                           float(p), seed, sd, _ptr(out), _ptr(out_lo), _ptr(lse), _dt(qkv), _ptr(cu_seqlens),
                                           int(max_len), _ptr(groups), n_groups, _stream()), "gmlm_attention_fwd")
        if short:
            ctx.save_for_backward(qkv, lse, kv_len, cu_seqlens, groups)        # the fused backward never reads the forward output
        else:
            ctx.save_for_backward(qkv, lse, kv_len, cu_seqlens, groups, out, out_lo)
        ctx.short = short
        ctx.cfg = (h, float(scale), float(p), seed, int(max_len), b, l, flops)
        ctx.bias_shapes = [(m.shape[0], m.dtype) for m in bias_masters]
        return out

    @staticmethod
    def backward(ctx, gout):
        if ctx.short:
            qkv, lse, kv_len, cu_seqlens, groups = ctx.saved_tensors
            out, out_lo = gout, None                            # placeholder pointer: not read on this path
        else:
            qkv, lse, kv_len, cu_seqlens, groups, out, out_lo = ctx.saved_tensors
        h, scale, p, seed, max_len, b, l, flops = ctx.cfg
        want_db = bool(ctx.bias_shapes) and any(ctx.needs_input_grad[10:])
        hd3 = qkv.shape[-1]
        hd = hd3 // 3
        d = hd // h
        gout = gout.contiguous().to(qkv.dtype)
        q, k, v = qkv[..., :hd], qkv[..., hd:2 * hd], qkv[..., 2 * hd:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :hd], dqkv[..., hd:2 * hd], dqkv[..., 2 * hd:]
        ws = _ws(lib().gmlm_attention_bwd_workspace_bytes(1 if cu_seqlens is not None else b, h, l, l, d), qkv.device)
        # short-sequence path (one launch per (sequence, head): same condition as the C entry): the column sums of dqkv --
        # the bias gradient of the fused QKV projection that produced qkv -- come out of the kernel as well
        rows = max_len if cu_seqlens is not None else l
        fused_db = want_db and ctx.short
        n_groups = 0 if groups is None else groups.numel() - 1
        part = db = None
        if fused_db:
            part = _ws(4 * (n_groups or b) * hd3, qkv.device).view(torch.float32)
            db = torch.empty(hd3, dtype=torch.float32, device=qkv.device)
        # q, k, v, dO in; dq, dk, dv out (+ o, o_lo on the streaming path, which forms delta from them)
        with _span("attn_bwd_d%d" % d, flops=2.5 * flops, bytes=(7 + (0 if ctx.short else 2)) * qkv.numel() // 3 * qkv.element_size()):
            sd = ctx.sd
            check(lib().gmlm_attention_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(gout), _ptr(lse), _ptr(kv_len), b, h,
                                           l, l, d, hd3, hd3, hd3, scale, p, seed, sd, _ptr(dq), _ptr(dk), _ptr(dv), hd3, hd3,
                                           hd3, _dt(qkv), _ptr(cu_seqlens), max_len, _ptr(ws), ws.numel(), _ptr(part), _ptr(db),
                                           _ptr(out_lo), _ptr(groups), n_groups, _stream()),
                  "gmlm_attention_bwd")
        dbs = ()
        if want_db:
            if db is None:
                db = column_sum(dqkv.reshape(-1, hd3))
            dbs = tuple(g if g.dtype == dt_ else g.to(dt_)
                        for g, (_, dt_) in zip(db.split([n_ for n_, _ in ctx.bias_shapes], 0), ctx.bias_shapes))
        return (dqkv, None, None, None, None, None, None, None, None, None, *dbs)


class AttentionBlock:
    """Per-block arithmetic of the ring K|V exchange (gmlm_amd.dist._RingAttention) on the HIP kernels, outside
    autograd: ``fwd`` returns the block's normalised output and its log-sum-exp; ``bwd`` evaluates the block's share of
    dQ / dK / dV from the GLOBAL output and log-sum-exp (P is recomputed as exp(S - lse_global), delta = rowsum(dO * O))."""

    def __init__(self, num_heads: int, scale: float, dropout_p: float = 0.0):
        self.h, self.scale, self.p = num_heads, float(scale), float(dropout_p)

    def fwd(self, q, k, v, kv_len, seed):
        _cuda(q, k, v)
        b, lq, hd = q.shape
        lk, d = k.shape[1], hd // self.h
        if d not in (64, 96):
            raise NotImplementedError("ring attention needs a native head dim (64 or 96)")
        out = torch.empty(b, lq, hd, dtype=q.dtype, device=q.device)
        lse = torch.empty(b, self.h, lq, dtype=torch.float32, device=q.device)
        with _span("attn_fwd_d%d" % d, flops=4.0 * b * self.h * lq * lk * d):
            sd = _sd()
            check(lib().gmlm_attention_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(kv_len), b, self.h, lq, lk, d, _rows_view(q, self.h, d),
                                           _rows_view(k, self.h, d), _rows_view(v, self.h, d), self.scale, self.p, int(seed), sd,
                                           _ptr(out), None, _ptr(lse), _dt(q), None, 0, None, 0, _stream()), "gmlm_attention_fwd")
        return out, lse

    def bwd(self, q, k, v, out, dout, lse, kv_len, seed, out_lo=None):
        """``out_lo``: optional bf16 residual of the (global) output, O = out + out_lo (the ring merges its blocks in fp32 and
        knows it): delta then has fp32-like accuracy."""
        b, lq, hd = q.shape
        lk, d = k.shape[1], hd // self.h
        dq = torch.empty(b, lq, hd, dtype=q.dtype, device=q.device)
        dk = torch.empty(b, lk, hd, dtype=q.dtype, device=q.device)
        dv = torch.empty(b, lk, hd, dtype=q.dtype, device=q.device)
        ws = _ws(lib().gmlm_attention_bwd_workspace_bytes(b, self.h, lq, lk, d), q.device)
        out, dout, lse = out.contiguous(), dout.contiguous(), lse.contiguous()
        with _span("attn_bwd_d%d" % d, flops=10.0 * b * self.h * lq * lk * d):
            sd = _sd()
            check(lib().gmlm_attention_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse), _ptr(kv_len), b, self.h,
                                           lq, lk, d, _rows_view(q, self.h, d), _rows_view(k, self.h, d), _rows_view(v, self.h, d),
                                           self.scale, self.p, int(seed), sd, _ptr(dq), _ptr(dk), _ptr(dv), hd, hd, hd, _dt(q), None, 0,
                                           _ptr(ws), ws.numel(), None, None, _ptr(out_lo), None, 0, _stream()), "gmlm_attention_bwd")
        return dq, dk, dv


def attention_qkv(qkv, kv_len, num_heads, scale, dropout_p=0.0, training=False, cu_seqlens=None, max_len=0, pair_count=None,
                  bias_masters=(), groups=None):
    p = float(dropout_p) if training else 0.0
    return AttentionQKV.apply(qkv, kv_len, num_heads, scale, p, draw_seed() if p > 0 else 0, cu_seqlens, max_len, pair_count,
                              groups, *bias_masters)


SHORT_GROUP_ROWS, SHORT_GROUP_SEQS = 128, 13


def pack_sequence_groups(lens) -> torch.Tensor:
    """Greedy packing of CONSECUTIVE sequences into work items of the short-sequence attention kernels: a group takes
    sequences while its rows stay <= 128 and its count <= 13 (gmlm_hip.h, seq_groups).  ``lens``: host sequence of ints
    (every length <= 128).  Returns the int32 [G + 1] boundaries (host tensor)."""
    bounds, rows, cnt = [0], 0, 0
    for i, n in enumerate(lens):
        n = int(n)
        if n > SHORT_GROUP_ROWS:
            raise ValueError(f"sequence {i} has {n} rows: the short-sequence kernels take <= {SHORT_GROUP_ROWS}")
        if cnt and (rows + n > SHORT_GROUP_ROWS or cnt == SHORT_GROUP_SEQS):
            bounds.append(i)
            rows = cnt = 0
        rows += n
        cnt += 1
    bounds.append(len(lens))
    return torch.tensor(bounds, dtype=torch.int32)


def attention(q, k, v, kv_len, num_heads, scale, dropout_p=0.0, training=False):
    p = float(dropout_p) if training else 0.0
    return Attention.apply(q, k, v, kv_len, num_heads, scale, p, draw_seed() if p > 0 else 0)


# ---------------------------------------------------------------------------------------------
# K8: masked mean pool + row scatter
# ---------------------------------------------------------------------------------------------
class MeanPoolScatter(torch.autograd.Function):
    """plm_embeds[node_idx[b]] = sum_t hs[b,t]*[t<len[b]] / max(len[b],1e-9)  (main.py:351-358), in place.
    hs is [b, l, p] (padded, ``lens``) or [total_rows, p] (packed, ``cu_seqlens``)."""

    @staticmethod
    def forward(ctx, plm_embeds, hs, lens, node_idx, cu_seqlens=None, valid_rows=None):
        """``valid_rows`` (packed): rows of ``hs`` covered by ``cu_seqlens``; rows behind them (padding the token count to a
        friendlier multiple) get a zero gradient."""
        _cuda(hs, plm_embeds)
        hs = hs.contiguous()
        if cu_seqlens is not None:
            b, l, p = cu_seqlens.numel() - 1, 0, hs.shape[-1]
        else:
            b, l, p = hs.shape
        check(lib().gmlm_meanpool_scatter_fwd(_ptr(hs), _ptr(lens), _ptr(node_idx), b, l, p, _ptr(plm_embeds), _dt(hs),
                                              _ptr(cu_seqlens), _stream()), "gmlm_meanpool_scatter_fwd")
        ctx.mark_dirty(plm_embeds)
        ctx.save_for_backward(lens, node_idx, cu_seqlens)
        ctx.cfg = (b, l, p, hs.dtype, tuple(hs.shape), valid_rows)
        return plm_embeds

    @staticmethod
    def backward(ctx, g):
        lens, node_idx, cu_seqlens = ctx.saved_tensors
        b, l, p, dtype, shape, valid_rows = ctx.cfg
        g = g.contiguous().float()
        dhs = torch.empty(shape, dtype=dtype, device=g.device)
        if valid_rows is not None and valid_rows < shape[0]:
            dhs[valid_rows:].zero_()
        check(lib().gmlm_meanpool_scatter_bwd(_ptr(g), _ptr(lens), _ptr(node_idx), b, l, p, _ptr(dhs), _dt(dhs),
                                              _ptr(cu_seqlens), _stream()), "gmlm_meanpool_scatter_bwd")
        # rows written by this micro-batch were overwritten: no gradient flows to their previous value;
        # the previous value is the zero-initialised buffer (a constant), so passing g through is harmless.
        return g, dhs, None, None, None, None


# ---------------------------------------------------------------------------------------------
# BERT embedding sum (word + token type 0 + position) and its segment-sum backward
# ---------------------------------------------------------------------------------------------
class EmbedSum(torch.autograd.Function):
    """out[t] = word[tok[t]] + type[0] + pos[pos_ids[t]]  (hf:modeling_bert.py:53-108 ahead of the LayerNorm) stored as
    ``out_dtype``: one pass instead of two gathers, two adds and a cast.  Backward: the output gradient is summed per token id
    and per position id with the aggregation kernel over a stable segment sort of the ids (K1 + K2: deterministic, no atomics);
    the type-0 row's gradient is the sum of the position rows' gradients (every token has exactly one position)."""

    @staticmethod
    def forward(ctx, tok, pos_ids, word_w, pos_w, type_w, out_dtype, bad_flag=None):
        """``bad_flag``: optional int32[1] device tensor the kernel sets to 1 when an id lies outside its table (the ids are
        then clamped, never read out of bounds); ``check_embed_ids`` is the raising, synchronising form."""
        _cuda(word_w, pos_w, type_w)
        tok, pos_ids = tok.long().contiguous(), pos_ids.long().contiguous()
        word, pos, typ = _f32c(word_w), _f32c(pos_w), _f32c(type_w)
        rows, p = tok.numel(), word.shape[1]
        out = torch.empty(rows, p, dtype=out_dtype, device=word.device)
        check(lib().gmlm_embed_sum_fwd(_ptr(word), _ptr(pos), _ptr(typ), _ptr(tok), _ptr(pos_ids), rows, p, word.shape[0],
                                       pos.shape[0], _ptr(out), _dt(out), _ptr(bad_flag), _stream()), "gmlm_embed_sum_fwd")
        ctx.save_for_backward(tok, pos_ids)
        ctx.cfg = (word.shape[0], pos.shape[0], type_w.shape[0], p)
        return out

    @staticmethod
    def backward(ctx, g):
        from .graph import _segment_sort
        tok, pos_ids = ctx.saved_tensors
        vocab, npos, ntype, p = ctx.cfg
        g = g.contiguous()
        grads = []
        # The segment sums feed fp32 master tables and a position row sums ~10^3 sequences: they are accumulated AND stored in
        # fp32.  The aggregation kernel keeps one storage dtype for source and result, so a bf16 gradient is widened once
        # ([T, P] fp32 transient: one pass, 1-2 % of the encoder backward) instead of rounding every sum to 8 bits.
        g32 = g if g.dtype == torch.float32 else g.float()
        # Real text makes a few segments very long ([CLS] / [SEP] once per sequence, every position id once per sequence that is
        # long enough, frequent words): segments of more than 64 rows are reduced chunk-wise through a split plan built on the
        # device from this step's rowptr (one lane group walking a 1,257-row - or, at arxiv size, 40,000-row - segment alone was
        # the whole duration of this kernel)
        for ids, nseg in ((tok, vocab), (pos_ids, npos)):
            _, perm, rowptr, _ = _segment_sort(ids, None, None, 1, nseg)
            acc = torch.empty(nseg, p, dtype=torch.float32, device=g.device)
            _spmm(g32, rowptr, perm, None, False, nseg, p, acc, device_split_plan(rowptr, ids.numel()))
            grads.append(acc)
        d_type = torch.zeros(ntype, p, dtype=torch.float32, device=g.device)
        d_type[0] = column_sum(grads[1])                   # (not ATen's global reduction: see column_sum)
        return None, None, grads[0], grads[1], d_type, None, None


def embed_sum(tok, pos_ids, word_w, pos_w, type_w, out_dtype, bad_flag=None):
    return EmbedSum.apply(tok, pos_ids, word_w, pos_w, type_w, out_dtype, bad_flag)


def check_embed_ids(input_ids: torch.Tensor, lens: torch.Tensor, vocab: int, npos: int) -> None:
    """What F.embedding / HF BERT enforce per call (an id outside the table raises), done ONCE for a static id tensor:
    token ids in [0, vocab), sequence lengths <= max_position_embeddings.  One device sync; ``TokenizedTexts`` remembers
    the tables it was checked against."""
    if input_ids.numel() == 0:
        return
    lo, hi, lmax = int(input_ids.min()), int(input_ids.max()), int(lens.max()) if lens.numel() else 0
    if lo < 0 or hi >= vocab:
        raise IndexError(f"token id out of range: ids span [{lo}, {hi}], the word embedding table has {vocab} rows")
    if lmax > npos:
        raise IndexError(f"sequence of {lmax} tokens exceeds max_position_embeddings = {npos}")


# ---------------------------------------------------------------------------------------------
# K9: soft-mask blend
# ---------------------------------------------------------------------------------------------
class SoftMask(torch.autograd.Function):
    """x~[i] = mask[i] ? (1-beta) x[i] + beta*token : x[i]  (main.py:92-99); output dtype / column
    padding (zeros) chosen by the caller so the first RGCN layer gets 16-byte aligned rows."""

    @staticmethod
    def forward(ctx, x, mask, token, beta, out_dtype, out_cols):
        _cuda(x)
        x = x.float().contiguous()
        n, f = x.shape
        out_cols = int(out_cols) if out_cols else f
        m8 = mask.to(torch.uint8).contiguous()
        tok = _f32c(token).view(-1)
        out = torch.empty(n, out_cols, dtype=out_dtype, device=x.device)
        check(lib().gmlm_softmask_blend_fwd(_ptr(x), _ptr(m8), _ptr(tok), float(beta), n, f, _ptr(out), out_cols, _dt(out),
                                            _stream()), "gmlm_softmask_blend_fwd")
        ctx.save_for_backward(m8)
        ctx.cfg = (float(beta), n, f, token.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (m8,) = ctx.saved_tensors
        beta, n, f, tshape = ctx.cfg
        g = g.float().contiguous()
        dtok = torch.empty(f, dtype=torch.float32, device=g.device)
        ws = _ws(lib().gmlm_colstats_workspace_bytes(n, f), g.device)
        check(lib().gmlm_softmask_blend_bwd(_ptr(g), g.stride(0), _ptr(m8), beta, n, f, _ptr(dtok), _ptr(ws), ws.numel(),
                                            _stream()), "gmlm_softmask_blend_bwd")
        return None, None, dtok.view(tshape), None, None, None


def soft_masking_gnn_input(x, gnn_perturb_mask, mask_token_embed, beta=0.7, out_dtype=torch.float32, out_cols=None):
    """Drop-in for main.py:92-99 (same name / argument order); gradient flows to ``mask_token_embed``."""
    return SoftMask.apply(x, gnn_perturb_mask.to(x.device), mask_token_embed.to(x.device), beta, out_dtype, out_cols)


# ---------------------------------------------------------------------------------------------
# bias + GELU (+dropout)
# ---------------------------------------------------------------------------------------------
class BiasGelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, p, seed):
        _cuda(x)
        shape = x.shape
        f = shape[-1]
        x2 = x.contiguous().view(-1, f)
        bf = None if bias is None else _f32c(bias)
        y = torch.empty_like(x2)
        sd = ctx.sd = _sd()
        check(lib().gmlm_bias_gelu_fwd(_ptr(x2), _ptr(bf), x2.shape[0], f, float(p), seed, sd, _ptr(y), _dt(x2), _stream()),
              "gmlm_bias_gelu_fwd")
        ctx.save_for_backward(x2, bf)
        ctx.cfg = (float(p), seed, shape, bias is not None)
        return y.view(shape)

    @staticmethod
    def backward(ctx, gy):
        x2, bf = ctx.saved_tensors
        p, seed, shape, has_bias = ctx.cfg
        rows, f = x2.shape
        gy2 = gy.contiguous().view(rows, f).to(x2.dtype)
        dx = torch.empty_like(x2)
        dbias = torch.empty(f, dtype=torch.float32, device=x2.device) if has_bias else None
        ws = _ws(lib().gmlm_bias_gelu_bwd_workspace_bytes(rows, f, _dt(x2)), x2.device)
        sd = ctx.sd
        check(lib().gmlm_bias_gelu_bwd(_ptr(gy2), _ptr(x2), _ptr(bf), rows, f, p, seed, sd, _ptr(dx), _ptr(dbias), _dt(x2),
                                       _ptr(ws), ws.numel(), _stream()), "gmlm_bias_gelu_bwd")
        return dx.view(shape), dbias, None, None


def bias_gelu(x, bias, p=0.0, training=False):
    p = float(p) if training else 0.0
    return BiasGelu.apply(x, bias, p, draw_seed() if p > 0 else 0)
