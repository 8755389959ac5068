"""nn.Module surface of the operators the reference reaches through PyG / its own small modules.

* ``RGCNConv`` / ``GraphNorm`` / ``degree`` keep the PyG call signatures used at main.py:189-203,
  256, 272-309 (parameter names ``weight, comp, root, bias`` / ``weight, bias, mean_scale``) so a
  reference state dict loads unchanged.
* ``CrossAttention`` / ``MultiScaleFusion`` keep main.py:139-180's names and shapes.
All arithmetic that is not a plain dense GEMM runs in libgmlm_hip.so (gmlm_amd.ops); dense
projections go to hipBLASLt through ``torch.mm`` / ``F.linear``.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import GraphCache, RelCSR

_GRAPHS = GraphCache(capacity=8)


def compute_dtype(module_dtype: Optional[torch.dtype] = None) -> torch.dtype:
    """fp32 unless the caller runs under ``torch.amp.autocast('cuda')`` (main.py:348,446,543) or set an
    explicit dtype; autocast (fp16 or bf16) maps to bf16, the reduced precision these kernels have."""
    if module_dtype is not None:
        return module_dtype
    if torch.is_autocast_enabled("cuda"):
        if torch.get_autocast_dtype("cuda") == torch.float16:
            _warn_fp16_once()
        return torch.bfloat16
    return torch.float32


_FP16_WARNED = False


def _warn_fp16_once() -> None:
    """The reference's ``torch.amp.autocast('cuda')`` defaults to fp16; these kernels compute the reduced-precision
    path in bf16 (same operand width, 3 fewer mantissa bits, no loss scaling needed).  Said once, not silently."""
    global _FP16_WARNED
    if not _FP16_WARNED:
        _FP16_WARNED = True
        import warnings
        warnings.warn("gmlm_amd: fp16 autocast is executed with bf16 operands (the HIP kernels' reduced precision); "
                      "use torch.amp.autocast('cuda', dtype=torch.bfloat16) to make that explicit", stacklevel=3)


def _glorot(t: torch.Tensor) -> None:
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


def _linear(x, weight, bias=None):
    """x @ weight^T (+ bias) in x's dtype with autocast off (explicit precision policy).  Inside ``shadow_params`` the operands
    come from the region's shadow and the gradients go to the fp32 masters directly; otherwise per-call casts."""
    sh = _SHADOW
    if sh is not None and x.dtype == sh.dtype:
        wc = sh.get(id(weight))
        bc = sh.get(id(bias)) if bias is not None else None
        if wc is not None and (bias is None or bc is not None):
            return _mm(x, wc, bc, (weight,) + ((bias,) if bias is not None else ()))
    with torch.autocast("cuda", enabled=False):
        w = weight.to(x.dtype)
        b = None if bias is None else bias.to(x.dtype)
        return F.linear(x, w, b)


_F32_OUT = [True]          # torch.bmm(..., out_dtype=float32) available (checked on first use)
_F32_MM = [True]           # torch.mm(..., out_dtype=float32) likewise


_SLICES_MEASURED = {(2304, 768): 14, (3072, 768): 14, (768, 3072): 14}
SPLITK_ROW_QUANTUM = 112   # lcm(14, 16): a token count that is a multiple of it leaves no tail rows for either slice count


def _splitk_wgrad(dy2: torch.Tensor, x2: torch.Tensor, keep_fp32: bool = False) -> torch.Tensor:
    """dW [N, K] = dy2^T [N, T] @ x2 [T, K] (returned in fp32 when ``keep_fp32``, else in dy2's dtype).  The reduction dim is the token count (10^4..10^5) while the
    output is only a few 256x256 tiles, so one hipBLASLt call leaves most CUs idle; cutting T into S
    slices (batched GEMM, fp32 partials) and adding them fills the chip (measured 1.5-2.7x on MI355X)."""
    t, n = dy2.shape
    k = x2.shape[1]
    tiles = ((n + 255) // 256) * ((k + 255) // 256)
    s = 1
    if tiles < 128 and t >= 16384:                    # (node-level layers, t = a few thousand rows: one GEMM; slicing them left a tail GEMM + an add per weight)
        import math
        # measured on MI355X (tools/ubench/wgrad_shapes.py, T = 91,712): 16 slices is within 5 % of the best power of two for every
        # BERT shape once T >= 64k; 14 fills the chip better for three of the four BERT-base weights (QKV 322 vs 415 us, FFN
        # 440 vs 455 and 444 vs 447 us; the 768 x 768 output projection stays at 16: 128 vs 161 us)
        s = _SLICES_MEASURED.get((n, k), 16) if t >= 65536 else min(16, max(1, 2 ** round(math.log2(200.0 / tiles))))
    if s == 1:
        if keep_fp32 and dy2.dtype != torch.float32 and _F32_MM[0]:
            try:                                          # fp32 result straight out of the GEMM: no cast pass over dW
                return torch.mm(dy2.t(), x2, out_dtype=torch.float32)
            except (RuntimeError, TypeError):
                _F32_MM[0] = False
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


# ---------------------------------------------------------------------------------------------
# Compute-dtype copies of fp32 master parameters: ONE flat buffer, ONE multi-tensor copy per region
# ---------------------------------------------------------------------------------------------
class ParamShadow:
    """bf16 operands of a region's dense layers.  ``specs``: list of (key, [master tensors stacked along dim 0], pad_rows):
    each entry becomes one contiguous [sum(rows) + pad_rows, ...] view of a flat zero-initialised buffer in ``dtype`` (the pad
    rows stay zero: the first RGCN layer's root is padded to the 16-byte-aligned input width).  Filled by ONE
    ``_foreach_copy_`` instead of a cast kernel per parameter; the layers hand their weight / bias gradients straight to the
    fp32 masters (``_Linear``), so no cast kernels run in backward either."""

    def __init__(self, specs, dtype, device):
        total, plan = 0, []
        for key, masters, pad in specs:
            rows = sum(m.shape[0] for m in masters) + pad
            tail = tuple(masters[0].shape[1:])
            numel = rows
            for d in tail:
                numel *= d
            plan.append((key, masters, (rows,) + tail, total, numel))
            total += (numel + 127) // 128 * 128                    # 256-byte aligned operands
        self.dtype = dtype
        with torch.no_grad():
            flat = torch.zeros(total, dtype=dtype, device=device)
            self.flat = flat
            self.views, dsts, srcs = {}, [], []
            for key, masters, shape, off, numel in plan:
                v = flat[off:off + numel].view(shape)
                self.views[key] = v
                r0 = 0
                for m in masters:
                    dsts.append(v[r0:r0 + m.shape[0]])
                    srcs.append(m.detach())
                    r0 += m.shape[0]
            if dsts:
                torch._foreach_copy_(dsts, srcs)

    def get(self, key):
        return self.views.get(key)

    def record_stream(self, stream) -> None:
        """The operands are also read by kernels on ``stream`` (a branch of a recorded step): keep the allocator from recycling
        the buffer on its own stream while that branch still runs."""
        if self.flat.is_cuda:
            self.flat.record_stream(stream)


_SHADOW: Optional[ParamShadow] = None


class shadow_params:
    """``with shadow_params(specs, dtype, device):`` - the dense layers called inside find their compute-dtype operands in the
    shadow (``_linear`` by ``id(weight)``); no-op for fp32 (the masters ARE the operands)."""

    def __init__(self, specs, dtype, device):
        self.args = (specs, dtype, device)
        self.prev = None

    def __enter__(self):
        global _SHADOW
        self.prev = _SHADOW
        specs, dtype, device = self.args
        _SHADOW = ParamShadow(specs, dtype, device) if (dtype != torch.float32 and device.type == "cuda") else None
        return _SHADOW

    def __exit__(self, *exc):
        global _SHADOW
        _SHADOW = self.prev
        return False


class use_shadow:
    """Re-enter an existing shadow (a checkpoint recompute runs a block in backward, outside the region's ``with``)."""

    def __init__(self, sh):
        self.sh, self.prev = sh, None

    def __enter__(self):
        global _SHADOW
        self.prev = _SHADOW
        if _SHADOW is None:
            _SHADOW = self.sh
        return _SHADOW

    def __exit__(self, *exc):
        global _SHADOW
        _SHADOW = self.prev
        return False


def linear_specs(*mods):
    """Shadow entries of plain ``nn.Linear`` modules: weight under id(weight), bias under id(bias)."""
    out = []
    for m in mods:
        out.append((id(m.weight), [m.weight], 0))
        if m.bias is not None:
            out.append((id(m.bias), [m.bias], 0))
    return out


class _RootAddmm(torch.autograd.Function):
    """bias + x @ root (RGCNConv's self-loop term, PyG RGCNConv.forward) with shadow operands; weight / bias gradients in fp32
    to the masters.  ``rootc`` is [in + pad, out] (pad rows zero), ``root`` the [in, out] master."""

    @staticmethod
    def forward(ctx, x, rootc, bc, root, bias):
        ctx.save_for_backward(x, rootc)
        ctx.rows = root.shape[0]
        return torch.addmm(bc, x, rootc)

    @staticmethod
    def backward(ctx, dy):
        x, rootc = ctx.saved_tensors
        dx = dy @ rootc.t() if ctx.needs_input_grad[0] else None
        droot = _splitk_wgrad(x, dy, keep_fp32=True)[:ctx.rows] if ctx.needs_input_grad[3] else None   # x^T dy: [in + pad, out]
        db = ops.column_sum(dy) if ctx.needs_input_grad[4] else None
        return dx, None, None, droot, db


class _SelectRows(torch.autograd.Function):
    """x[idx] for UNIQUE row indices (the relations that occur): the backward is a plain row copy into zeros, not the
    atomicAdd scatter of ``index_select``'s backward (unique rows need no accumulation; no atomics = bitwise reproducible)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.rows = x.shape[0]
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = g.new_zeros((ctx.rows,) + tuple(g.shape[1:]))
        out.index_copy_(0, idx, g)
        return out, None


class _BasisCompose(torch.autograd.Function):
    """W_r = sum_b comp[r, b] * weight[b] for the relations that occur (K10, one streaming pass over the
    bases).  In the node-partitioned run the gradient of the (R_a x in*out) composed weights is all-reduced
    BEFORE it is expanded to the 30 bases: R_a/30 of the bytes of reducing d(weight) itself (the bases are
    70 % of all parameters at hc = 768)."""

    @staticmethod
    def forward(ctx, comp_a, weight2d, reducer):
        comp_a, weight2d = comp_a.detach().float().contiguous(), weight2d.detach().float().contiguous()
        ra, nb = comp_a.shape
        cols = weight2d.shape[1]
        ctx.save_for_backward(comp_a, weight2d)
        ctx.reducer = reducer
        if not weight2d.is_cuda or nb > 32 or ra > 5 or cols % 4:
            if not weight2d.is_cuda:
                raise ops._lib.GmlmHipError("gmlm_amd ops run on the GPU only; there is no CPU path")
            ctx.hip = False
            return comp_a @ weight2d                    # unusual geometry (> 32 bases / > 5 relations): hipBLASLt
        ctx.hip = True
        w = torch.empty(ra, cols, dtype=torch.float32, device=weight2d.device)
        ops.check(ops.lib().gmlm_basis_compose_fwd(ops._ptr(comp_a), ops._ptr(weight2d), ra, nb, cols, ops._ptr(w),
                                                   ops._stream()), "gmlm_basis_compose_fwd")
        return w

    @staticmethod
    def backward(ctx, dw):
        comp_a, weight2d = ctx.saved_tensors
        dw = dw.float().contiguous()
        if ctx.reducer is not None:
            ctx.reducer(dw)
        if not ctx.hip:
            return dw @ weight2d.t(), comp_a.t() @ dw, None
        ra, nb = comp_a.shape
        cols = weight2d.shape[1]
        dweight = torch.empty_like(weight2d)
        dcomp = torch.empty_like(comp_a)
        ws = ops._ws(ops.lib().gmlm_basis_compose_bwd_workspace_bytes(ra, nb, cols), dw.device)
        ops.check(ops.lib().gmlm_basis_compose_bwd(ops._ptr(comp_a), ops._ptr(weight2d), ops._ptr(dw), ra, nb, cols,
                                                   ops._ptr(dweight), ops._ptr(dcomp), ops._ptr(ws), ws.numel(), ops._stream()),
                  "gmlm_basis_compose_bwd")
        return dcomp, dweight, None


class RGCNConv(nn.Module):
    """PyG ``RGCNConv(in, out, num_relations, num_bases)`` with aggr='mean', root weight and bias."""

    def __init__(self, in_channels: int, out_channels: int, num_relations: int, num_bases: Optional[int] = None, **kw):
        super().__init__()
        if kw:
            raise TypeError(f"unsupported RGCNConv options: {sorted(kw)}")
        if num_bases is None:
            raise NotImplementedError("only the basis-decomposed form used by the reference (num_bases=30) is implemented")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_relations, self.num_bases = num_relations, num_bases
        self.weight = nn.Parameter(torch.empty(num_bases, in_channels, out_channels))
        self.comp = nn.Parameter(torch.empty(num_relations, num_bases))
        self.root = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        _glorot(self.weight), _glorot(self.comp), _glorot(self.root)
        nn.init.zeros_(self.bias)

    def relation_weights(self, csr: RelCSR, dtype, in_pad: int = 0, reducer=None) -> torch.Tensor:
        """[R_a * (in + pad), out]: W_r = sum_b comp[r, b] weight[b] for the relations that occur.
        ``reducer`` (node-partitioned run): all-reduces d(W_r); ``weight``/``comp`` gradients then come out
        already summed over ranks and are flagged so the bucketed gradient all-reduce skips them."""
        comp = _SelectRows.apply(self.comp, csr.active_index)                # [R_a, B]
        w = _BasisCompose.apply(comp, self.weight.view(self.num_bases, -1), reducer)
        self.weight._gmlm_grad_reduced = self.comp._gmlm_grad_reduced = reducer is not None
        w = w.view(len(csr.active_relations), self.in_channels, self.out_channels)
        if in_pad:
            w = F.pad(w, (0, 0, 0, in_pad))
        return w.reshape(-1, self.out_channels).to(dtype)

    def forward_csr(self, x: torch.Tensor, csr: RelCSR, reducer=None, halo_wait=None) -> torch.Tensor:
        """x: [n_src, in (+pad)] in the compute dtype -> [n, out] in the same dtype (two accumulating
        hipBLASLt GEMMs: bias + x root, then += H W_cat; no fp32 staging passes over [n, out]).

        Node partition: rows [n, n_src) of x are halo rows whose all-to-all may still be in flight; the root GEMM
        (owned rows only) and the basis composition run first, ``x = halo_wait(x)`` right before the aggregation
        (``PartitionContext.halo_ready``), so the exchange hides under them; in backward the same node starts the
        reverse exchange ahead of the root GEMM's backward."""
        in_pad = x.shape[1] - self.in_channels
        n = csr.num_nodes
        sh = _SHADOW
        rootc = sh.get(id(self.root)) if (sh is not None and x.dtype == sh.dtype) else None
        bc = sh.get(id(self.bias)) if rootc is not None else None
        with torch.autocast("cuda", enabled=False):
            w = self.relation_weights(csr, x.dtype, in_pad, reducer)
            if bc is not None and rootc.shape[0] == x.shape[1]:
                out = _RootAddmm.apply(x if x.shape[0] == n else x[:n], rootc, bc, self.root, self.bias)   # shadow operands, fp32 gradients to the masters
            else:
                root = self.root if not in_pad else F.pad(self.root, (0, 0, 0, in_pad))
                out = torch.addmm(self.bias.to(x.dtype), x[:n], root.to(x.dtype))
        if halo_wait is not None:
            x = halo_wait(x)
        h = ops.RGCNAggregate.apply(x, csr)                                   # [n, R_a*(in+pad)]
        with torch.autocast("cuda", enabled=False):
            return out.addmm_(h, w)

    def forward(self, x, edge_index, edge_type):
        csr = _GRAPHS.get(edge_index, x.size(0), self.num_relations, edge_type)
        cd = compute_dtype()
        return self.forward_csr(x.to(cd).contiguous(), csr)


class GraphNorm(nn.Module):
    """PyG ``GraphNorm(in_channels, eps=1e-5)`` for a single graph (batch=None)."""

    def __init__(self, in_channels: int, eps: float = 1e-5):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = nn.Parameter(torch.ones(in_channels))

    def forward(self, x, batch=None, *, act: bool = False, dropout_p: float = 0.0, out_dtype=None, reducer=None,
                n_total=None):
        if batch is not None:
            raise NotImplementedError("GraphNorm over several graphs (batch vector) is outside the hot path")
        p = float(dropout_p) if self.training else 0.0
        return ops.GraphNormAct.apply(x, self.weight, self.bias, self.mean_scale, self.eps, act, p,
                                      ops.draw_seed() if p > 0 else 0, out_dtype or torch.float32, reducer, n_total)


def degree(index, num_nodes=None, dtype=None):
    return ops.degree(index, num_nodes, dtype or torch.float32)


def cross_attention_specs(mod):
    """Shadow entries of a CrossAttention module: q / out projections by parameter id, the fused K|V operand under
    ("kv_w" | "kv_b", id(module))."""
    return linear_specs(mod.q_proj, mod.out_proj) + [(("kv_w", id(mod)), [mod.k_proj.weight, mod.v_proj.weight], 0),
                                                     (("kv_b", id(mod)), [mod.k_proj.bias, mod.v_proj.bias], 0)]


class CrossAttention(nn.Module):
    """main.py:139-165.  The dense [1,8,N,N] softmax is replaced by the streaming kernel (K7)."""

    def __init__(self, dim, num_heads=8, dropout=0.1):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q_proj = nn.Linear(dim, dim)
        self.k_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.out_proj = nn.Linear(dim, dim)
        self.dropout = nn.Dropout(dropout)
        self.compute_dtype: Optional[torch.dtype] = None

    def forward(self, x, y, kv_gather=None, ring=None):
        """x: [B, N, C] queries, y: [B, M, C] keys/values (the reference calls it with B == 1).

        Node partition: ``kv_gather`` maps the local [B, M, 2C] K|V projection to the
        global one (all-gather, small graphs); ``ring(q, kv, num_heads, block, seed)`` instead runs the ring
        exchange with carried online-softmax state (gmlm_amd.dist.PartitionContext.ring_attention)."""
        cd = compute_dtype(self.compute_dtype)
        b, n, c = x.shape
        xq, yk = x.to(cd), y.to(cd)
        q = _linear(xq, self.q_proj.weight, self.q_proj.bias)
        sh = _SHADOW
        wkv = sh.get(("kv_w", id(self))) if (sh is not None and yk.dtype == sh.dtype) else None
        if wkv is not None:                                           # fused K|V operand straight from the region's shadow
            kv = _mm(yk, wkv, sh.get(("kv_b", id(self))),
                     (self.k_proj.weight, self.v_proj.weight, self.k_proj.bias, self.v_proj.bias))
        else:
            wkv = torch.cat([self.k_proj.weight, self.v_proj.weight], 0)
            bkv = torch.cat([self.k_proj.bias, self.v_proj.bias], 0)
            kv = _linear(yk, wkv, bkv)                                # [B, M, 2C] fused K|V
        if ring is not None:
            p = self.dropout.p if self.training else 0.0
            o = ring(q, kv, self.num_heads, ops.AttentionBlock(self.num_heads, self.scale, p), ops.draw_seed() if p > 0 else 0)
            return _linear(o, self.out_proj.weight, self.out_proj.bias)
        if kv_gather is not None:
            kv = kv_gather(kv)
        k, v = _SplitLast.apply(kv, c)
        o = attention_any_dim(q, k, v, None, self.num_heads, self.scale, self.dropout.p, self.training)
        return _linear(o, self.out_proj.weight, self.out_proj.bias)


class _SplitLast(torch.autograd.Function):
    """(x[..., :c], x[..., c:]) as views; the backward is ONE concatenation of the two gradients instead of autograd's two
    zero-filled full-size buffers, two slice copies and an add."""

    @staticmethod
    def forward(ctx, x, c):
        ctx.c = c
        return x[..., :c], x[..., c:]

    @staticmethod
    def backward(ctx, ga, gb):
        return torch.cat([ga, gb], -1), None


def attention_any_dim(q, k, v, kv_len, num_heads, scale, dropout_p, training):
    """K5/K7 with head dims the MFMA tiling does not cover natively zero-padded to 64 / 96 (exact:
    zero columns change neither the scores nor the real output columns)."""
    b, lq, hd = q.shape
    d = hd // num_heads
    if d in (64, 96):
        return ops.attention(q, k, v, kv_len, num_heads, scale, dropout_p, training)
    if d > 96:
        raise NotImplementedError(f"attention head dim {d} > 96 is not supported by the gfx950 kernels")
    dp = 64 if d < 64 else 96

    def pad(t):
        return F.pad(t.reshape(t.shape[0], t.shape[1], num_heads, d), (0, dp - d)).reshape(t.shape[0], t.shape[1], num_heads * dp)

    o = ops.attention(pad(q), pad(k), pad(v), kv_len, num_heads, scale, dropout_p, training)
    return o.reshape(b, lq, num_heads, dp)[..., :d].reshape(b, lq, hd)


class _ScaledProjectionSum(torch.autograd.Function):
    """acc [N, P] fp32 = sum_k w_k * (emb_k W_k^T + b_k), with the scalar folded into the (small) weight instead of
    the (large) GEMM output: one [N, P] accumulator and one transient GEMM output are alive at a time, and nothing of
    size N x P is saved for backward (the straightforward form keeps four fp32 [N, P] products: 12 KB per node, which
    is what decided whether the 10M-node graph fits on one GPU)."""

    @staticmethod
    def forward(ctx, weights, cd, *args):
        k = len(args) // 3
        embs, ws, bs = args[:k], args[k:2 * k], args[2 * k:]
        acc = None
        with torch.autocast("cuda", enabled=False):
            for i in range(k):
                t = F.linear(embs[i].to(cd), (ws[i].float() * weights[i]).to(cd))
                if acc is None:
                    acc = t.float()
                else:
                    acc.add_(t)
                del t
            acc.add_(sum(bs[i].float() * weights[i] for i in range(k)))
        ctx.save_for_backward(weights, *embs, *ws, *bs)
        ctx.cd, ctx.k = cd, k
        return acc

    @staticmethod
    def backward(ctx, g):
        k, cd = ctx.k, ctx.cd
        weights = ctx.saved_tensors[0]
        embs, ws, bs = ctx.saved_tensors[1:1 + k], ctx.saved_tensors[1 + k:1 + 2 * k], ctx.saved_tensors[1 + 2 * k:]
        g_cd = g.to(cd)
        gsum = ops.column_sum(g) if g.is_cuda else g.sum(0)
        d_weights = torch.zeros_like(weights)
        d_embs, d_ws, d_bs = [], [], []
        with torch.autocast("cuda", enabled=False):
            for i in range(k):
                e = embs[i].to(cd)
                d_embs.append((g_cd @ (ws[i].float() * weights[i]).to(cd)).to(embs[i].dtype))
                dws = (g_cd.t() @ e).float()                     # gradient of the scaled weight
                d_ws.append(dws * weights[i])
                d_bs.append(gsum * weights[i])
                d_weights[i] = (dws * ws[i].float()).sum() + (gsum * bs[i].float()).sum()
        return (d_weights, None, *d_embs, *d_ws, *d_bs)


class _ScaledProjectionCat(torch.autograd.Function):
    """The same sum as ONE GEMM: acc = [e_1 | ... | e_k] [w_1 W_1 | ... | w_k W_k]^T + sum_k w_k b_k, fp32 out of the GEMM
    (bf16 operands, fp32 accumulation across ALL scales: one rounding less per scale than k separate GEMM outputs).  Costs a
    transient [N, sum dims] operand, so MultiScaleFusion uses it while that is small (every benchmarked single-GPU workload
    except the 10M-node graph) and the streaming form above otherwise.  Backward: one data-gradient GEMM, one weight-gradient
    GEMM in fp32, the scalar weights' gradients from column sums of dW * W."""

    @staticmethod
    def forward(ctx, weights, cd, seg, onehot, *args):
        k = len(args) // 3
        embs, ws, bs = args[:k], args[k:2 * k], args[2 * k:]
        with torch.autocast("cuda", enabled=False):
            x = torch.cat([e if e.dtype == cd else e.to(cd) for e in embs], 1)            # [N, D]
            wcat = torch.cat([w.float() for w in ws], 1)                                    # [P, D] fp32, unscaled
            srow = weights.float().index_select(0, seg)                                     # [D]: w_k of each column's scale
            wc = (wcat * srow).to(cd)
            acc = _mm_f32(x, wc)
            bstack = torch.stack([b.float() for b in bs])                                   # [k, P]
            acc.add_(weights.float() @ bstack)
        ctx.save_for_backward(weights, x, wcat, wc, srow, bstack, onehot)
        ctx.cd, ctx.k, ctx.dims, ctx.emb_dtypes = cd, k, [e.shape[1] for e in embs], [e.dtype for e in embs]
        return acc

    @staticmethod
    def backward(ctx, g):
        weights, x, wcat, wc, srow, bstack, onehot = ctx.saved_tensors
        k, cd, dims = ctx.k, ctx.cd, ctx.dims
        with torch.autocast("cuda", enabled=False):
            g_cd = g.to(cd)
            gsum = ops.column_sum(g)                                                        # [P] fp32
            dx = g_cd @ wc                                                                  # [N, D]
            d_embs = [t if t.dtype == dt else t.to(dt) for t, dt in zip(dx.split(dims, 1), ctx.emb_dtypes)]
            dwc = _splitk_wgrad(g_cd, x, keep_fp32=True)                                    # [P, D] fp32: gradient of the SCALED weights
            d_ws = list((dwc * srow).split(dims, 1))
            colsum = ops.column_sum(dwc * wcat)                                             # [D]: sum_p dW[p, c] W[p, c]
            d_weights = (onehot @ colsum + bstack @ gsum).to(weights.dtype)
            d_bs = list((weights.float().view(-1, 1) * gsum.view(1, -1)).unbind(0))
        return (d_weights, None, None, None, *d_embs, *d_ws, *d_bs)


def _mm_f32(a, b):
    """a [N, K] @ b [P, K]^T with an fp32 result straight out of the GEMM when the operands are 16-bit."""
    if a.dtype != torch.float32 and _F32_MM[0]:
        try:
            return torch.mm(a, b.t(), out_dtype=torch.float32)
        except (RuntimeError, TypeError):
            _F32_MM[0] = False
    return (a @ b.t()).float()


class MultiScaleFusion(nn.Module):
    """main.py:167-180: LayerNorm(sum_k softmax(w)_k * Linear_k(emb_k))."""

    def __init__(self, hidden_dims, output_dim):
        super().__init__()
        self.scale_weights = nn.Parameter(torch.ones(len(hidden_dims)) / len(hidden_dims))
        self.projections = nn.ModuleList([nn.Linear(dim, output_dim) for dim in hidden_dims])
        self.layer_norm = nn.LayerNorm(output_dim)
        self.compute_dtype: Optional[torch.dtype] = None
        self.cat_bytes_limit = 1 << 30          # largest transient [N, sum dims] operand of the one-GEMM form (_ScaledProjectionCat)
        self._seg = None

    def _segments(self, dims, device):
        """(int64 [sum dims]: index of the scale every concatenated column belongs to, fp32 [k, sum dims]: the same as a one-hot
        matrix), built once per device / geometry."""
        if self._seg is None or self._seg[0] != (tuple(dims), device):
            seg = torch.repeat_interleave(torch.arange(len(dims)), torch.tensor(dims)).to(device)
            onehot = torch.zeros(len(dims), seg.numel(), device=device).scatter_(0, seg.view(1, -1), 1.0)
            self._seg = ((tuple(dims), device), seg, onehot)
        return self._seg[1], self._seg[2]

    def forward(self, embeddings_list):
        cd = compute_dtype(self.compute_dtype)
        weights = F.softmax(self.scale_weights.float(), dim=0)
        dims = [e.shape[1] for e in embeddings_list]
        n_rows = embeddings_list[0].shape[0]
        if embeddings_list[0].is_cuda and n_rows * sum(dims) * 2 <= self.cat_bytes_limit:
            seg, onehot = self._segments(dims, embeddings_list[0].device)
            acc = _ScaledProjectionCat.apply(weights, cd, seg, onehot, *embeddings_list, *[p.weight for p in self.projections],
                                             *[p.bias for p in self.projections])
        else:
            acc = _ScaledProjectionSum.apply(weights, cd, *embeddings_list, *[p.weight for p in self.projections],
                                             *[p.bias for p in self.projections])
        ln = self.layer_norm
        return ops.bias_res_layernorm(acc, None, None, ln.weight, ln.bias, ln.eps)
