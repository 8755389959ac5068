"""1-D node partition of the hot path across the GPUs of one node.

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).  Rank g owns the
contiguous target rows [lo_g, hi_g) of every [N, .] tensor and every edge whose TARGET it owns;
parameters are replicated.  Exchange steps (nothing else communicates):

  a2  edge types       global out-degree of each source: computed from the full edge list at
                       partition time (static graph), integer, bit-exact
  a3  RGCN aggregation halo exchange before each layer: all-to-all-v of the owned rows that remote
                       targets reference (deduplicated), reverse all-to-all-v + per-peer (collision-free, fixed-order) row adds backward
  a4  GraphNorm        all-reduce(sum) of [2, F] fp32 column statistics (forward: twice, exact
                       two-pass; backward: once)
  a9  CrossAttention   small graphs: all-gather of the fused K|V projection rows (local Q x all keys), backward =
                       reduce-scatter of dK|dV.  Large graphs (``use_ring``): ring exchange — the K|V blocks
                       travel rank to rank (point-to-point send/recv posted BEFORE the block's attention is
                       computed, so the transfer hides under it), each step yields (O_j, lse_j) and the running
                       (O, lse) is merged with carried online-softmax state; K|V is never gathered.  Backward is
                       a second ring: the blocks travel again, each rank adds its dK|dV contribution to the
                       accumulator that travels with the block.
  grads               bucketed all-reduce(sum) of the replicated parameter gradients.  ``GradBuckets`` keeps the
                       gradients as views into a few flat fp32 buffers (no torch.cat / copy-back) and launches a
                       bucket's all-reduce from a post-accumulate hook as soon as its last gradient is final,
                       i.e. under the rest of backward.  The RGCN basis weights (70 % of the parameters) are
                       reduced as the R_a composed relation weights instead of the 30 bases (30/R_a times fewer
                       bytes)

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the halo all-to-all-v drives all links at once,
and gradients go out as a few large flat buckets rather than per-tensor rings.

The plan (who needs which rows) is pure index arithmetic on torch tensors and runs on any device,
so it is covered by world_size-2 gloo tests on CPU; only ``build_csr`` needs the GPU.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.distributed as dist


def _is_gloo(group) -> bool:
    return dist.get_backend(group) == "gloo"


def _staged(t: torch.Tensor, group) -> torch.Tensor:
    """gloo has no device collectives: stage through the host (tests / 1-GPU rehearsals only)."""
    return t.cpu() if (t.is_cuda and _is_gloo(group)) else t


def row_range(n_total: int, world: int, rank: int):
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def owner_of(ids: torch.Tensor, n_total: int, world: int) -> torch.Tensor:
    base, rem = divmod(n_total, world)
    cut = rem * (base + 1)
    big = ids // (base + 1)
    small = rem + (ids - cut) // max(base, 1)
    return torch.where(ids < cut, big, small)


@dataclass
class PartitionPlan:
    n_total: int
    world: int
    rank: int
    lo: int
    hi: int
    local_edge_index: torch.Tensor      # int64 [2, E_local]; src in [0, n_local + n_halo), dst in [0, n_local)
    local_edge_type: torch.Tensor       # int64 [E_local] (from GLOBAL out-degrees)
    halo_ids: torch.Tensor              # int64 [n_halo] global ids of remote source rows, sorted
    recv_counts: List[int]              # rows received from each peer (sum = n_halo), peer-major = sorted order
    send_idx: torch.Tensor              # int64 [sum(send_counts)] LOCAL row ids to send, peer-major
    send_counts: List[int]
    active_relations: List[int] = None  # relations that occur anywhere in the GLOBAL graph (same slots on every rank)

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    @property
    def n_halo(self) -> int:
        return int(self.halo_ids.numel())


def plan_partition(edge_index: torch.Tensor, n_total: int, world: int, rank: int,
                   edge_type: Optional[torch.Tensor] = None) -> PartitionPlan:
    """Every rank holds the full edge list at setup (static graph) and derives its own plan and, without
    communication, what every peer will ask of it."""
    edge_index = edge_index.to(torch.long)
    src, dst = edge_index[0], edge_index[1]
    if edge_type is None:
        deg = torch.bincount(src, minlength=n_total)            # GLOBAL out-degree (main.py:256 semantics)
        d = deg[src]
        edge_type = torch.full_like(src, 3)
        edge_type[d <= 10] = 2
        edge_type[d <= 5] = 1
        edge_type[d <= 2] = 0
    active = sorted(int(r) for r in torch.unique(edge_type).tolist()) or [0]
    lo, hi = row_range(n_total, world, rank)
    mine = (dst >= lo) & (dst < hi)
    s, t_, et = src[mine], dst[mine], edge_type[mine]
    remote = (s < lo) | (s >= hi)
    halo_ids = torch.unique(s[remote])                          # sorted => grouped by owner (contiguous ranges)
    src_local = torch.where(remote, (hi - lo) + torch.searchsorted(halo_ids, s), s - lo)
    owners = owner_of(halo_ids, n_total, world)
    recv_counts = torch.bincount(owners, minlength=world).tolist()
    # what each peer p needs from me: unique sources I own among edges whose target p owns
    own_src = (src >= lo) & (src < hi)
    dst_owner = owner_of(dst, n_total, world)
    need = own_src & (dst_owner != rank)
    key = torch.unique(dst_owner[need] * n_total + src[need])   # sorted by (peer, global src id)
    peer = key // n_total
    send_idx = key % n_total - lo
    send_counts = torch.bincount(peer, minlength=world).tolist()
    return PartitionPlan(n_total, world, rank, lo, hi, torch.stack([src_local, t_ - lo]), et, halo_ids, recv_counts,
                         send_idx, send_counts, active)


class _HaloExchange(torch.autograd.Function):
    """x [n_local, ...] -> [n_local + n_halo, ...]: owned rows, then the remote rows this rank's edges reference.

    The receive lands directly in the tail of the output buffer (no torch.cat pass).  With a device-capable backend
    (RCCL) the all-to-all-v is started asynchronously and the wait is deferred to ``PartitionContext.halo_ready()`` /
    ``wait_halo()``: the caller runs the work that does not need the halo rows (the root GEMM x W_root of RGCNConv) in
    between, so the exchange of layer k hides under it.

    Backward mirrors that.  ``halo_ready`` is an autograd node of its own (``_HaloReady``) created AFTER the root GEMM,
    so in backward it runs BEFORE the root GEMM's backward: it starts the reverse all-to-all-v of the halo rows'
    gradients asynchronously; the root GEMM's data- and weight-gradient GEMMs run meanwhile; this node's backward, which
    runs last, waits and applies the returned rows.  Without a ``halo_ready`` node (plain ``wait_halo()``) the reverse
    exchange is done here, synchronously."""

    @staticmethod
    def forward(ctx, x, part, defer, token):
        plan, group = part.plan, part.group
        n = plan.n_local
        ctx.part, ctx.token = part, token
        send = x.index_select(0, part.send_idx).contiguous()
        out = x.new_empty((n + plan.n_halo,) + tuple(x.shape[1:]))
        out[:n] = x
        if x.is_cuda and _is_gloo(group):                     # rehearsal path: stage through the host, synchronous
            recv = send.cpu().new_empty((plan.n_halo,) + tuple(x.shape[1:]))
            dist.all_to_all_single(recv, send.cpu(), plan.recv_counts, plan.send_counts, group=group)
            out[n:] = recv.to(x.device)
            return out
        work = dist.all_to_all_single(out[n:], send, plan.recv_counts, plan.send_counts, group=group, async_op=True)
        part._halo_pending.append((work, send))               # keep the send buffer alive until the wait
        if not defer:
            part.wait_halo()
        return out

    @staticmethod
    def backward(ctx, g):
        part = ctx.part
        plan = part.plan
        n = plan.n_local
        started = part._halo_bwd.pop(ctx.token, None)
        if started is None:
            started = _start_halo_backward(part, g)
        work, back, _keep = started
        if work is not None:
            work.wait()
        gx = g[:n].clone()
        back = back.to(g.device)
        # deterministic: a row goes to a given peer at most once, so each peer's slice has unique targets (its adds
        # cannot collide) and the slices are applied in fixed peer order: no run-to-run reordering of the sum
        off = 0
        for cnt in plan.send_counts:
            if cnt:
                gx.index_add_(0, part.send_idx[off:off + cnt], back[off:off + cnt])
            off += cnt
        return gx, None, None, None


def _start_halo_backward(part, g):
    """Reverse all-to-all-v of the halo rows' gradients: (work | None, receive buffer, keep-alive)."""
    plan, group = part.plan, part.group
    g_halo = _staged(g[plan.n_local:].contiguous(), group)
    back = g_halo.new_empty((part.send_idx.numel(),) + tuple(g.shape[1:]))
    if g.is_cuda and _is_gloo(group):                         # rehearsal path: synchronous, staged through the host
        dist.all_to_all_single(back, g_halo, plan.send_counts, plan.recv_counts, group=group)
        return None, back, g_halo
    work = dist.all_to_all_single(back, g_halo, plan.send_counts, plan.recv_counts, group=group, async_op=True)
    return work, back, g_halo


class _HaloReady(torch.autograd.Function):
    """Identity on the [n_local + n_halo, ...] buffer that (forward) waits for the halo rows and (backward) STARTS the
    reverse exchange of their gradients; see ``_HaloExchange``."""

    @staticmethod
    def forward(ctx, xh, part, token):
        part.wait_halo()
        ctx.part, ctx.token = part, token
        return xh.view_as(xh)

    @staticmethod
    def backward(ctx, g):
        part = ctx.part
        if part.plan.n_halo or any(part.plan.send_counts):
            part._halo_bwd[ctx.token] = _start_halo_backward(part, g)
        return g, None, None


class _AllGatherRows(torch.autograd.Function):
    """[B, n_local, C] -> [B, N_total, C] in rank order; backward = reduce-scatter (sum) of the gradient.
    Ranks may own different row counts: rows are padded to the largest share so both collectives are
    the equal-size tensor forms (one large message per peer)."""

    @staticmethod
    def forward(ctx, x, part):
        group, world = part.group, part.plan.world
        sizes = [r[1] - r[0] for r in part.ranges]
        mx = max(sizes)
        xs = _staged(x.transpose(0, 1).contiguous(), group)                    # rows leading
        if xs.shape[0] < mx:
            xs = torch.cat([xs, xs.new_zeros((mx - xs.shape[0],) + tuple(xs.shape[1:]))], 0)
        out = xs.new_empty((world * mx,) + tuple(xs.shape[1:]))
        dist.all_gather_into_tensor(out, xs, group=group)
        ctx.part, ctx.sizes, ctx.mx = part, sizes, mx
        if min(sizes) != mx:
            out = torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], 0)
        return out.to(x.device).transpose(0, 1).contiguous()

    @staticmethod
    def backward(ctx, g):
        part, sizes, mx = ctx.part, ctx.sizes, ctx.mx
        group, world, rank = part.group, part.plan.world, part.plan.rank
        gs = _staged(g.transpose(0, 1).contiguous(), group)
        if min(sizes) != mx:
            padded = gs.new_zeros((world * mx,) + tuple(gs.shape[1:]))
            off = 0
            for r in range(world):
                padded[r * mx:r * mx + sizes[r]] = gs[off:off + sizes[r]]
                off += sizes[r]
            gs = padded
        if _is_gloo(group):   # gloo has no reduce_scatter: all-reduce then slice (tests / rehearsals only)
            dist.all_reduce(gs, group=group)
            out = gs[rank * mx:rank * mx + sizes[rank]]
        else:
            out = gs.new_empty((mx,) + tuple(gs.shape[1:]))
            dist.reduce_scatter_tensor(out, gs, group=group)
            out = out[:sizes[rank]]
        return out.to(g.device).transpose(0, 1).contiguous(), None


def _merge_partial(o_acc, lse_acc, o_j, lse_j):
    """Carried online-softmax state: (O, lse) of the keys seen so far merged with a new block's (O_j, lse_j).
    o: [b, n, h, d] fp32 (normalised per block), lse: [b, h, n] fp32 (natural log of the block's softmax sum)."""
    lse_new = torch.logaddexp(lse_acc, lse_j)
    w_acc = torch.exp(lse_acc - lse_new).transpose(1, 2).unsqueeze(-1)          # [b, n, h, 1]; exp(-inf) = 0 for the empty start
    w_j = torch.exp(lse_j - lse_new).transpose(1, 2).unsqueeze(-1)
    return o_acc * w_acc + o_j * w_j, lse_new


class _RingAttention(torch.autograd.Function):
    """softmax(Q_local K_all^T * scale) V_all without ever holding K_all / V_all: W steps, each on one rank's K|V block.

    ``block`` supplies the per-block arithmetic: ``block.fwd(q, k, v, kv_len, seed) -> (o [b,n,h*d], lse [b,h,n])`` and
    ``block.bwd(q, k, v, out, dout, lse, kv_len, seed) -> (dq, dk, dv)`` given the GLOBAL out / lse (the HIP kernels on
    the GPU: gmlm_amd.ops.AttentionBlock; a torch restatement in the CPU tests).  Blocks are padded to the largest
    share and masked by their true row count, so every message has the same size.  Attention-probability dropout
    draws a per-block seed (seed + owner rank): the masks differ from the all-gather path's, the distribution does
    not."""

    @staticmethod
    def forward(ctx, q, kv, part, h, block, seed):
        world, rank, group = part.plan.world, part.plan.rank, part.group
        sizes = [r[1] - r[0] for r in part.ranges]
        mx = max(sizes)
        b, n, c = q.shape
        cur = kv.new_zeros(b, mx, 2 * c)
        cur[:, :kv.shape[1]] = kv
        acc_t = torch.promote_types(q.dtype, torch.float32)       # fp32 state for bf16 / fp32 operands
        o_acc = torch.zeros(b, n, h, c // h, dtype=acc_t, device=q.device)
        lse_acc = torch.full((b, h, n), float("-inf"), dtype=acc_t, device=q.device)
        nxt_rank, prv_rank = (rank + 1) % world, (rank - 1) % world
        for s in range(world):
            owner = (rank - s) % world
            pend, nxt = None, None
            if s + 1 < world:                                   # the next block is on its way while this one is computed
                nxt = torch.empty_like(cur)
                pend = _exchange(cur, nxt, nxt_rank, prv_rank, group)
            kv_len = torch.full((b,), sizes[owner], dtype=torch.int32, device=q.device)
            o_j, lse_j = block.fwd(q, cur[..., :c], cur[..., c:], kv_len, seed + owner)
            o_acc, lse_acc = _merge_partial(o_acc, lse_acc, o_j.to(acc_t).view(b, n, h, c // h), lse_j.to(acc_t))
            if pend is not None:
                pend.wait()
                cur = nxt
        out = o_acc.view(b, n, c).to(q.dtype)
        # the merged output exists in fp32 here: keep its rounding residual for the backward's delta = rowsum(dO * O) (the
        # 2^-9 error of the stored output alone does not cancel in dS = P (dP - delta))
        ctx.out_lo = (o_acc.view(b, n, c) - out.to(acc_t)).to(q.dtype) if out.dtype != acc_t else None
        ctx.save_for_backward(q, kv, out, lse_acc)
        ctx.cfg = (part, h, block, seed, sizes, mx)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse = ctx.saved_tensors
        part, h, block, seed, sizes, mx = ctx.cfg
        world, rank, group = part.plan.world, part.plan.rank, part.group
        b, n, c = q.shape
        dout = dout.contiguous().to(q.dtype)
        cur = kv.new_zeros(b, mx, 2 * c)
        cur[:, :kv.shape[1]] = kv
        acc_t = torch.promote_types(q.dtype, torch.float32)
        dcur = torch.zeros(b, mx, 2 * c, dtype=acc_t, device=q.device)            # gradient accumulator that travels with the block
        dq = torch.zeros(b, n, c, dtype=acc_t, device=q.device)
        nxt_rank, prv_rank = (rank + 1) % world, (rank - 1) % world
        lse32 = lse.to(torch.promote_types(lse.dtype, torch.float32))
        lo_kw = {"out_lo": ctx.out_lo} if ctx.out_lo is not None else {}
        pend_d, dnxt = None, None
        for s in range(world):
            owner = (rank - s) % world
            pend, nxt = None, None
            if s + 1 < world:
                nxt = torch.empty_like(cur)
                pend = _exchange(cur, nxt, nxt_rank, prv_rank, group)
            kv_len = torch.full((b,), sizes[owner], dtype=torch.int32, device=q.device)
            dq_j, dk_j, dv_j = block.bwd(q, cur[..., :c], cur[..., c:], out, dout, lse32, kv_len, seed + owner, **lo_kw)
            # the accumulator of THIS block left the previous rank while this block's arithmetic ran (posted at the end
            # of the previous step): its hop is hidden, only now is it needed
            if pend_d is not None:
                pend_d.wait()
                dcur = dnxt
            dq += dq_j.to(acc_t)
            dcur[..., :c] += dk_j.to(acc_t)
            dcur[..., c:] += dv_j.to(acc_t)
            if pend is not None:
                pend.wait()
            # the accumulator follows its block (after the last step it makes the closing hop home); every rank posts
            # K|V(s), dK|dV(s), K|V(s+1), ... in the same order
            dnxt = torch.empty_like(dcur)
            pend_d = _exchange(dcur, dnxt, nxt_rank, prv_rank, group)
            if nxt is not None:
                cur = nxt
        pend_d.wait()
        dcur = dnxt
        return dq.to(q.dtype), dcur[:, :kv.shape[1]].to(kv.dtype), None, None, None, None


def _exchange(send_t, recv_t, dst, src, group):
    """Post send(send_t -> dst) and recv(recv_t <- src) together.  Returns an object whose ``wait()`` completes both
    (gloo cannot move device memory: rehearsals stage through the host and copy back inside ``wait``)."""
    staged = send_t.is_cuda and _is_gloo(group)
    s_buf = send_t.cpu() if staged else send_t
    r_buf = torch.empty(recv_t.shape, dtype=recv_t.dtype) if staged else recv_t
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, s_buf, dst, group), dist.P2POp(dist.irecv, r_buf, src, group)])

    class _Pending:
        def wait(self):
            for r in reqs:
                r.wait()
            if staged:
                recv_t.copy_(r_buf)
            del s_buf_keep[:]
    s_buf_keep = [s_buf]                                      # the send buffer must outlive the transfer
    return _Pending()


class GradBuckets:
    """Gradient all-reduce that overlaps with backward and never copies gradients.

    The gradients of the replicated parameters live as VIEWS inside a few flat fp32 buffers (``prepare()`` installs
    the views and zeroes the buffers with one memset each: it replaces ``zero_grad``).  A post-accumulate hook counts
    a bucket's gradients down; when the last one is final the bucket's all-reduce starts, asynchronously, while
    backward continues with the earlier layers (buckets are filled in reverse parameter order).

    Ordering.  The bucket all-reduces run on their OWN process group / communicator (``part.grad_group``), not on the
    group that carries backward's other collectives (GraphNorm statistics, halo all-to-all, RGCN basis reducer, K|V
    reduce-scatter / ring).  WHEN a bucket becomes final depends on the rank: a rank without active text nodes never
    sees a PLM gradient and launches those buckets in ``finish()``, i.e. after the GNN-backward collectives that an
    active rank issues after them.  On a shared communicator that is a cross-rank order mismatch (gloo: an
    EnforceNotMet abort; RCCL: a hang or silent corruption); on a communicator of their own the buckets only have to
    be ordered among themselves, and they are: always LAUNCHED in bucket order.

    Which parameters take part is a collective decision made on the first step (``finish()`` votes with MAX on "got a
    gradient"): a parameter without a gradient on ANY rank (dead ``residual_proj3``, BERT pooler) keeps ``grad =
    None`` like the single-GPU run, and is not waited for afterwards.  Every later step repeats the vote (one small
    all-reduce): a parameter of the learned set that got a gradient on NO rank in this step goes back to ``grad =
    None`` for this step (AdamW then skips it exactly like the single-GPU / reference run: e.g. the PLM during a
    pre-training step), and if a parameter outside the set receives a gradient the set is re-learned (that step reduces
    it separately)."""

    def __init__(self, part, module, bucket_bytes: int = 256 << 20):
        from .nn import RGCNConv
        self.part = part
        self.group = part.grad_group()                          # collective: every rank builds its GradBuckets at the same point
        skip = set()
        for m in module.modules():                              # reduced inside backward as composed relation weights
            if isinstance(m, RGCNConv):
                skip.update((id(m.weight), id(m.comp)))
        self.params = [p for p in module.parameters() if p.requires_grad and id(p) not in skip][::-1]
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.bucket_bytes = bucket_bytes
        self.expected = None                                    # bool list once learned
        self.buckets = []
        self._works = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._active = False

    # -- layout --------------------------------------------------------------------------------------------------
    def _build(self):
        keep = [i for i in range(len(self.params)) if self.expected is None or self.expected[i]]
        self.buckets, cur, size = [], [], 0
        for i in keep:
            cur.append(i)
            size += self.params[i].numel() * 4
            if size >= self.bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.flats = []
        self.bucket_of = {}
        for b, idxs in enumerate(self.buckets):
            dev = self.params[idxs[0]].device
            self.flats.append(torch.zeros(sum(self.params[i].numel() for i in idxs), dtype=torch.float32, device=dev))
            for i in idxs:
                self.bucket_of[i] = b

    def prepare(self):
        """Before backward (instead of zero_grad): zero the flat buffers, point every expected parameter's .grad at
        its slice, drop the others' gradients."""
        if not self.buckets or self._layout_for is not self.expected:
            self._build()
            self._layout_for = self.expected
        self.fired = [False] * len(self.params)
        self.pending = [len(b) for b in self.buckets]
        self.next_launch = 0
        self._works = []
        for p in self.params:
            p.grad = None
        for b, idxs in enumerate(self.buckets):
            self.flats[b].zero_()
            off = 0
            for i in idxs:
                p = self.params[i]
                p.grad = self.flats[b][off:off + p.numel()].view_as(p)
                off += p.numel()
        self._active = True

    _layout_for = object()

    # -- during backward -------------------------------------------------------------------------------------------
    def _launch_ready(self, force=False):
        while self.next_launch < len(self.buckets) and (force or self.pending[self.next_launch] == 0):
            flat = self.flats[self.next_launch]
            if flat.is_cuda and _is_gloo(self.group):           # rehearsal: synchronous, staged
                self.part.all_reduce_sum(flat, group=self.group)
            else:
                self._works.append(dist.all_reduce(flat, group=self.group, async_op=True))
            self.next_launch += 1

    def _on_grad(self, p):
        if not self._active:
            return
        i = self.index[id(p)]
        self.fired[i] = True
        b = self.bucket_of.get(i)
        if b is None or self.expected is None:
            return                                              # first step (learning) / unexpected gradient: handled in finish()
        self.pending[b] -= 1
        self._launch_ready()

    # -- after backward --------------------------------------------------------------------------------------------
    def finish(self):
        part = self.part
        self._active = False
        self._launch_ready(force=True)                          # whatever is left, in bucket order
        for w in self._works:
            w.wait()
        self._works = []
        # one vote per step: "some rank has a gradient for parameter i" (after backward: every rank is at the same
        # point of the main group's collective sequence)
        has = torch.tensor([1.0 if f else 0.0 for f in self.fired])
        if has.numel():
            part.all_reduce_max(has)
        has = [bool(v > 0.5) for v in has.tolist()]
        if self.expected is None:
            # learning step: the parameters without a gradient anywhere go back to grad = None
            self.expected = has
            for p, keep in zip(self.params, self.expected):
                if not keep:
                    p.grad = None
            return
        stray = False
        for i, (p, got) in enumerate(zip(self.params, has)):
            if self.expected[i] and not got:
                p.grad = None                                   # no rank produced it in THIS step: same as the single-GPU run
            elif got and not self.expected[i]:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                part.all_reduce_sum(p.grad)                     # outside the learned set: reduced on its own, same order on every rank
                stray = True
        if stray:
            self.expected = None                                # learn the set again on the next step

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class PartitionContext:
    """Attached to ``GraphTextLM.dist``; owns the plan, the local CSR and the collectives."""

    def __init__(self, plan: PartitionPlan, device, group=None):
        self.plan = plan
        self.group = group if group is not None else dist.group.WORLD
        self.device = torch.device(device)
        self.n_total = plan.n_total
        self.ranges = [row_range(plan.n_total, plan.world, r) for r in range(plan.world)]
        self.send_idx = plan.send_idx.to(self.device)
        self.csr = None
        self.use_ring = False            # CrossAttention: ring K|V exchange instead of the K|V all-gather
        self._halo_pending = []
        self._halo_bwd = {}              # token -> reverse exchange started by _HaloReady.backward, consumed by _HaloExchange.backward
        self._halo_token = None
        self._grad_group = None

    def build_csr(self, num_relations: int):
        from .graph import build_rel_csr
        p = self.plan
        self.csr = build_rel_csr(p.local_edge_index.to(self.device), p.n_local, num_relations,
                                 p.local_edge_type.to(self.device), num_src=p.n_local + p.n_halo,
                                 active_relations=p.active_relations)
        return self.csr

    # -- exchange steps ---------------------------------------------------------------------
    def grad_group(self):
        """Process group (communicator) of the gradient-bucket all-reduces: the same ranks as ``group``, a collective
        sequence of its own (see ``GradBuckets``).  Creating it is a collective over the default group, so every rank
        calls this at the same point (``grad_buckets(model)`` on the first step)."""
        if self._grad_group is None:
            ranks = dist.get_process_group_ranks(self.group)
            self._grad_group = dist.new_group(ranks=ranks, backend=dist.get_backend(self.group))
        return self._grad_group

    def with_halo(self, x: torch.Tensor, defer: bool = False) -> torch.Tensor:
        """``defer=True``: the exchange is only STARTED; rows [n_local, n_local + n_halo) of the result are valid
        after ``halo_ready(result)`` (the caller puts independent work in between and uses what ``halo_ready``
        returns: that node also overlaps the BACKWARD exchange with the backward of the work in between) or after a
        plain ``wait_halo()`` (backward exchange synchronous)."""
        token = object()
        out = _HaloExchange.apply(x, self, defer, token)
        self._halo_token = token if defer else None          # one exchange in flight at a time (the layers are sequential)
        return out

    def halo_ready(self, xh: torch.Tensor) -> torch.Tensor:
        """Wait for the exchange started by the last ``with_halo(..., defer=True)`` and return the buffer."""
        token, self._halo_token = self._halo_token, None
        return _HaloReady.apply(xh, self, token if token is not None else object())

    def wait_halo(self) -> None:
        while self._halo_pending:
            work, _send = self._halo_pending.pop()
            work.wait()

    def ring_attention(self, q: torch.Tensor, kv: torch.Tensor, num_heads: int, block, seed: int = 0) -> torch.Tensor:
        return _RingAttention.apply(q, kv, self, num_heads, block, seed)

    def _all_reduce(self, t: torch.Tensor, op, group=None) -> torch.Tensor:
        """In-place all-reduce of ``t`` wherever it lives: gloo has no device collectives (device tensors are staged
        through the host: tests / 1-GPU rehearsals), RCCL has no host collectives (small host tensors are staged
        through the device)."""
        group = self.group if group is None else group
        gloo = _is_gloo(group)
        if t.is_cuda == (not gloo):
            dist.all_reduce(t, op=op, group=group)
        else:
            c = t.cpu() if gloo else t.to(self.device)
            dist.all_reduce(c, op=op, group=group)
            t.copy_(c)
        return t

    def all_reduce_sum(self, t: torch.Tensor, group=None) -> torch.Tensor:
        return self._all_reduce(t, dist.ReduceOp.SUM, group)

    def all_reduce_min(self, t: torch.Tensor) -> torch.Tensor:
        """MIN over ranks (collective yes / no decisions)."""
        return self._all_reduce(t, dist.ReduceOp.MIN)

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        return self._all_reduce(t, dist.ReduceOp.MAX)

    def all_gather_rows(self, x: torch.Tensor) -> torch.Tensor:
        return _AllGatherRows.apply(x, self)

    def local_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.plan.lo:self.plan.hi]

    def grad_buckets(self, module: torch.nn.Module, bucket_bytes: int = 256 << 20) -> "GradBuckets":
        """The overlapping, copy-free gradient reducer for ``module`` (created once, cached on the module)."""
        gb = getattr(module, "_gmlm_grad_buckets", None)
        if gb is None or gb.part is not self:
            gb = GradBuckets(self, module, bucket_bytes)
            module._gmlm_grad_buckets = gb
        return gb

    def all_reduce_grads(self, module: torch.nn.Module, bucket_bytes: int = 256 << 20) -> None:
        """After-backward form (no overlap) for callers that did not ``grad_buckets(module).prepare()`` before backward:
        sums the replicated parameter gradients over ranks in a few large flat fp32 buckets (xGMI is per-link bound: few
        big messages).  Which parameters take part is decided collectively: a parameter that has a gradient on SOME
        rank is summed (ranks without one contribute zeros); a parameter without a gradient on ANY rank (the dead
        ``residual_proj3`` branch, the unused BERT pooler) keeps ``grad = None`` exactly like the single-GPU /
        reference run, so AdamW neither decays it nor creates state for it.  The gradients are moved INTO the flat
        reduce buffer with one multi-tensor copy and then re-pointed at it (views): no ``torch.cat``, no copy back."""
        params = [p for p in module.parameters()
                  if p.requires_grad and not getattr(p, "_gmlm_grad_reduced", False)]   # else: frozen, or summed inside backward (RGCN bases)
        has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params])
        if has.numel():
            self.all_reduce_max(has)
        bucket, size = [], 0

        def flush():
            nonlocal bucket, size
            if not bucket:
                return
            flat = torch.zeros(sum(p.numel() for p in bucket), dtype=torch.float32, device=bucket[0].device)
            views, off = [], 0
            for p in bucket:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            have = [(v, p.grad) for v, p in zip(views, bucket) if p.grad is not None]
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            self.all_reduce_sum(flat)
            for p, v in zip(bucket, views):
                p.grad = v if p.dtype == torch.float32 else v.to(p.dtype)
            bucket, size = [], 0

        for p, h in zip(params, has.tolist()):
            if h < 0.5:
                continue                      # no rank has a gradient for it
            bucket.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                flush()
        flush()


# ---------------------------------------------------------------------------------------------
# partition shard files: one pickle-free .npz per rank, so a graph is partitioned once (offline, on the host)
# and a rank of a later run reads only its own shard instead of the whole edge list
# ---------------------------------------------------------------------------------------------
PARTITION_FORMAT = "gmlm-partition-v1"
_PLAN_ARRAYS = ("local_edge_index", "local_edge_type", "halo_ids", "send_idx")


def partition_file(directory: str, rank: int, world: int) -> str:
    import os
    return os.path.join(directory, f"part-{rank:05d}-of-{world:05d}.npz")


def save_partition(plan: PartitionPlan, path: str) -> None:
    """Write one rank's plan: int64 index arrays + a small int64 header (numpy ``savez``, no pickled objects)."""
    import numpy as np
    arrays = {k: getattr(plan, k).cpu().numpy().astype(np.int64) for k in _PLAN_ARRAYS}
    arrays["header"] = np.asarray([plan.n_total, plan.world, plan.rank, plan.lo, plan.hi], dtype=np.int64)
    arrays["recv_counts"] = np.asarray(plan.recv_counts, dtype=np.int64)
    arrays["send_counts"] = np.asarray(plan.send_counts, dtype=np.int64)
    arrays["active_relations"] = np.asarray(plan.active_relations, dtype=np.int64)
    arrays["format"] = np.frombuffer(PARTITION_FORMAT.encode(), dtype=np.uint8)
    np.savez(path, **arrays)


def load_partition(path: str, world: Optional[int] = None, rank: Optional[int] = None) -> PartitionPlan:
    """Read a shard written by ``save_partition``; checks the format tag, the (world, rank) it was made for
    and the internal consistency of the counts before anything is launched with it."""
    import numpy as np
    with np.load(path, allow_pickle=False) as z:
        if "format" not in z.files or bytes(z["format"]).decode() != PARTITION_FORMAT:
            raise ValueError(f"{path}: not a {PARTITION_FORMAT} file")
        n_total, w, r, lo, hi = (int(v) for v in z["header"])
        t = {k: torch.from_numpy(z[k].astype(np.int64)) for k in _PLAN_ARRAYS}
        recv, send = [int(v) for v in z["recv_counts"]], [int(v) for v in z["send_counts"]]
        active = [int(v) for v in z["active_relations"]]
    if world is not None and (w != world or (rank is not None and r != rank)):
        raise ValueError(f"{path}: shard is rank {r} of {w}, wanted rank {rank} of {world}")
    if (lo, hi) != row_range(n_total, w, r) or len(recv) != w or len(send) != w:
        raise ValueError(f"{path}: header does not describe a 1-D row partition")
    if sum(recv) != t["halo_ids"].numel() or sum(send) != t["send_idx"].numel() or recv[r] or send[r]:
        raise ValueError(f"{path}: exchange counts do not match the index arrays")
    ei = t["local_edge_index"]
    n_loc, n_src = hi - lo, hi - lo + t["halo_ids"].numel()
    if ei.numel() and (int(ei[1].max()) >= n_loc or int(ei[0].max()) >= n_src or int(ei.min()) < 0):
        raise ValueError(f"{path}: local edge index out of range")
    if t["send_idx"].numel() and (int(t["send_idx"].max()) >= n_loc or int(t["send_idx"].min()) < 0):
        raise ValueError(f"{path}: send index out of range")
    return PartitionPlan(n_total, w, r, lo, hi, ei, t["local_edge_type"], t["halo_ids"], recv, t["send_idx"], send, active)


def write_partition_files(edge_index: torch.Tensor, n_total: int, world: int, directory: str,
                          edge_type: Optional[torch.Tensor] = None) -> List[str]:
    """Offline step: partition a graph for ``world`` ranks and write one shard per rank."""
    import os
    os.makedirs(directory, exist_ok=True)
    paths = []
    for r in range(world):
        path = partition_file(directory, r, world)
        save_partition(plan_partition(edge_index, n_total, world, r, edge_type), path)
        paths.append(path)
    return paths


def attach_partition(model, edge_index: Optional[torch.Tensor], n_total: int, device, group=None, edge_type=None,
                     partition_dir: Optional[str] = None) -> PartitionContext:
    """Partition the (replicated, static) graph for this rank — or read this rank's shard from
    ``partition_dir`` — and attach the context to ``model``."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if partition_dir is not None:
        plan = load_partition(partition_file(partition_dir, rank, world), world, rank)
        if plan.n_total != n_total:
            raise ValueError(f"partition shard is for {plan.n_total} nodes, the graph has {n_total}")
    else:
        plan = plan_partition(edge_index, n_total, world, rank, edge_type)
    ctx = PartitionContext(plan, device, group)
    ctx.build_csr(model.num_relations)
    model.dist = ctx
    return ctx
