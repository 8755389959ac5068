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
  a9  CrossAttention   all-gather of the fused K|V projection rows (local Q x all keys); backward =
                       reduce-scatter of dK|dV.  (Ring exchange for graphs whose K/V do not fit is the
                       next step; S3/S4-size K/V are a few MB.)
  grads               one bucketed all-reduce(sum) of the replicated parameter gradients per step; the RGCN
                       basis weights (70 % of the parameters) are reduced as the R_a composed relation
                       weights instead of the 30 bases (30/R_a times fewer bytes)

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
    @staticmethod
    def forward(ctx, x, part):
        plan, group = part.plan, part.group
        send = _staged(x.index_select(0, part.send_idx).contiguous(), group)
        recv = send.new_empty((plan.n_halo,) + tuple(x.shape[1:]))
        dist.all_to_all_single(recv, send, plan.recv_counts, plan.send_counts, group=group)
        ctx.part = part
        return torch.cat([x, recv.to(x.device)], 0)

    @staticmethod
    def backward(ctx, g):
        part = ctx.part
        plan, group = part.plan, part.group
        n = plan.n_local
        g_halo = _staged(g[n:].contiguous(), group)
        back = g_halo.new_empty((part.send_idx.numel(),) + tuple(g.shape[1:]))
        dist.all_to_all_single(back, g_halo, plan.send_counts, plan.recv_counts, group=group)
        gx = g[:n].clone()
        back = back.to(g.device)
        # deterministic: a row goes to a given peer at most once, so each peer's slice has unique targets (its adds
        # cannot collide) and the slices are applied in fixed peer order: no run-to-run reordering of the sum
        off = 0
        for cnt in plan.send_counts:
            if cnt:
                gx.index_add_(0, part.send_idx[off:off + cnt], back[off:off + cnt])
            off += cnt
        return gx, None


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

    def build_csr(self, num_relations: int):
        from .graph import build_rel_csr
        p = self.plan
        self.csr = build_rel_csr(p.local_edge_index.to(self.device), p.n_local, num_relations,
                                 p.local_edge_type.to(self.device), num_src=p.n_local + p.n_halo,
                                 active_relations=p.active_relations)
        return self.csr

    # -- exchange steps ---------------------------------------------------------------------
    def with_halo(self, x: torch.Tensor) -> torch.Tensor:
        return _HaloExchange.apply(x, self)

    def _all_reduce(self, t: torch.Tensor, op) -> torch.Tensor:
        """In-place all-reduce of ``t`` wherever it lives: gloo has no device collectives (device tensors are staged
        through the host: tests / 1-GPU rehearsals), RCCL has no host collectives (small host tensors are staged
        through the device)."""
        gloo = _is_gloo(self.group)
        if t.is_cuda == (not gloo):
            dist.all_reduce(t, op=op, group=self.group)
        else:
            c = t.cpu() if gloo else t.to(self.device)
            dist.all_reduce(c, op=op, group=self.group)
            t.copy_(c)
        return t

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        return self._all_reduce(t, dist.ReduceOp.SUM)

    def all_reduce_min(self, t: torch.Tensor) -> torch.Tensor:
        """MIN over ranks (collective yes / no decisions)."""
        return self._all_reduce(t, dist.ReduceOp.MIN)

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        return self._all_reduce(t, dist.ReduceOp.MAX)

    def all_gather_rows(self, x: torch.Tensor) -> torch.Tensor:
        return _AllGatherRows.apply(x, self)

    def local_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.plan.lo:self.plan.hi]

    def all_reduce_grads(self, module: torch.nn.Module, bucket_bytes: int = 256 << 20) -> None:
        """Sum replicated parameter gradients over ranks in a few large flat buckets (xGMI is per-link
        bound: few big messages).  Which parameters take part is decided collectively: a parameter that has a
        gradient on SOME rank is summed (ranks without one contribute zeros); a parameter without a gradient on
        ANY rank (the dead ``residual_proj3`` branch, the unused BERT pooler) keeps ``grad = None`` exactly like
        the single-GPU / reference run, so AdamW neither decays it nor creates state for it."""
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
            flat = torch.cat([g.reshape(-1).float() for g in bucket])
            self.all_reduce_sum(flat)
            off = 0
            for g in bucket:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                off += n
            bucket, size = [], 0

        for p, h in zip(params, has.tolist()):
            if h < 0.5:
                continue                      # no rank has a gradient for it
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            bucket.append(p.grad)
            size += p.grad.numel() * 4
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
