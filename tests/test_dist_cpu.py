"""world_size-2 gloo tests (CPU) of the 1-D node partition: plan, halo exchange (+ backward),
GraphNorm statistic all-reduce, K|V all-gather (+ reduce-scatter backward), gradient bucket all-reduce.
The oracle is the checker: partitioned aggregation over local+halo rows must equal the global one."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, n, e, f, seed):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gmlm_oracle as O
        from gmlm_amd.dist import PartitionContext, owner_of, plan_partition, row_range
        g = torch.Generator().manual_seed(seed)
        ei = torch.randint(0, n, (2, e), generator=g)
        x = torch.randn(n, f, generator=g)
        plan = plan_partition(ei, n, world, rank)
        ctx = PartitionContext(plan, "cpu")
        lo, hi = plan.lo, plan.hi
        assert (lo, hi) == row_range(n, world, rank)
        assert torch.equal(owner_of(torch.arange(n), n, world)[lo:hi], torch.full((hi - lo,), rank))
        # integer part, bit-exact: edge types from GLOBAL degrees == the oracle's on the selected edges
        et_global = O.edge_types_from_degree(ei, n)
        mine = (ei[1] >= lo) & (ei[1] < hi)
        assert torch.equal(plan.local_edge_type, et_global[mine])
        # halo exchange delivers exactly the referenced remote rows, in halo order
        xl = x[lo:hi].clone().requires_grad_(True)
        xh = ctx.with_halo(xl)
        assert xh.shape[0] == plan.n_local + plan.n_halo
        assert torch.equal(xh[plan.n_local:].detach(), x[plan.halo_ids])
        # partitioned mean aggregation == rows [lo, hi) of the global one (oracle does the arithmetic)
        h_local = O.rgcn_mean_aggregate(xh, plan.local_edge_index, plan.local_edge_type, 5)[:, :plan.n_local]
        xg = x.clone().requires_grad_(True)
        h_global = O.rgcn_mean_aggregate(xg, ei, et_global, 5)
        assert torch.allclose(h_local, h_global[:, lo:hi], rtol=1e-6, atol=1e-6)
        # backward through the exchange: gradient wrt the owned rows matches the global gradient
        go = torch.randn(h_global.shape, generator=g)
        h_global.backward(go)
        h_local.backward(go[:, lo:hi])
        assert torch.allclose(xl.grad, xg.grad[lo:hi], rtol=1e-5, atol=1e-6)
        # GraphNorm statistics: all-reduced column sums == global column sums
        s = torch.stack([x[lo:hi].sum(0), (x[lo:hi] ** 2).sum(0)])
        ctx.all_reduce_sum(s)
        assert torch.allclose(s, torch.stack([x.sum(0), (x ** 2).sum(0)]), rtol=1e-5, atol=1e-4)
        # K|V all-gather and its reduce-scatter backward
        kv = x[lo:hi].unsqueeze(0).clone().requires_grad_(True)
        full = ctx.all_gather_rows(kv)
        assert torch.equal(full.detach()[0], x)
        w = torch.randn(1, n, f, generator=g) * (rank + 1)
        (full * w).sum().backward()
        gathered = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(gathered, w)
        assert torch.allclose(kv.grad[0], sum(gathered)[0, lo:hi], rtol=1e-6, atol=1e-6)
        # gradient buckets
        lin = torch.nn.Linear(f, 3)
        torch.manual_seed(1)
        for p in lin.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        ctx.all_reduce_grads(lin, bucket_bytes=16)
        for p in lin.parameters():
            assert torch.equal(p.grad, torch.full_like(p, float(sum(range(1, world + 1)))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,e,f", [(101, 700, 8), (64, 64, 4)])
def test_partition_world2_gloo(n, e, f):
    port = 29500 + (os.getpid() % 2000) + n
    mp.spawn(_worker, args=(2, port, n, e, f, 11), nprocs=2, join=True)


def test_partition_world3_uneven():
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(3, port, 100, 900, 6, 5), nprocs=3, join=True)


def test_partition_shard_files_round_trip(tmp_path):
    """Shard files (pickle-free .npz, one per rank) reproduce the plans bit for bit, the ranks' exchange counts
    mirror each other, and damaged / mismatched shards are refused."""
    import numpy as np
    from gmlm_amd import dist as gd
    n, e, world = 57, 400, 3
    g = torch.Generator().manual_seed(4)
    ei = torch.randint(0, n, (2, e), generator=g)
    paths = gd.write_partition_files(ei, n, world, str(tmp_path))
    plans = [gd.load_partition(p, world, r) for r, p in enumerate(paths)]
    for r, pl in enumerate(plans):
        ref = gd.plan_partition(ei, n, world, r)
        for k in ("local_edge_index", "local_edge_type", "halo_ids", "send_idx"):
            assert torch.equal(getattr(pl, k), getattr(ref, k)), k
        assert (pl.lo, pl.hi, pl.recv_counts, pl.send_counts, pl.active_relations) == \
               (ref.lo, ref.hi, ref.recv_counts, ref.send_counts, ref.active_relations)
    for a in range(world):
        for b in range(world):
            assert plans[a].send_counts[b] == plans[b].recv_counts[a]
    with pytest.raises(ValueError):
        gd.load_partition(paths[0], world, 1)                       # wrong rank
    with np.load(paths[1]) as z:
        bad = {k: z[k] for k in z.files}
    bad["recv_counts"] = bad["recv_counts"] + 1
    np.savez(tmp_path / "bad.npz", **bad)
    with pytest.raises(ValueError):
        gd.load_partition(str(tmp_path / "bad.npz"))
    bad["format"] = np.frombuffer(b"something-else", dtype=np.uint8)
    np.savez(tmp_path / "bad2.npz", **bad)
    with pytest.raises(ValueError):
        gd.load_partition(str(tmp_path / "bad2.npz"))


class _TorchBlock:
    """torch (float64) restatement of the per-block attention arithmetic the ring exchange is built on — the checker's
    stand-in for gmlm_amd.ops.AttentionBlock (HIP kernels), same contract: fwd -> (normalised block output, block
    log-sum-exp), bwd -> the block's share of dQ / dK / dV given the GLOBAL output and log-sum-exp."""

    def __init__(self, h, scale):
        self.h, self.scale = h, scale

    def _heads(self, t):
        b, l, c = t.shape
        return t.view(b, l, self.h, c // self.h).transpose(1, 2)                       # [b, h, l, d]

    def fwd(self, q, k, v, kv_len, seed):
        s = torch.matmul(self._heads(q), self._heads(k).transpose(-1, -2)) * self.scale
        keep = torch.arange(k.shape[1])[None, None, None, :] < kv_len.view(-1, 1, 1, 1)
        s = s.masked_fill(~keep, float("-inf"))
        lse = torch.logsumexp(s, -1)
        o = torch.matmul(torch.exp(s - lse.unsqueeze(-1)), self._heads(v))
        return o.transpose(1, 2).reshape(q.shape), lse

    def bwd(self, q, k, v, out, dout, lse, kv_len, seed):
        qh, kh, vh, oh, gh = (self._heads(t) for t in (q, k, v, out, dout))
        s = torch.matmul(qh, kh.transpose(-1, -2)) * self.scale
        keep = torch.arange(k.shape[1])[None, None, None, :] < kv_len.view(-1, 1, 1, 1)
        p = torch.exp(s - lse.unsqueeze(-1)).masked_fill(~keep, 0.0)                   # GLOBAL lse: P is the global softmax restricted to the block
        dv = torch.matmul(p.transpose(-1, -2), gh)
        delta = (gh * oh).sum(-1, keepdim=True)
        ds = p * (torch.matmul(gh, vh.transpose(-1, -2)) - delta) * self.scale
        back = lambda t: t.transpose(1, 2).reshape(t.shape[0], t.shape[2], -1)         # noqa: E731
        return back(torch.matmul(ds, kh)), back(torch.matmul(ds.transpose(-1, -2), qh)), back(dv)


def _ring_worker(rank, world, port, n, c, h):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gmlm_amd.dist import PartitionContext, plan_partition
        g = torch.Generator().manual_seed(21)
        ei = torch.randint(0, n, (2, 4 * n), generator=g)
        q_all = torch.randn(1, n, c, generator=g, dtype=torch.float64)
        kv_all = torch.randn(1, n, 2 * c, generator=g, dtype=torch.float64) * 1.5
        go_all = torch.randn(1, n, c, generator=g, dtype=torch.float64)
        ctx = PartitionContext(plan_partition(ei, n, world, rank), "cpu")
        lo, hi = ctx.plan.lo, ctx.plan.hi
        blk = _TorchBlock(h, (c // h) ** -0.5)
        # reference 1: the all-gather path on the same ranks
        q1 = q_all[:, lo:hi].clone().requires_grad_(True)
        kv1 = kv_all[:, lo:hi].clone().requires_grad_(True)
        full = ctx.all_gather_rows(kv1)
        o1, _ = blk.fwd(q1, full[..., :c], full[..., c:], torch.tensor([n]), 0)
        o1.backward(go_all[:, lo:hi])
        # reference 2: single process, all rows
        qa, kva = q_all.clone().requires_grad_(True), kv_all.clone().requires_grad_(True)
        oa, _ = blk.fwd(qa, kva[..., :c], kva[..., c:], torch.tensor([n]), 0)
        oa.backward(go_all)
        # ring
        q2 = q_all[:, lo:hi].clone().requires_grad_(True)
        kv2 = kv_all[:, lo:hi].clone().requires_grad_(True)
        o2 = ctx.ring_attention(q2, kv2, h, blk)
        o2.backward(go_all[:, lo:hi])
        for a, b_, name in ((o2, o1, "out"), (q2.grad, q1.grad, "dq"), (kv2.grad, kv1.grad, "dkv"),
                            (o2, oa[:, lo:hi], "out vs 1-process"), (q2.grad, qa.grad[:, lo:hi], "dq vs 1-process"),
                            (kv2.grad, kva.grad[:, lo:hi], "dkv vs 1-process")):
            assert torch.allclose(a.detach(), b_.detach(), rtol=1e-10, atol=1e-11), (rank, name, float((a - b_).abs().max()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 37), (3, 50), (4, 10)])
def test_ring_attention_matches_all_gather(world, n):
    """Ring K|V exchange with carried online-softmax state == K|V all-gather == one process (float64, 1e-10), forward
    and all three gradients, uneven row shares (padded blocks masked by their true length)."""
    port = 36000 + (os.getpid() % 2000) + world
    mp.spawn(_ring_worker, args=(world, port, n, 12, 3), nprocs=world, join=True)


def _bucket_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gmlm_amd.dist import PartitionContext, plan_partition
        torch.manual_seed(0)                                    # same parameters on every rank (replicated)

        ctx = PartitionContext(plan_partition(torch.randint(0, 8, (2, 16)), 8, world, rank), "cpu")

        class _SumOverRanksInBackward(torch.autograd.Function):
            """Stands for the collectives the real model's backward issues on the partition group (GraphNorm statistics,
            halo all-to-all, basis reducer): identity forward, all-reduce of a statistic in backward.  ``live`` = False
            (the reference run) skips the collective."""

            @staticmethod
            def forward(fctx, x, live):
                fctx.live = live
                return x.view_as(x)

            @staticmethod
            def backward(fctx, g):
                if fctx.live:
                    ctx.all_reduce_sum(g.sum(0, keepdim=True).clone())      # on part.group, in the middle of backward
                return g, None

        class Net(torch.nn.Module):
            def __init__(self, live=True):
                super().__init__()
                self.a, self.b, self.c = torch.nn.Linear(6, 5), torch.nn.Linear(5, 4), torch.nn.Linear(4, 3)
                self.text = torch.nn.Linear(6, 3)               # used only by ranks that have "active text rows"
                self.dead = torch.nn.Linear(3, 3)               # never used by anyone: must keep grad = None
                self.live = live

            def forward(self, x, use_text, use_gnn=True):
                y = x.new_zeros(x.shape[0], 3)
                if use_gnn:
                    h = _SumOverRanksInBackward.apply(torch.relu(self.a(x)), self.live)
                    h = _SumOverRanksInBackward.apply(torch.relu(self.b(h)), self.live)
                    y = self.c(h)
                # the text branch is created LAST: its gradients are final FIRST in backward, i.e. on a rank that has it
                # its bucket launches ahead of the collectives above, on an idle rank only in finish()
                return y + self.text(x) if use_text else y

        net = Net()
        gb = ctx.grad_buckets(net, bucket_bytes=64)             # several tiny buckets
        assert gb.group is not ctx.group                        # bucket all-reduces have a communicator of their own
        g = torch.Generator().manual_seed(100 + rank)
        for step in range(6):
            x = torch.randn(7, 6, generator=g)
            # rank 0 always has text rows; the others only on step 3: steps 1, 2, 4 run with the learned bucket set while
            # the ranks' buckets become final at different points of backward (collective-order mismatch on a shared group).
            # Step 5: a "pre-training" step that touches neither the text branch on any rank (its parameters then have
            # grad None, like the single-GPU run, although they are in the learned set)
            use_text = (rank == 0 or step == 3) and step != 5
            gb.prepare()
            net(x, use_text).square().sum().backward()
            gb.finish()
            # the same sum computed the plain way
            ref = Net(live=False)
            ref.load_state_dict(net.state_dict())
            ref(x, use_text).square().sum().backward()
            for (k, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
                local = torch.zeros_like(pr) if pr.grad is None else pr.grad.clone()
                dist.all_reduce(local)
                if k.startswith("dead") or (step == 5 and k.startswith("text")):
                    assert p.grad is None, (step, k)
                else:
                    assert p.grad is not None and torch.allclose(p.grad, local, rtol=1e-6, atol=1e-6), (rank, step, k)
            if 1 <= step <= 4:                                  # learned set: gradients are views of the flat buckets (no copies)
                assert all(p.grad.data_ptr() >= f.data_ptr() and p.grad.data_ptr() < f.data_ptr() + f.numel() * 4
                           for i, p in enumerate(gb.params) if gb.expected[i] for f in [gb.flats[gb.bucket_of[i]]])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_grad_buckets_overlap_matches_plain_sum(world):
    """Flat-buffer gradient buckets launched from backward hooks, next to OTHER collectives inside backward (as the real
    model has): same sums as a plain all-reduce when the ranks' gradients become final at different times / not at all
    (a rank that never touches the text branch launches those buckets after the in-backward collectives, an active
    rank before them: the buckets' own communicator keeps that from being a collective-order mismatch), over several
    steps with the learned set; a parameter no rank uses keeps grad = None, and so does a learned parameter in a step
    where no rank produces its gradient."""
    port = 38000 + (os.getpid() % 2000) + world
    mp.spawn(_bucket_worker, args=(world, port), nprocs=world, join=True)


def _halo_overlap_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gmlm_amd.dist import PartitionContext, plan_partition
        n, e, f = 83, 600, 5
        g = torch.Generator().manual_seed(3)
        ei = torch.randint(0, n, (2, e), generator=g)
        x = torch.randn(n, f, generator=g)
        ctx = PartitionContext(plan_partition(ei, n, world, rank), "cpu")
        lo, hi = ctx.plan.lo, ctx.plan.hi
        outs = []
        wgt = torch.randn(n + 50, f, generator=g)
        for overlap in (False, True, "node"):
            xl = x[lo:hi].clone().requires_grad_(True)
            xh = ctx.with_halo(xl, defer=bool(overlap))
            local_work = xl @ torch.ones(f, 2)                  # what RGCNConv does meanwhile: the root GEMM on owned rows
            assert (len(ctx._halo_pending) > 0) == bool(overlap)   # deferred only when asked
            if overlap == "node":
                # the autograd-node form RGCNConv.forward_csr uses: forward waits here; BACKWARD starts the reverse
                # exchange here (ahead of local_work's backward) and finishes it in with_halo's node
                xh = ctx.halo_ready(xh)
                assert not ctx._halo_pending
            else:
                ctx.wait_halo()
            assert torch.equal(xh[ctx.plan.n_local:].detach(), x[ctx.plan.halo_ids])
            ((xh * wgt[:xh.shape[0]]).sum() * 2 + local_work.sum()).backward()
            assert not ctx._halo_bwd                            # the started reverse exchange was consumed
            outs.append((xh.detach().clone(), xl.grad.clone()))
        for o in outs[1:]:
            assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])   # bit for bit
        # and the gradient is the right one: d/dx_i of sum_r 2 * wgt * (rows every rank took from x)
        gathered = [torch.zeros(n, f) for _ in range(world)]
        mine = torch.zeros(n, f)
        mine[lo:hi] += 2 * wgt[:hi - lo]
        mine.index_add_(0, ctx.plan.halo_ids, 2 * wgt[hi - lo:hi - lo + ctx.plan.n_halo])
        dist.all_gather(gathered, mine)
        want = sum(gathered)[lo:hi] + 2.0                       # + local_work's gradient
        assert torch.allclose(outs[2][1], want, rtol=1e-6, atol=1e-6)
    finally:
        dist.destroy_process_group()


def test_halo_exchange_deferred_wait_is_bit_identical():
    port = 39000 + (os.getpid() % 2000)
    mp.spawn(_halo_overlap_worker, args=(3, port), nprocs=3, join=True)


def _world8_worker(rank, world, port):
    """World-size-8 rehearsal of what the first 8-GPU run executes: plan with uneven shares and a rank
    without any halo rows, halo exchange through the autograd nodes, and the ring K|V exchange with the overlapped
    dK|dV hop against one process."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        import gmlm_oracle as O
        from gmlm_amd.dist import PartitionContext, plan_partition, row_range
        n, f = 83, 6                                            # 83 = 8 * 10 + 3: ranks 0-2 own 11 rows, the others 10
        g = torch.Generator().manual_seed(17)
        ei = torch.randint(0, n, (2, 900), generator=g)
        lo7, hi7 = row_range(n, world, 7)
        keep = ~((ei[1] >= lo7) & (ei[1] < hi7) & ((ei[0] < lo7) | (ei[0] >= hi7)))   # rank 7: every in-edge comes from its own rows
        ei = ei[:, keep]
        x = torch.randn(n, f, generator=g)
        plan = plan_partition(ei, n, world, rank)
        ctx = PartitionContext(plan, "cpu")
        lo, hi = plan.lo, plan.hi
        assert hi - lo == (11 if rank < 3 else 10)
        if rank == 7:
            assert plan.n_halo == 0 and sum(plan.recv_counts) == 0
        counts = [torch.zeros(world, dtype=torch.long) for _ in range(world)]
        dist.all_gather(counts, torch.tensor(plan.send_counts))
        assert [int(counts[p][rank]) for p in range(world)] == plan.recv_counts      # the ranks' plans mirror each other
        et_global = O.edge_types_from_degree(ei, n)
        xl = x[lo:hi].clone().requires_grad_(True)
        xh = ctx.halo_ready(ctx.with_halo(xl, defer=True))
        h_local = O.rgcn_mean_aggregate(xh, plan.local_edge_index, plan.local_edge_type, 5)[:, :plan.n_local]
        xg = x.clone().requires_grad_(True)
        h_global = O.rgcn_mean_aggregate(xg, ei, et_global, 5)
        assert torch.allclose(h_local, h_global[:, lo:hi], rtol=1e-6, atol=1e-6)
        go = torch.randn(h_global.shape, generator=g)
        h_global.backward(go)
        h_local.backward(go[:, lo:hi])
        assert torch.allclose(xl.grad, xg.grad[lo:hi], rtol=1e-5, atol=1e-6)
        # ring K|V exchange at world 8 (7 hops, uneven blocks) == one process
        c, h = 12, 3
        q_all = torch.randn(1, n, c, generator=g, dtype=torch.float64)
        kv_all = torch.randn(1, n, 2 * c, generator=g, dtype=torch.float64) * 1.5
        go_all = torch.randn(1, n, c, generator=g, dtype=torch.float64)
        blk = _TorchBlock(h, (c // h) ** -0.5)
        qa, kva = q_all.clone().requires_grad_(True), kv_all.clone().requires_grad_(True)
        oa, _ = blk.fwd(qa, kva[..., :c], kva[..., c:], torch.tensor([n]), 0)
        oa.backward(go_all)
        q2 = q_all[:, lo:hi].clone().requires_grad_(True)
        kv2 = kv_all[:, lo:hi].clone().requires_grad_(True)
        o2 = ctx.ring_attention(q2, kv2, h, blk)
        o2.backward(go_all[:, lo:hi])
        for a, b_, name in ((o2, oa[:, lo:hi], "out"), (q2.grad, qa.grad[:, lo:hi], "dq"), (kv2.grad, kva.grad[:, lo:hi], "dkv")):
            assert torch.allclose(a.detach(), b_.detach(), rtol=1e-10, atol=1e-11), (rank, name, float((a - b_).abs().max()))
    finally:
        dist.destroy_process_group()


def test_world8_plan_halo_and_ring():
    port = 40500 + (os.getpid() % 2000)
    mp.spawn(_world8_worker, args=(8, port), nprocs=8, join=True)
