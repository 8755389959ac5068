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
