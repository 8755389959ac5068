"""2-rank rehearsal of the 1-D node partition ON ONE GPU (gloo backend staging through the host,
both ranks on cuda:0): partitioned forward/backward through the HIP kernels must reproduce the
single-GPU result (logits rows [lo, hi) and the all-reduced parameter gradients)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _build(cfg, dev):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from test_gpu_model import build_model
    return build_model(cfg, dev)


def _worker(rank, world, port, out_dir, ring=False, shards=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gmlm_amd
        import gmlm_oracle as O
        from gmlm_amd.dist import attach_partition
        dev = torch.device("cuda:0")
        # ring: P = 512 so that CrossAttention's 8 heads have the native head dim 64 the ring blocks need
        plm = dict(hidden=512 if ring else 128, layers=1 if ring else 2, heads=8 if ring else 2, inter=256, max_pos=64, vocab=200)
        n, e = 301, 2500
        cfg = dict(n=n, e=e, f_in=40, hc=32, c=5, plm=plm, seed=77)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(n, 40, generator=g)
        ei = torch.randint(0, n, (2, e), generator=g)
        y = torch.randint(0, 5, (n,), generator=g)
        mask = torch.rand(n, generator=g) < 0.5
        ids, am = O.synthetic_tokens(n, 12, 200, 5, 2)
        n_act = int(mask.sum())

        def run(model, lo, hi, part):
            tokens = gmlm_amd.TokenizedTexts.from_mask(ids[lo:hi].to(dev), am[lo:hi].to(dev))
            mk = mask[lo:hi].to(dev)
            xm = model.soft_mask_input(x[lo:hi].to(dev), mk, 0.7)
            logits = model(xm, ei.to(dev), tokens, mk, plm_batch_size=64)
            loss = F.cross_entropy(logits[mk], y[lo:hi].to(dev)[mk], label_smoothing=0.2, reduction="sum") / n_act
            loss.backward()
            if part is not None:
                part.all_reduce_grads(model)
            return logits.detach().cpu(), {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}

        ref_model = _build(cfg, dev).train()
        ref_logits, ref_grads = run(ref_model, 0, n, None)
        model = _build(cfg, dev).train()
        if shards:                                              # partitioned offline into shard files; each rank reads only its own
            from gmlm_amd.dist import write_partition_files
            pdir = os.path.join(out_dir, "parts")
            if rank == 0:
                write_partition_files(ei, n, world, pdir)
            dist.barrier()
            part = attach_partition(model, None, n, dev, partition_dir=pdir)
        else:
            part = attach_partition(model, ei, n, dev)
        part.use_ring = ring                                    # K|V blocks travel rank to rank instead of the all-gather
        lo, hi = part.plan.lo, part.plan.hi
        logits, grads = run(model, lo, hi, part)
        err = (logits - ref_logits[lo:hi]).abs().max().item()
        assert err < 2e-5, f"rank {rank}: partitioned logits differ by {err}"
        for k, gr in ref_grads.items():
            gn = float(gr.norm())
            d = float((grads[k] - gr).norm())
            assert d <= 2e-4 * max(gn, 1e-3), (rank, k, d, gn)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ring,shards", [(False, False), (True, False), (False, True)])
def test_partitioned_model_matches_single_gpu(tmp_path, ring, shards):
    """Halo exchange with the deferred wait (root GEMM under the all-to-all), GraphNorm all-reduce, K|V all-gather
    or ring exchange through the HIP attention kernels: partitioned logits / gradients == single GPU.  shards: the
    partition comes from part-RRRRR-of-WWWWW.npz files instead of the replicated edge list."""
    port = 33000 + (os.getpid() % 2000) + (7 if ring else 0) + (13 if shards else 0)
    mp.spawn(_worker, args=(2, port, str(tmp_path), ring, shards), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _harness_worker(rank, world, port, out_dir):
    """harness.train_step / pretrain_step under the node partition, with an UNBALANCED active mask: rank 1 owns no
    active node at all.  The step must not hang (collective skip decisions), the loss must be the global mean and
    the clipped, all-reduced gradients must equal the single-GPU step's."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gmlm_amd
        import gmlm_oracle as O
        from gmlm_amd import harness
        from gmlm_amd.dist import attach_partition
        dev = torch.device("cuda:0")
        plm = dict(hidden=128, layers=2, heads=2, inter=256, max_pos=64, vocab=200)
        n, e = 257, 2100
        cfg = dict(n=n, e=e, f_in=40, hc=32, c=5, plm=plm, seed=78)
        g = torch.Generator().manual_seed(4)
        x = torch.randn(n, 40, generator=g)
        ei = torch.randint(0, n, (2, e), generator=g)
        y = torch.randint(0, 5, (n,), generator=g)
        mask = torch.zeros(n, dtype=torch.bool)
        mask[torch.randperm(n // 2 - 4, generator=g)[:37]] = True          # every active node in rank 0's rows
        ids, am = O.synthetic_tokens(n, 12, 200, 5, 2)

        def opt_for(m):
            return harness.setup_optimizer(m, 1e-3, 1e-5, 1e-4, 0.01)

        ref = _build(cfg, dev)
        ref_opt = opt_for(ref)
        tok = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
        r0 = harness.train_step(ref, ref_opt, None, x.to(dev), ei.to(dev), tok, y.to(dev), mask.to(dev), autocast=False)
        ref_grads = {k: p.grad.detach().cpu() for k, p in ref.named_parameters() if p.grad is not None}
        model = _build(cfg, dev)
        opt = opt_for(model)
        part = attach_partition(model, ei, n, dev)
        lo, hi = part.plan.lo, part.plan.hi
        assert rank == 0 or int(mask[lo:hi].sum()) == 0                    # the case under test
        tok_l = gmlm_amd.TokenizedTexts.from_mask(ids[lo:hi].to(dev), am[lo:hi].to(dev))
        r1 = harness.train_step(model, opt, None, x[lo:hi].to(dev), ei.to(dev), tok_l, y[lo:hi].to(dev), mask[lo:hi].to(dev),
                                autocast=False)
        assert not r1.skipped and abs(r1.loss - r0.loss) < 2e-5 and abs(r1.accuracy - r0.accuracy) < 1e-6, (r1, r0)
        grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
        assert set(grads) == set(ref_grads), set(grads) ^ set(ref_grads)   # dead branch / pooler keep grad = None on every rank
        for k, gr in ref_grads.items():
            gn = float(gr.norm())
            assert float((grads[k] - gr).norm()) <= 3e-4 * max(gn, 1e-3), (rank, k)
        # steps 2-4 run with the LEARNED bucket set: rank 0's PLM / head buckets become final (and launch) ahead of the GNN
        # backward's collectives, rank 1 (no active node, no PLM gradient) launches them in finish() - the order mismatch
        # that needs the buckets' own communicator; the trajectories must stay on the single-GPU ones
        for it in range(3):
            ra = harness.train_step(ref, ref_opt, None, x.to(dev), ei.to(dev), tok, y.to(dev), mask.to(dev), autocast=False)
            rb = harness.train_step(model, opt, None, x[lo:hi].to(dev), ei.to(dev), tok_l, y[lo:hi].to(dev), mask[lo:hi].to(dev),
                                    autocast=False)
            assert not rb.skipped and abs(rb.loss - ra.loss) < 5e-5 * max(1.0, abs(ra.loss)), (it, rb, ra)
        # no active node anywhere: every rank skips (and none is left waiting in a collective)
        none = torch.zeros(hi - lo, dtype=torch.bool, device=dev)
        assert harness.train_step(model, opt, None, x[lo:hi].to(dev), ei.to(dev), tok_l, y[lo:hi].to(dev), none, autocast=False).skipped
        # pre-training step: global NT-Xent over the gathered embeddings == the single-GPU loss
        m1 = torch.rand(n, generator=g) < 0.4
        m2 = torch.rand(n, generator=g) < 0.4
        lp_ref = harness.pretrain_step(ref, ref_opt, x.to(dev), ei.to(dev), m1.to(dev), m2.to(dev), autocast=False)
        lp = harness.pretrain_step(model, opt, x[lo:hi].to(dev), ei.to(dev), m1[lo:hi].to(dev), m2[lo:hi].to(dev), autocast=False)
        assert abs(lp - lp_ref) < 5e-4 * max(1.0, abs(lp_ref)), (lp, lp_ref)
        # the pre-training step produces no PLM / head / cross-attention gradient on any rank: those parameters are in the
        # learned bucket set (train_step ran first) and must still end with grad None, like the single-GPU step, so that
        # AdamW does not decay them or move their moments
        none_ref = {k for k, p in ref.named_parameters() if p.grad is None}
        none_got = {k for k, p in model.named_parameters() if p.grad is None}
        assert none_got == none_ref and any(k.startswith("plm_encoder.") for k in none_got), none_got ^ none_ref
        open(os.path.join(out_dir, f"hok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_harness_steps_under_partition_with_an_idle_rank(tmp_path):
    port = 35000 + (os.getpid() % 2000)
    mp.spawn(_harness_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "hok0").exists() and (tmp_path / "hok1").exists()
