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


def _worker(rank, world, port, out_dir):
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
        plm = dict(hidden=128, layers=2, heads=2, inter=256, max_pos=64, vocab=200)
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
        part = attach_partition(model, ei, n, dev)
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


def test_partitioned_model_matches_single_gpu(tmp_path):
    port = 33000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
