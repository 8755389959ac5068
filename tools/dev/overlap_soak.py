"""Eager single-stream step vs ``model.overlap_streams`` (text encoder on a second HIP stream): bitwise comparison of logits
and every gradient over many steps.  usage: [BF16=1] python tools/dev/overlap_soak.py [steps]"""
import sys, os
here = os.path.dirname(os.path.abspath(__file__))
for p in ("../../tests", "../..", "../../oracle"):
    sys.path.insert(0, os.path.join(here, p))
import torch
import test_gpu_graphs as T
from test_gpu_model import build_model

dev = torch.device("cuda:0")
cd = torch.bfloat16 if os.environ.get("BF16") else torch.float32
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg = T._cfg(0.0)
x, ei, y, tokens, masks = T._data(cfg, dev)
a = build_model(cfg, dev, compute_dtype=cd).train()
b = build_model(cfg, dev, compute_dtype=cd).train()
b.overlap_streams = True
bad = 0
for s in range(steps):
    m = masks[s % len(masks)]
    l0, g0 = T._step(a, x, ei, y, tokens, m, 512)
    l1, g1 = T._step(b, x, ei, y, tokens, m, 512)
    off = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    if off or not torch.equal(l0, l1):
        bad += 1
        print(f"step {s}: logits equal {torch.equal(l0, l1)}; differing:", [(k, float((g0[k] - g1[k]).abs().max() / (g0[k].abs().max() + 1e-30))) for k in off][:4])
print(f"{steps} steps, {bad} with a difference")
