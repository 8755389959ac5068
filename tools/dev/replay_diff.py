"""Which gradient tensors differ between the eager bucketed step and a recorded step (diagnostic for the open issue of the
branched whole-step graph).
usage: [BF16=1] python tools/dev/replay_diff.py enc|whole|ee|gg [mask order, e.g. 01201201]
  enc   eager vs three linear recordings     whole  eager vs the whole-step graph with branches
  ee    eager vs eager (never differs)       gg     whole-step graph vs whole-step graph (differs sporadically)
  conc  eager vs the linear recordings with GNN and encoder replayed on two streams at once"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
import torch
if os.environ.get("GMLM_LIB"):                       # A/B: another build of the library
    import gmlm_amd._lib as _L
    _L.LIB_PATH = os.environ["GMLM_LIB"]
import test_gpu_graphs as T
from test_gpu_model import build_model

dev = torch.device("cuda:0")
cd = torch.bfloat16 if os.environ.get("BF16") else torch.float32
mode = sys.argv[1] if len(sys.argv) > 1 else "enc"
cfg = T._cfg(0.0)
x, ei, y, tokens, masks = T._data(cfg, dev)
eager = build_model(cfg, dev, compute_dtype=cd).train()
eager.plm_bucketed = True
if mode == "gg":
    eager.capture_hip_graphs(eager.soft_mask_input(x, masks[0], 0.7), ei, encoder=True, whole_step=True)
graphed = build_model(cfg, dev, compute_dtype=cd).train()
if mode == "ee":
    graphed.plm_bucketed = True
    class _G: _encoders = {}; _steps = {}
    g = _G()
else:
    g = graphed.capture_hip_graphs(graphed.soft_mask_input(x, masks[0], 0.7), ei, encoder=True, whole_step=(mode in ("whole", "gg")),
                                   concurrent=(mode == "conc"))
order = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else '0120')]
seen_eager = {}            # mask id -> eager gradients (deterministic per mask): is a wrong replay value a STALE one of another step?
prev_replay = None
for step, mask in enumerate([masks[i] for i in order]):
    l0, g0 = T._step(eager, x, ei, y, tokens, mask, 512)
    l1, g1 = T._step(graphed, x, ei, y, tokens, mask, 512)
    bad = [(k, float((g0[k] - g1[k]).abs().max()), float(g0[k].abs().max()), float(g1[k].abs().max())) for k in g0 if not torch.equal(g0[k], g1[k])]
    if bad or step % 50 == 0:
        print("   buckets:", [k[1:] for k in list(g._encoders) + list(g._steps)])
        print(f"step {step}: logits equal {torch.equal(l0, l1)}; {len(bad)} of {len(g0)} gradient tensors differ", flush=True)
    for k, d, a, b in bad[:8]:
        print(f"   {k:70s} max|diff| {d:.3e}  eager max {a:.3e}  replay max {b:.3e}")
        e, r = g0[k].flatten(), g1[k].flatten()
        for i in (e != r).nonzero().flatten().tolist()[:40]:
            others = {m: float(v[k].flatten()[i]) for m, v in seen_eager.items() if m != order[step]}
            stale = [m for m, v in others.items() if v == float(r[i])]
            pr = float(prev_replay[k].flatten()[i]) if prev_replay is not None else None
            print(f"      [{i}] eager {float(e[i]):+.9e} replay {float(r[i]):+.9e}  previous replay {pr}  equals eager value of mask {stale or None}; other masks: {others}")
    seen_eager[order[step]] = g0
    prev_replay = g1
