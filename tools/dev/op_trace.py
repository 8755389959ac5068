"""Which autograd nodes / aten parents launch a given torch op in one eager step.
usage: python tools/dev/op_trace.py [aten op substring, default fill_] [workload, default chameleon]"""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, torch.nn.functional as F
import bench
want = sys.argv[1] if len(sys.argv) > 1 else "fill_"
wl = sys.argv[2] if len(sys.argv) > 2 else "chameleon"
geo = {"chameleon": (256, 256, 4), "cornell": (512, 768, 12)}.get(wl, (768, 768, 12))
sys.argv = ["bench.py", "--workload", "chameleon"]
import argparse
# reuse bench's builders
ap_defaults = dict(workload=wl, hc=geo[0], plm_hidden=geo[1], plm_layers=geo[2], vocab=30522, max_len=128, dtype="bf16", plm_batch=4096,
                   reference_recompute=False, overlap_streams=False, plm_ckpt=False, activation_ckpt=False, act_ckpt=False)
args = argparse.Namespace(**ap_defaults)
dev = torch.device("cuda:0")
data = bench.synthetic(wl, n_parts=1)
ids, am = bench.synthetic_tokens(data["n"], 128, 30522, seed=data["n"])
model = bench.build_model(args, data, dev)
import gmlm_amd
x, y, active, ei = data["x"].to(dev), data["y"].to(dev), data["active"].to(dev), data["edge_index"].to(dev)
tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))

def step():
    model.zero_grad(set_to_none=True)
    logits = model(model.soft_mask_input(x, active, 0.7), ei, tokens, active, plm_batch_size=4096)
    idx = model.active_index
    F.cross_entropy(logits.index_select(0, idx), y.index_select(0, idx), label_smoothing=0.2).backward()

for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if want in ev.name:
        chain, p = [], ev.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name[:60])
            p = p.cpu_parent
        shape = ""
        cnt[(ev.name, " <- ".join(chain))] += 1
for (name, chain), c in cnt.most_common(30):
    print(c, name, "<-", chain)
