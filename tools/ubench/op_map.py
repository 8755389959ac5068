"""Which torch op launches which small kernel in the bench step (torch.profiler, one step): prints aten ops with their CUDA time.
usage: python tools/ubench/op_map.py"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
import bench, gmlm_amd
from torch.profiler import profile, ProfilerActivity
args = types.SimpleNamespace(vocab=30522, plm_hidden=768, plm_layers=12, max_len=128, dtype="bf16", hc=768, plm_ckpt=False, workload="squirrel", plm_batch=4096)
dev = torch.device("cuda")
data = bench.synthetic("squirrel")
ids, am = bench.synthetic_tokens(data["n"], 128, 30522, seed=data["n"])
model = bench.build_model(args, data, dev)
x, y, active, ei = data["x"].to(dev), data["y"].to(dev), data["active"].to(dev), data["edge_index"].to(dev)
tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
n_act = int(data["active"].sum())
def step():
    model.zero_grad(set_to_none=True)
    lg = model(model.soft_mask_input(x, active, 0.7), ei, tokens, active, plm_batch_size=4096)
    idx = model.active_index
    (F.cross_entropy(lg.index_select(0, idx), y.index_select(0, idx), label_smoothing=0.2, reduction="sum") / n_act).backward()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total / 1e3) for e in prof.key_averages() if e.self_device_time_total > 0]
rows.sort(key=lambda r: -r[2])
for k, c, t in rows[:45]:
    print(f"{t:8.3f} ms {c:5d} calls  {k[:90]}")
