"""Where a launch-bound small-config step spends its wall time: whole step, GNN part (get_graph_embeddings fwd+bwd) alone,
and the kernel-busy time of each (torch profiler off; busy time from a CUDA-event bracket after a device-side idle).
usage: python tools/ubench/small_step_split.py [workload] [hc] [plm_hidden] [plm_layers]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
import bench, gmlm_amd

wl = sys.argv[1] if len(sys.argv) > 1 else "chameleon"
args = types.SimpleNamespace(vocab=30522, plm_hidden=int(sys.argv[3]) if len(sys.argv) > 3 else 256,
                             plm_layers=int(sys.argv[4]) if len(sys.argv) > 4 else 4, max_len=128, dtype="bf16",
                             hc=int(sys.argv[2]) if len(sys.argv) > 2 else 256, plm_ckpt=False, workload=wl, plm_batch=4096)
dev = torch.device("cuda")
data = bench.synthetic(wl)
ids, am = bench.synthetic_tokens(data["n"], args.max_len, args.vocab, seed=data["n"])
model = bench.build_model(args, data, dev)
x, y, active, ei = data["x"].to(dev), data["y"].to(dev), data["active"].to(dev), data["edge_index"].to(dev)
tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
n_act = int(data["active"].sum())

def full():
    model.zero_grad(set_to_none=True)
    xm = model.soft_mask_input(x, active, 0.7)
    logits = model(xm, ei, tokens, active, plm_batch_size=4096)
    idx = model.active_index
    loss = F.cross_entropy(logits.index_select(0, idx), y.index_select(0, idx), label_smoothing=0.2, reduction="sum") / n_act
    loss.backward()

def gnn():
    model.zero_grad(set_to_none=True)
    out = model.get_graph_embeddings(model.soft_mask_input(x, active, 0.7), ei)
    out.float().square().mean().backward()

def wall(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def host_only(f, n=20):
    """host time to ENQUEUE a step (no sync inside): the launch-bound part"""
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3

print(f"{wl}: N={data['n']} active={n_act} hc={args.hc} plm={args.plm_hidden}x{args.plm_layers}")
print(f"full step  wall {wall(full):7.2f} ms   host-enqueue {host_only(full):7.2f} ms")
print(f"GNN only   wall {wall(gnn):7.2f} ms   host-enqueue {host_only(gnn):7.2f} ms")
from torch.profiler import profile, ProfilerActivity
for name, f in (("full", full), ("gnn", gnn)):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5): f()
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    busy = sum(e.device_time for e in ev) / 5 / 1e3
    print(f"{name}: {len(ev) / 5:.0f} device activities/step, kernel-busy {busy:.2f} ms/step")
