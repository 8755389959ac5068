"""Probe: packed (variable-length) BERT attention at the bench's sequence-length mix, with and without dropout."""
import sys, torch
sys.path.insert(0, ".")
from gmlm_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
def run(lens, p, label, h=12, d=64):
    cu = torch.zeros(lens.numel() + 1, dtype=torch.int32)
    cu[1:] = lens.cumsum(0)
    T = int(cu[-1])
    qkv = (torch.randn(T, 3 * h * d, generator=g) * 0.5).to(dev, torch.bfloat16).requires_grad_(True)
    cud = cu.to(dev)
    go = torch.randn(T, h * d, device=dev, dtype=torch.bfloat16)
    mx = int(lens.max())
    def fwd(): return ops.attention_qkv(qkv, None, h, d ** -0.5, p, True, cud, mx)
    def ev(fn, n=10):
        for _ in range(2): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
    tf = ev(fwd)
    def fb():
        qkv.grad = None
        fwd().backward(go)
    tb = ev(fb) - tf
    traffic_f, traffic_b = 4 * T * h * d * 2, 8 * T * h * d * 2
    print(f"{label}: T={T} p={p}: fwd {tf:.0f} us ({traffic_f/tf/1e6:.2f} TB/s of q,k,v,o)  bwd {tb:.0f} us ({traffic_b/tb/1e6:.2f} TB/s of 8 tensors)", flush=True)
mix = torch.randint(16, 129, (1257,), generator=g)
for p in (0.0, 0.1):
    run(mix, p, "mix 16..128 x1257")
    run(torch.full((1257,), 128), p, "all 128 x1257")
    run(torch.full((2514,), 64), p, "all 64 x2514")
    run(torch.full((5028,), 32), p, "all 32 x5028")
