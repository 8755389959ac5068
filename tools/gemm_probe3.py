"""Probe: forward / data-gradient GEMM layouts at the packed token count (bf16)."""
import torch, time
import torch.nn.functional as F
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for T in (91698, 91648, 90112):
    print("T", T)
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
        x = torch.randn(T, K, device=dev, dtype=torch.bfloat16); dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02; b = torch.randn(N, device=dev, dtype=torch.bfloat16)
        wt = w.t().contiguous()
        fl = 2.0 * T * N * K
        r = {}
        r["fwd linear(x,w,b)"] = t(lambda: F.linear(x, w, b))
        r["fwd linear(x,w)"] = t(lambda: F.linear(x, w))
        r["fwd x@wt_contig"] = t(lambda: x @ wt)
        r["dgrad dy@w"] = t(lambda: dy @ w)
        r["dgrad linear(dy,wt)"] = t(lambda: F.linear(dy, wt))
        print(f" N={N} K={K}: " + " | ".join(f"{k} {v:.3f} ({fl/v/1e9:.0f}TF)" for k, v in r.items()), flush=True)
