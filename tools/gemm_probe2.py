"""Probe: split-K wgrad orientation / slice count at the packed token count."""
import torch, time
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
T = 90496
for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16); dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * N * K
    p = t(lambda: torch.mm(dy.t(), x)); q = t(lambda: torch.mm(x.t(), dy))
    print(f"N={N} K={K}: plain mm dyT@x {p:.3f} ({fl/p/1e9:.0f}TF)  xT@dy {q:.3f} ({fl/q/1e9:.0f}TF)", flush=True)
    for S in (4, 8, 16, 32, 64, 128):
        if T % S: continue
        a = t(lambda: torch.bmm(dy.view(S, T // S, N).transpose(1, 2), x.view(S, T // S, K), out_dtype=torch.float32).sum(0))
        b = t(lambda: torch.bmm(x.view(S, T // S, K).transpose(1, 2), dy.view(S, T // S, N), out_dtype=torch.float32).sum(0))
        print(f"   S={S}: dyT@x {a:.3f} ({fl/a/1e9:.0f}TF) xT@dy {b:.3f} ({fl/b/1e9:.0f}TF)", flush=True)
