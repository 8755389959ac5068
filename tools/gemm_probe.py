"""Probe: wgrad GEMM (dW = dY^T X, K = tokens) as one hipBLASLt call vs split-K via bmm."""
import torch, time
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for T, N, K in ((32768, 768, 768), (32768, 2304, 768), (32768, 3072, 768), (32768, 768, 3072), (26880, 3072, 768), (8388, 3072, 768)):
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16); dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * N * K
    base = t(lambda: dy.t() @ x)
    line = f"T={T} N={N} K={K}: wgrad {base:.3f} ms {fl/base/1e9:.0f} TF |"
    for S in (4, 8, 16, 32):
        if T % S: continue
        def sk():
            return torch.bmm(dy.view(S, T // S, N).transpose(1, 2), x.view(S, T // S, K)).float().sum(0)
        ms = t(sk); line += f" S={S}: {ms:.3f} ({fl/ms/1e9:.0f} TF)"
    try:
        S = 8
        ms = t(lambda: torch.bmm(dy.view(S, T // S, N).transpose(1, 2), x.view(S, T // S, K), out_dtype=torch.float32).sum(0))
        line += f" | S=8 f32out {ms:.3f}"
    except Exception as e:
        line += f" | f32out unsupported ({type(e).__name__})"
    fwd = t(lambda: x @ w.t()); dgr = t(lambda: dy @ w)
    line += f" | fwd {fwd:.3f} ({fl/fwd/1e9:.0f} TF) dgrad {dgr:.3f} ({fl/dgr/1e9:.0f} TF)"
    print(line, flush=True)
