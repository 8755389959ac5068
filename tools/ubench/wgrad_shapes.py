"""Split-K weight-gradient GEMM forms for the BERT shapes of the bench (T = 91,698 tokens, S = 16 slices):
fp32-output batched GEMM (what bert._splitk_wgrad issues; not covered by TunableOp) against the bf16-output batched form
that TunableOp can tune.  usage: [PYTORCH_TUNABLEOP_ENABLED=1] [T=tokens] [GEOMETRY=base|mini] python tools/ubench/wgrad_shapes.py [S,S,...]"""
import torch
dev = torch.device("cuda")
import sys
import os
T = int(os.environ.get("T", "91712"))
SLICES = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (8, 16, 32)

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

SHAPES = {"base": (("qkv", 2304, 768), ("o", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072)),
          "mini": (("qkv", 768, 256), ("o", 256, 256), ("ffn1", 1024, 256), ("ffn2", 256, 1024))}[os.environ.get("GEOMETRY", "base")]   # BERT-base / BERT-mini (Chameleon-size workload)
for name, n, k in SHAPES:
  dy = torch.randn(T, n, device=dev, dtype=torch.bfloat16)
  x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
  for S in SLICES:
    q = T // S
    a = dy[:S * q].view(S, q, n).transpose(1, 2)
    b = x[:S * q].view(S, q, k)
    fl = 2.0 * S * q * n * k
    t32 = timeit(lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0))
    t16 = timeit(lambda: torch.bmm(a, b).float().sum(0))
    t32g = timeit(lambda: torch.bmm(a, b, out_dtype=torch.float32))
    t16g = timeit(lambda: torch.bmm(a, b))
    print(f"{name:5s} S={S:2d} f32-out {t32:7.1f} us ({fl / t32 / 1e6:6.1f} TF/s; GEMM alone {t32g:7.1f})   bf16-out {t16:7.1f} us ({fl / t16 / 1e6:6.1f} TF/s; GEMM alone {t16g:7.1f})")
