"""Per-shape timing of the PLM GEMMs of the bench workload (T = 91,698 packed tokens, BERT-base), as torch issues them,
optionally under TunableOp (PYTORCH_TUNABLEOP_ENABLED=1) to see what the library could reach with another algorithm.
usage: python tools/ubench/gemm_shapes.py [T]"""
import sys, time, torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 91698
dev = torch.device("cuda")
bf = torch.bfloat16

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

rows = []
for name, k, n in (("qkv", 768, 2304), ("o", 768, 768), ("ffn1", 768, 3072), ("ffn2", 3072, 768)):
    x = torch.randn(T, k, device=dev, dtype=bf); w = torch.randn(n, k, device=dev, dtype=bf) * 0.02
    b = torch.randn(n, device=dev, dtype=bf)
    dy = torch.randn(T, n, device=dev, dtype=bf)
    fl = 2.0 * T * k * n
    t = timeit(lambda: torch.addmm(b, x, w.t()));            rows.append((name + " fwd addmm", t, fl / t / 1e6))
    t = timeit(lambda: torch.mm(x, w.t()));                  rows.append((name + " fwd mm", t, fl / t / 1e6))
    t = timeit(lambda: torch.mm(dy, w));                     rows.append((name + " dgrad", t, fl / t / 1e6))
    t = timeit(lambda: torch.mm(dy.t(), x));                 rows.append((name + " wgrad bf16out", t, fl / t / 1e6))
    o32 = torch.empty(n, k, device=dev, dtype=torch.float32)
    try:
        t = timeit(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32)); rows.append((name + " wgrad f32out", t, fl / t / 1e6))
    except Exception as ex:
        rows.append((name + " wgrad f32out: " + str(ex)[:60], 0, 0))
for r in rows:
    print(f"{r[0]:28s} {r[1]:8.1f} us {r[2]:8.1f} TF/s")
