"""Probe: K10 basis composition forward/backward, cache-cold (a 2 GB fill runs between launches), vs a clone."""
import torch, sys
sys.path.insert(0, ".")
from gmlm_amd.nn import _BasisCompose
dev = torch.device("cuda")
junk = torch.empty(512 * 1024 * 1024, device=dev)          # 2 GB
def cold(fn, n=5):
    ts = []
    for _ in range(n):
        junk.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]
for ra in (1,):
    for inc, outc in ((2089, 768), (768, 1536), (1536, 3072), (3072, 768), (3072, 776)):
        cols = inc * outc
        w = torch.randn(30, cols, device=dev, requires_grad=True)
        comp = torch.randn(ra, 30, device=dev, requires_grad=True)
        g = torch.randn(ra, cols, device=dev)
        mb = 30 * cols * 4 / 1e6
        f = cold(lambda: _BasisCompose.apply(comp, w, None))
        out = _BasisCompose.apply(comp, w, None)
        def bw():
            w.grad = None; comp.grad = None
            out.backward(g, retain_graph=True)
        b = cold(bw)
        c = cold(lambda: w.detach().clone())
        print(f"ra={ra} in={inc} out={outc} ({mb:.0f} MB) COLD: fwd {f:.0f} us ({mb/f:.2f} TB/s)  bwd {b:.0f} us ({2*mb/b:.2f} TB/s)  clone {c:.0f} us ({2*mb/c:.2f} TB/s)", flush=True)
