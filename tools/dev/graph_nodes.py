"""How many hipMemsetAsync calls does recording one whole step issue?  Run under ``rocprofv3 --hip-trace --stats`` once with
NO_CAPTURE=1 (warm-up only) and once without: the difference in the hipMemsetAsync count is what the capture added, i.e. the
number of memset nodes in the recorded graph (a memset node misbehaves on replay: memset_graph_probe.py).
usage: [BF16=1] [NO_CAPTURE=1] python tools/dev/graph_nodes.py"""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
for p in ("../../tests", "../..", "../../oracle"):
    sys.path.insert(0, os.path.join(here, p))
import torch
import test_gpu_graphs as T
from test_gpu_model import build_model
from gmlm_amd import ops

dev = torch.device("cuda:0")
cd = torch.bfloat16 if os.environ.get("BF16") else torch.float32
cfg = T._cfg(0.0)
x, ei, y, tokens, masks = T._data(cfg, dev)
m = build_model(cfg, dev, compute_dtype=cd).train()
mask = masks[0]
seen = m._active_set(mask, None)
key, args, _ = m._bucket_tables(tokens, seen, cd)
xm = m.soft_mask_input(x, mask, 0.7).detach().requires_grad_(True)
params = [p for p in m.parameters() if p.requires_grad]


def fwd():
    plm = m.encode_packed_static(tokens, key, *args)
    gnn = m.get_graph_embeddings(xm, ei, None)
    return m.head(gnn, plm)


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        out = fwd()
        torch.autograd.grad(out.sum(), [p for p in params], allow_unused=True)
torch.cuda.current_stream().wait_stream(s)
if os.environ.get("NO_CAPTURE"):
    torch.cuda.synchronize()
    print("warm-up only")
    sys.exit(0)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd()
    grads = torch.autograd.grad(out.sum(), params, allow_unused=True)
torch.cuda.synchronize()
print("captured forward + backward of one whole step")
