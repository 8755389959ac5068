"""Development probe (GPU box): where do the bf16 HIP GNN block and the emulation part ways?  Layer 1 of the Squirrel workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
import gmlm_amd, gmlm_oracle as O, bf16_emulation as E
from gmlm_amd import ops
from helpers import oracle_model_from_config
from test_gpu_model import build_model
dev = torch.device("cuda:0")
n, e, f_in, c = O.WORKLOADS["squirrel"]
plm = dict(hidden=768, layers=1, heads=12, inter=128, max_pos=64, vocab=200)
cfg = dict(n=n, e=e, f_in=f_in, hc=768, c=c, plm=plm, seed=768)
data = O.synthetic_graph("squirrel")
mask = torch.zeros(n, dtype=torch.bool); mask[data["active_mask"].nonzero().reshape(-1)[:128]] = True
om, _ = oracle_model_from_config(cfg)
m = build_model(cfg, dev, compute_dtype=torch.bfloat16).eval()
x, ei = data["x"].to(dev), data["edge_index"].to(dev)
def cmp(name, a, b):
    a, b = a.detach().float().cpu(), b.detach().float()
    d = (a - b).abs()
    print(f"{name:28s} max {float(d.max()):.3e} mean {float(d.mean()):.3e} frac!=0 {float((d > 0).float().mean()):.4f} scale {float(b.abs().mean()):.3e}")
with torch.no_grad():
    x0 = m.soft_mask_input(x, mask.to(dev), 0.7)
    xm = O.soft_masking_gnn_input(data["x"], mask, om.gnn_mask_token_embed, 0.7)
    x0e = E.r(F.pad(xm, (0, (-f_in) % 8)))
    cmp("x0", x0, x0e)
    csr = m.graph(ei, n)
    et = O.edge_types_from_degree(data["edge_index"], n)
    rels = sorted(set(et.tolist()))
    conv, norm = m.rgcn1, m.gnorm1
    # HIP pieces
    w = conv.relation_weights(csr, x0.dtype, x0.shape[1] - conv.in_channels)
    root = F.pad(conv.root, (0, 0, 0, x0.shape[1] - conv.in_channels))
    out1 = torch.addmm(conv.bias.to(x0.dtype), x0, root.to(x0.dtype))
    h = ops.RGCNAggregate.apply(x0, csr)
    z = out1.clone().addmm_(h, w)
    y = m._block(1, x0, csr)
    # emulation pieces
    oc, on = om.rgcn1, om.gnorm1
    w_rel = (oc.comp[rels] @ oc.weight.view(oc.num_bases, -1)).view(len(rels), oc.in_channels, oc.out_channels)
    pad = x0e.shape[1] - oc.in_channels
    w_rel = F.pad(w_rel, (0, 0, 0, pad)); roote = F.pad(oc.root, (0, 0, 0, pad))
    cmp("W_cat (bf16)", w, E.r(w_rel.reshape(-1, oc.out_channels)))
    out1e = E.lin(x0e, roote.t(), oc.bias)
    cmp("out1 = addmm(bias,x,root)", out1, out1e)
    he = E._aggregate(x0e, data["edge_index"], et, rels)
    cmp("H aggregate", h, he)
    ze = E.r(out1e + x0e.new_zeros(1) + he @ E.r(w_rel.reshape(-1, oc.out_channels)))
    cmp("z = out1 + H W", z, ze)
    # z computed from the HIP operands by emulated arithmetic: isolates the GEMM
    zmix = E.r(out1.float().cpu() + h.float().cpu() @ w.float().cpu())
    cmp("z (HIP operands, CPU GEMM)", z, zmix)
    o1mix = E.r(x0.float().cpu() @ E.r(roote) + E.r(oc.bias))
    cmp("out1 (HIP x0, CPU GEMM)", out1, o1mix)
    ye = E._rgcn_block(oc, on, x0e, data["edge_index"], et, rels)
    cmp("y block 1", y, ye)
    ymix = E.r(F.gelu(O.graph_norm(z.float().cpu(), on.weight, on.bias, on.mean_scale, on.eps)))
    cmp("y (HIP z, CPU norm+gelu)", y, ymix)
    # residual
    x1 = (y.float() + gmlm_amd.nn._linear(x0[:, :f_in], m.residual_proj1.weight, m.residual_proj1.bias).float()).to(torch.bfloat16)
    x1e = E.r(ye + E.lin(x0e[:, :f_in], om.residual_proj1.weight, om.residual_proj1.bias))
    cmp("x1", x1, x1e)
    lin_h = gmlm_amd.nn._linear(x0[:, :f_in], m.residual_proj1.weight, m.residual_proj1.bias)
    cmp("residual_proj1 out", lin_h, E.lin(x0e[:, :f_in], om.residual_proj1.weight, om.residual_proj1.bias))
