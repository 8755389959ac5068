"""Development probe (GPU box): bf16 HIP path vs the oracle's bf16-emulating mode on the Squirrel h768 workload, stage by
stage, and the per-tensor gradient cosine table (against the emulation and against the fp32 HIP path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
import gmlm_amd, gmlm_oracle as O, bf16_emulation as E
from helpers import oracle_model_from_config
from test_gpu_model import build_model

BERT_BASE = dict(hidden=768, layers=12, heads=12, inter=3072, max_pos=512, vocab=30522)
dev = torch.device("cuda:0")
n, e, f_in, c = O.WORKLOADS["squirrel"]
hc = int(os.environ.get("HC", 768))
cfg = dict(n=n, e=e, f_in=f_in, hc=hc, c=c, plm=BERT_BASE, seed=768, max_len=32)
data = O.synthetic_graph("squirrel")
ids, am = O.synthetic_tokens(n, 32, BERT_BASE["vocab"], seed=n, min_len=8)
act = data["active_mask"].nonzero().reshape(-1)[:128]
mask = torch.zeros(n, dtype=torch.bool); mask[act] = True
om, _ = oracle_model_from_config(cfg)

def run(cd):
    m = build_model(cfg, dev, compute_dtype=cd).train()
    x, ei, mk = data["x"].to(dev), data["edge_index"].to(dev), mask.to(dev)
    tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
    parts = {}
    xm = m.soft_mask_input(x, mk, 0.7)
    parts["gnn_embeds"] = m.get_graph_embeddings(xm, ei).detach().float().cpu()
    parts["plm_embeds"] = m.encode_texts(tokens, mk, 128).detach().float().cpu()
    m.zero_grad(set_to_none=True)
    logits = m(xm, ei, tokens, mk, plm_batch_size=128)
    loss = F.cross_entropy(logits[mk], data["y"].to(dev)[mk], label_smoothing=0.2)
    loss.backward()
    return m, logits.detach().float().cpu(), float(loss), parts

m32, l32, loss32, p32 = run(torch.float32)
g32 = {k: p.grad.detach().double().cpu() for k, p in m32.named_parameters() if p.grad is not None}
del m32; torch.cuda.empty_cache()
mbf, lbf, lossbf, pbf = run(torch.bfloat16)
t0 = time.time()
le, pe = E.forward(om, data["x"], data["edge_index"], ids, am, mask, return_parts=True)
loss_e = F.cross_entropy(le[mask], data["y"][mask], label_smoothing=0.2); loss_e.backward()
print("emulation fwd+bwd %.1fs" % (time.time() - t0))
for k in ("gnn_embeds", "plm_embeds"):
    a, b, f = pbf[k], pe[k].detach(), p32[k]
    print(f"{k}: |hip-emul| max {float((a-b).abs().max()):.3e} mean {float((a-b).abs().mean()):.3e} | |hip-fp32| max {float((a-f).abs().max()):.3e} mean {float((a-f).abs().mean()):.3e} | scale {float(f.abs().mean()):.3e}")
print(f"logits: |hip-emul| max {float((lbf-le.detach()).abs().max()):.3e} mean {float((lbf-le.detach()).abs().mean()):.3e} | |hip-fp32| max {float((lbf-l32).abs().max()):.3e} mean {float((lbf-l32).abs().mean()):.3e}")
ge = {k: p.grad for k, p in om.named_parameters()}
rows = []
for k, p in mbf.named_parameters():
    ok = "plm_params." + k[len("plm_encoder."):].replace(".", "/") if k.startswith("plm_encoder.") else k
    b_ = ge.get(ok)
    if p.grad is None or b_ is None: continue
    a, b_ = p.grad.detach().double().cpu().reshape(-1), b_.double().reshape(-1)
    a32 = g32[k].reshape(-1)
    cos = lambda u, v: float(torch.dot(u, v) / (u.norm() * v.norm()).clamp(min=1e-30))
    rows.append((k, float(b_.norm()), float(a.norm()), cos(a, b_), cos(a, a32), cos(b_, a32)))
for r_ in sorted(rows, key=lambda r_: r_[3])[:40]:
    print(f"  {r_[0]:62s} |emul| {r_[1]:.3e} |hip| {r_[2]:.3e} cos(hip,emul) {r_[3]:.5f} cos(hip,fp32) {r_[4]:.4f} cos(emul,fp32) {r_[5]:.4f}")
