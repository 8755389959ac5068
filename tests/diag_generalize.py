"""Diagnostic (not a test; run by hand on the GPU box: python tests/diag_generalize.py): does training with dropout ON
generalise like training with dropout OFF (same init, same masks)?  Lives under tests/ because it uses the oracle's
synthetic-input helpers, which only test code may import."""
import sys, numpy as np, torch, torch.nn.functional as F
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, d))
import gmlm_oracle as O
from helpers import model_state_template
from param_recipe import recipe_state_dict
import gmlm_amd
from gmlm_amd import harness
from transformers import BertConfig, BertModel
dev = torch.device("cuda:0")
plm = dict(hidden=128, layers=2, heads=2, inter=256, max_pos=64, vocab=200)
n, e, f_in, c = 600, 4000, 64, 4
g = torch.Generator().manual_seed(1)
x = torch.randn(n, f_in, generator=g); proto = torch.randn(c, f_in, generator=g)
y = (x @ proto.t()).argmax(1)
ei = torch.randint(0, n, (2, e), generator=g)
ids, am = O.synthetic_tokens(n, 16, 200, 3, 4)
ids[:, 1] = 5 + y
train = torch.zeros(n, dtype=torch.bool); train[: n // 2] = True
masks = []
for s in range(40):
    idx = torch.randperm(n // 2, generator=g)[:180]
    mk = torch.zeros(n, dtype=torch.bool); mk[idx] = True; masks.append(mk)
lrs = dict(lr_graph=3e-3, lr_bert=3e-4, lr_other=3e-3, weight_decay=0.01)
sd = recipe_state_dict(model_state_template(f_in, 32, c, plm), 31)
for gd, pd, init in ((0.0, 0.0, "recipe"), (0.1, 0.0, "recipe"), (0.0, 0.1, "recipe"), (0.1, 0.1, "recipe"), (0.1, 0.1, "hf"), (0.0, 0.0, "hf")):
    torch.manual_seed(0)
    enc = BertModel(BertConfig(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                               max_position_embeddings=64, hidden_dropout_prob=pd, attention_probs_dropout_prob=pd))
    m = gmlm_amd.GraphTextLM(f_in, 32, c, dropout_rate=gd, plm_encoder=enc, plm_max_length=16)
    if init == "recipe":
        m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    opt = harness.setup_optimizer(m, **lrs); sched = harness.linear_warmup_schedule(opt, 3, 60)
    tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
    L = []
    for mk in masks:
        r = harness.train_step(m, opt, sched, x.to(dev), ei.to(dev), tokens, y.to(dev), mk.to(dev), plm_batch_size=4096, autocast=True)
        L.append(round(r.loss, 3))
    print(f"gnn_drop {gd} plm_drop {pd} init {init}: losses", L[:3], L[-3:], flush=True)
    for nm, msk in (("train", train), ("held", ~train)):
        print("    eval", nm, harness.eval_step(m, x.to(dev), ei.to(dev), tokens, y.to(dev), msk.to(dev), plm_batch_size=4096, autocast=True), flush=True)
