#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference (build container only).  TEST INFRASTRUCTURE.

    python oracle/make_golden.py            # writes tests/golden/

What runs here is /root/reference/main.py itself (``main.GraphTextLM``, ``main.CrossAttention``,
``main.MultiScaleFusion``, ``main.soft_masking_gnn_input``, ``main.nt_xent_loss``) and HuggingFace's
``BertModel`` (transformers 5.15.0).  Two things are substituted because the image lacks them:

* ``torch_geometric`` -> ``oracle/pyg_standin`` (the oracle's restatement of RGCNConv / GraphNorm /
  degree from PyG's published semantics: "[PyG, parity unpinned]");
* the PLM checkpoint -> a random-weight local ``BertModel`` directory passed as ``model_name``
  (``main.py:213-214`` would otherwise fetch ``thenlper/gte-base`` from the network).

``tokenizer.batch_encode_plus`` (``main.py:342``) no longer exists in transformers 5.x; the harness
aliases it to ``__call__`` on the instance.  Weights come from ``oracle/param_recipe.py`` so fixtures
hold inputs + outputs only.  Nothing under /root/reference is copied; this script and the fixtures
are what travels to the GPU box.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "pyg_standin"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from param_recipe import recipe_state_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"w{i}" for i in range(195)]


def _import_reference():
    scratch = tempfile.mkdtemp(prefix="gmlm_ref_")
    os.chdir(scratch)  # main.py:36 creates training_<ts>.log in the cwd at import time
    import main  # noqa: WPS433  (the reference)
    return main, scratch


def make_plm_dir(path, hidden, layers, heads, inter, max_pos=64):
    from transformers import BertConfig, BertModel, BertTokenizer
    cfg = BertConfig(vocab_size=len(VOCAB), hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                     intermediate_size=inter, max_position_embeddings=max_pos, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0)
    BertModel(cfg).save_pretrained(path)
    BertTokenizer(vocab={w: i for i, w in enumerate(VOCAB)}).save_pretrained(path)
    return cfg


def rand_texts(n, max_words, g):
    texts = []
    for i in range(n):
        k = int(torch.randint(1, max_words + 1, (1,), generator=g))
        if i == 3:
            k = 1
        words = torch.randint(0, 195, (k,), generator=g).tolist()
        texts.append(" ".join(f"w{w}" for w in words))
    texts[5 % n] = ""  # only [CLS][SEP]
    return texts


def run_model_case(main, scratch, name, n, e, f_in, hc, c, plm, seed, store_full_grads, special_graph=False):
    hidden, layers, heads, inter = plm
    plm_dir = os.path.join(scratch, f"plm_{name}")
    make_plm_dir(plm_dir, hidden, layers, heads, inter)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, f_in, generator=g)
    edge_index = torch.randint(0, n, (2, e), generator=g)
    if special_graph:
        # isolated nodes (0,1 never appear), self loops, duplicate edges, out-degree exactly 2/5/10/11
        edge_index = edge_index.clamp(min=2)
        k = 0
        for node, d in ((2, 2), (3, 5), (4, 10), (6, 11)):
            edge_index[0][edge_index[0] == node] = 7
            edge_index[0, k:k + d] = node
            k += d
        edge_index[:, k] = torch.tensor([9, 9])           # self loop
        edge_index[:, k + 1] = edge_index[:, k + 2]        # duplicate edge
    y = torch.randint(0, c, (n,), generator=g)
    node_mask = torch.rand(n, generator=g) < 0.4
    node_mask[3] = True
    node_mask[5 % n] = True
    texts = rand_texts(n, 12, g)
    max_len = 16

    model = main.GraphTextLM(f_in, hc, c, dropout_rate=0.0, model_name=plm_dir, plm_max_length=max_len)
    model.plm_tokenizer.batch_encode_plus = model.plm_tokenizer.__call__   # API drift shim (main.py:342)
    sd = recipe_state_dict(model.state_dict(), seed)
    model.load_state_dict(sd)
    model.train()

    cap = {}
    hooks = []
    for k in range(1, 5):
        hooks.append(getattr(model, f"rgcn{k}").register_forward_hook(
            lambda m, i, o, k=k: cap.update({f"rgcn{k}_out": o.detach().clone()})))
        hooks.append(getattr(model, f"gnorm{k}").register_forward_hook(
            lambda m, i, o, k=k: cap.update({f"gnorm{k}_out": o.detach().clone()})))
    hooks.append(model.rgcn1.register_forward_hook(lambda m, i, o: cap.update(edge_type=i[2].detach().clone())))
    hooks.append(model.graph_to_text_attn.register_forward_hook(
        lambda m, i, o: cap.update(gnn_embeds=i[0][0].detach().clone(), plm_embeds=i[1][0].detach().clone(),
                                   gnn_attended=o[0].detach().clone())))
    hooks.append(model.text_to_graph_attn.register_forward_hook(
        lambda m, i, o: cap.update(text_attended=o[0].detach().clone())))

    beta = 0.7
    xm = main.soft_masking_gnn_input(x, node_mask, model.gnn_mask_token_embed, beta=beta)
    logits = model(xm, edge_index, texts, node_mask, edge_type=None, plm_batch_size=8)
    loss = torch.nn.CrossEntropyLoss(label_smoothing=0.2)(logits[node_mask], y[node_mask])
    loss.backward()
    for h in hooks:
        h.remove()

    enc = model.plm_tokenizer(texts, padding="max_length", truncation=True, max_length=max_len, return_tensors="pt")
    out = dict(
        config=np.array(json.dumps(dict(n=n, e=e, f_in=f_in, hc=hc, c=c, plm=dict(hidden=hidden, layers=layers, heads=heads,
                                        inter=inter, max_pos=64, vocab=len(VOCAB)), seed=seed, beta=beta, max_len=max_len,
                                        plm_batch_size=8))),
        x=x.numpy(), edge_index=edge_index.numpy(), y=y.numpy(), node_mask=node_mask.numpy(),
        input_ids=enc["input_ids"].numpy().astype(np.int32), attention_mask=enc["attention_mask"].numpy().astype(np.int8),
        x_soft_masked=xm.detach().numpy(), logits=logits.detach().numpy(), loss=np.float32(loss.item()),
    )
    for k, v in cap.items():
        out[k] = v.numpy()
    if store_full_grads:   # toy case: keep the raw texts + vocabulary so the text -> tokenizer -> model path is testable
        out["texts"] = np.array(texts, dtype=np.str_)
        out["vocab"] = np.array(VOCAB, dtype=np.str_)
    gn = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            gn[k] = -1.0
            continue
        gn[k] = float(p.grad.double().norm())
        if store_full_grads or k in ("gnn_mask_token_embed", "classifier.3.weight", "gnorm2.mean_scale",
                                     "rgcn1.comp", "multi_scale_fusion.scale_weights"):
            out["grad:" + k] = p.grad.numpy()
    out["grad_norms"] = np.array(json.dumps(gn))
    # get_graph_embeddings standalone (second caller: main.py:447-448)
    with torch.no_grad():
        out["gge_only"] = model.get_graph_embeddings(x, edge_index, None).numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, "loss", loss.item(), "logits", tuple(logits.shape), "keys", len(out))


def int_fixtures(main, scratch):
    """G3: edge types produced by the reference loop (main.py:253-267) on awkward graphs."""
    out = {}
    plm_dir = os.path.join(scratch, "plm_int")
    make_plm_dir(plm_dir, 32, 1, 2, 64)
    model = main.GraphTextLM(4, 4, 2, dropout_rate=0.0, model_name=plm_dir, plm_max_length=8)
    model.eval()
    g = torch.Generator().manual_seed(7)
    cases = []
    for n, e in ((1, 0), (1, 3), (2, 1), (16, 40), (50, 400), (97, 1500), (300, 900)):
        ei = torch.randint(0, n, (2, e), generator=g)
        cases.append((n, ei))
    # out-degree exactly 0,1,2,3,5,6,10,11,12 for nodes 0..8, targets random
    degs = [0, 1, 2, 3, 5, 6, 10, 11, 12]
    src = torch.cat([torch.full((d,), i) for i, d in enumerate(degs)])
    ei = torch.stack([src, torch.randint(0, 9, (src.numel(),), generator=g)])
    ei = ei[:, torch.randperm(ei.size(1), generator=g)]
    cases.append((9, ei))
    for ci, (n, ei) in enumerate(cases):
        cap = {}
        h = model.rgcn1.register_forward_hook(lambda m, i, o: cap.update(et=i[2].clone()))
        with torch.no_grad():
            model.get_graph_embeddings(torch.zeros(n, 4), ei, None)
        h.remove()
        out[f"c{ci}_n"] = np.int64(n)
        out[f"c{ci}_edge_index"] = ei.numpy()
        out[f"c{ci}_edge_type"] = cap["et"].numpy()
        out[f"c{ci}_degree"] = main.degree(ei[0], num_nodes=n).numpy()
    out["num_cases"] = np.int64(len(cases))
    # generate_active_node_mask index parts (main.py:49, 86-88): nonzero order + mask build
    m = torch.rand(200, generator=g) < 0.3
    out["mask"] = m.numpy()
    out["mask_nonzero"] = m.nonzero(as_tuple=False).reshape(-1).numpy()
    np.savez_compressed(os.path.join(OUT, "g3_int.npz"), **out)
    print("g3_int cases", len(cases))


def bert_fixture(name, hidden, layers, heads, inter, b, l, seed):
    """G4: HF BertModel in/out on a padded batch (mixed lengths incl. 1) + the reference's mean pool."""
    from transformers import BertConfig, BertModel
    cfg = BertConfig(vocab_size=len(VOCAB), hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                     intermediate_size=inter, max_position_embeddings=max(64, l), hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0)
    m = BertModel(cfg).eval()
    m.load_state_dict(recipe_state_dict(m.state_dict(), seed))
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, l + 1, (b,), generator=g)
    lens[0], lens[1] = l, 1
    ids = torch.randint(5, len(VOCAB), (b, l), generator=g)
    am = (torch.arange(l)[None] < lens[:, None]).long()
    ids = ids * am
    hs = m(input_ids=ids, attention_mask=am).last_hidden_state
    me = am.unsqueeze(-1).expand(hs.size()).float()           # main.py:353-356
    pooled = torch.sum(hs * me, 1) / torch.clamp(me.sum(1), min=1e-9)
    gout = torch.randn(b, hidden, generator=g)
    (pooled * gout).sum().backward()
    out = dict(config=np.array(json.dumps(dict(hidden=hidden, layers=layers, heads=heads, inter=inter,
                                                max_pos=max(64, l), vocab=len(VOCAB), seed=seed))),
               input_ids=ids.numpy().astype(np.int32), attention_mask=am.numpy().astype(np.int8),
               last_hidden_state=hs.detach().numpy(), pooled=pooled.detach().numpy(), grad_pooled=gout.numpy())
    gn = {k: float(p.grad.double().norm()) for k, p in m.named_parameters() if p.grad is not None}
    out["grad_norms"] = np.array(json.dumps(gn))
    out["grad:embeddings.LayerNorm.weight"] = m.embeddings.LayerNorm.weight.grad.numpy()
    out["grad:encoder.layer.0.attention.self.query.bias"] = m.encoder.layer[0].attention.self.query.bias.grad.numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, tuple(hs.shape))


def func_fixtures(main):
    """G5: reference-owned functions on their own (CrossAttention, MultiScaleFusion, soft mask, NT-Xent)."""
    out = {}
    g = torch.Generator().manual_seed(11)
    for tag, n, dim in (("small", 37, 64), ("p768", 150, 768)):
        ca = main.CrossAttention(dim, num_heads=8, dropout=0.0)
        ca.load_state_dict(recipe_state_dict(ca.state_dict(), 21))
        x = torch.randn(1, n, dim, generator=g, requires_grad=True)
        yv = torch.randn(1, n, dim, generator=g)
        yv[0, ::3] = 0          # inactive nodes: plm_embeds rows are zero (main.py:328)
        yv.requires_grad_(True)
        o = ca(x, yv)
        go = torch.randn(o.shape, generator=g)
        (o * go).sum().backward()
        out.update({f"ca_{tag}_x": x.detach()[0].numpy(), f"ca_{tag}_y": yv.detach()[0].numpy(), f"ca_{tag}_out": o.detach()[0].numpy(),
                    f"ca_{tag}_gout": go[0].numpy(), f"ca_{tag}_gx": x.grad[0].numpy(), f"ca_{tag}_gy": yv.grad[0].numpy(),
                    f"ca_{tag}_gwq": ca.q_proj.weight.grad.numpy(), f"ca_{tag}_gbv": ca.v_proj.bias.grad.numpy()})
    msf = main.MultiScaleFusion([8, 16, 32, 64], 48)
    msf.load_state_dict(recipe_state_dict(msf.state_dict(), 22))
    embs = [torch.randn(29, d, generator=g) for d in (8, 16, 32, 64)]
    out["msf_out"] = msf(embs).detach().numpy()
    for i, e_ in enumerate(embs):
        out[f"msf_in{i}"] = e_.numpy()
    x = torch.randn(40, 12, generator=g)
    m = torch.rand(40, generator=g) < 0.5
    tok = torch.randn(1, 12, generator=g)
    out.update(sm_x=x.numpy(), sm_mask=m.numpy(), sm_tok=tok.numpy(),
               sm_out=main.soft_masking_gnn_input(x, m, tok, beta=0.7).numpy(),
               sm_out_empty=main.soft_masking_gnn_input(x, torch.zeros(40, dtype=torch.bool), tok, beta=0.7).numpy())
    z1, z2 = torch.randn(21, 48, generator=g), torch.randn(21, 48, generator=g)
    out.update(ntx_z1=z1.numpy(), ntx_z2=z2.numpy(), ntx_loss=np.float32(main.nt_xent_loss(z1, z2, 0.5, 8).item()))
    np.savez_compressed(os.path.join(OUT, "g5_funcs.npz"), **out)
    print("g5_funcs", len(out))


def main_():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main, scratch = _import_reference()
    run_model_case(main, scratch, "g1_toy", n=64, e=256, f_in=32, hc=16, c=5, plm=(64, 2, 4, 128), seed=101,
                   store_full_grads=True, special_graph=True)
    run_model_case(main, scratch, "g2_cornell", n=183, e=298, f_in=1703, hc=64, c=5, plm=(128, 2, 4, 256), seed=102,
                   store_full_grads=False)
    int_fixtures(main, scratch)
    bert_fixture("g4_bert_tiny", 64, 2, 4, 128, b=6, l=24, seed=104)
    bert_fixture("g4_bert_base", 768, 12, 12, 3072, b=4, l=48, seed=105)
    func_fixtures(main)


if __name__ == "__main__":
    main_()
