"""GPU parity of the whole hot path (gmlm_amd.GraphTextLM forward + backward) against
 (a) the committed golden vectors produced by running the reference main.py, and
 (b) the CPU oracle on seeded synthetic graphs of BASELINE.json's sizes.
fp32 mode: logits within 1e-4 (north_star tolerance); bf16 mode: tolerance stated at the check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import gmlm_oracle as O
from helpers import bert_state_template, load_golden, model_state_template, oracle_model_from_config, t
from param_recipe import recipe_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def hf_bert(plm):
    from transformers import BertConfig, BertModel
    return BertModel(BertConfig(vocab_size=plm["vocab"], hidden_size=plm["hidden"], num_hidden_layers=plm["layers"],
                                num_attention_heads=plm["heads"], intermediate_size=plm["inter"],
                                max_position_embeddings=plm["max_pos"], hidden_dropout_prob=0.0,
                                attention_probs_dropout_prob=0.0))


def build_model(cfg, dev, **kw):
    import gmlm_amd
    m = gmlm_amd.GraphTextLM(cfg["f_in"], cfg["hc"], cfg["c"], dropout_rate=0.0, plm_encoder=hf_bert(cfg["plm"]),
                             plm_max_length=cfg.get("max_len", 16), **kw)
    sd = recipe_state_dict(model_state_template(cfg["f_in"], cfg["hc"], cfg["c"], cfg["plm"]), cfg["seed"])
    m.load_state_dict(sd, strict=True)          # reference state-dict keys load unchanged
    return m.to(dev)


@pytest.mark.parametrize("name", ["g1_toy", "g2_cornell"])
def test_golden_forward_backward(dev, name):
    import gmlm_amd
    g = load_golden(name)
    cfg = g["config"]
    m = build_model(cfg, dev).train()
    x, ei, mask = t(g["x"]).to(dev), t(g["edge_index"]).to(dev), t(g["node_mask"]).to(dev)
    tokens = gmlm_amd.TokenizedTexts.from_mask(t(g["input_ids"]).to(dev), t(g["attention_mask"]).to(dev))
    csr = m.graph(ei, cfg["n"])
    assert np.array_equal(csr.edge_type.cpu().numpy(), g["edge_type"])                    # bit-exact
    xm = gmlm_amd.soft_masking_gnn_input(x, mask, m.gnn_mask_token_embed, cfg["beta"])
    np.testing.assert_allclose(xm.detach().cpu().numpy(), g["x_soft_masked"], rtol=0, atol=1e-6)
    gnn = m.get_graph_embeddings(xm, ei)
    np.testing.assert_allclose(gnn.detach().cpu().numpy(), g["gnn_embeds"], rtol=1e-4, atol=1e-4)
    plm = m.encode_texts(tokens, mask, cfg["plm_batch_size"])
    np.testing.assert_allclose(plm.detach().cpu().numpy(), g["plm_embeds"], rtol=1e-4, atol=1e-4)
    logits = m(xm, ei, tokens, mask, plm_batch_size=cfg["plm_batch_size"])
    assert logits.dtype == torch.float32
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=0, atol=1e-4)   # north_star: 1e-4
    y = t(g["y"]).to(dev)
    loss = F.cross_entropy(logits[mask], y[mask], label_smoothing=0.2)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    for k, ref in g["grad_norms"].items():
        gr = grads[k]
        if ref < 0:
            assert gr is None or float(gr.abs().max()) == 0.0, k     # dead branch / unused pooler
            continue
        assert gr is not None, k
        nrm = float(gr.double().norm())
        assert abs(nrm - ref) <= 1e-3 * max(ref, 1e-3) + 1e-6, (k, nrm, ref)
        if "grad:" + k in g:
            np.testing.assert_allclose(gr.cpu().numpy(), g["grad:" + k], rtol=5e-3, atol=2e-5 * max(1.0, ref), err_msg=k)
    with torch.no_grad():
        np.testing.assert_allclose(m.get_graph_embeddings(x, ei).cpu().numpy(), g["gge_only"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["g4_bert_tiny", "g4_bert_base"])
def test_bert_block_vs_hf_golden(dev, name):
    from gmlm_amd import bert, ops
    g = load_golden(name)
    c = g["config"]
    plm = hf_bert(dict(vocab=c["vocab"], hidden=c["hidden"], layers=c["layers"], heads=c["heads"], inter=c["inter"],
                       max_pos=c["max_pos"]))
    plm.load_state_dict(recipe_state_dict(bert_state_template(c["hidden"], c["layers"], c["inter"], c["vocab"], c["max_pos"]),
                                          c["seed"]))
    plm = plm.to(dev).train()
    ids, am = t(g["input_ids"]).to(dev), t(g["attention_mask"]).to(dev)
    lens = am.sum(-1).to(torch.int32)
    hs = bert.bert_encode(plm, ids, lens, torch.float32, training=True)
    valid = am.bool().cpu()
    np.testing.assert_allclose(hs.detach().cpu()[valid].numpy(), t(g["last_hidden_state"])[valid].numpy(), rtol=1e-4, atol=1e-4)
    b, p = ids.shape[0], c["hidden"]
    pooled = ops.MeanPoolScatter.apply(torch.zeros(b, p, device=dev), hs, lens, torch.arange(b, device=dev))
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), g["pooled"], rtol=1e-4, atol=5e-5)
    (pooled * t(g["grad_pooled"]).to(dev)).sum().backward()
    grads = {k: v.grad for k, v in plm.named_parameters()}
    for k, ref in g["grad_norms"].items():
        nrm = float(grads[k].double().norm())
        assert abs(nrm - ref) <= 2e-3 * max(ref, 1e-3) + 1e-6, (k, nrm, ref)
    for k in ("embeddings.LayerNorm.weight", "encoder.layer.0.attention.self.query.bias"):
        np.testing.assert_allclose(grads[k].cpu().numpy(), g["grad:" + k], rtol=5e-3, atol=1e-4 * g["grad_norms"][k])


def test_hf_attention_interface_dropin(dev):
    """Unmodified HF BertModel with config._attn_implementation = 'gmlm_hip' == its own sdpa path."""
    from gmlm_amd import bert
    g = load_golden("g4_bert_tiny")
    c = g["config"]
    plm = hf_bert(dict(vocab=c["vocab"], hidden=c["hidden"], layers=c["layers"], heads=c["heads"], inter=c["inter"],
                       max_pos=c["max_pos"]))
    plm.load_state_dict(recipe_state_dict(bert_state_template(c["hidden"], c["layers"], c["inter"], c["vocab"], c["max_pos"]),
                                          c["seed"]))
    plm = plm.to(dev).eval()
    plm.config._attn_implementation = bert.register_hf_attention()
    ids, am = t(g["input_ids"]).long().to(dev), t(g["attention_mask"]).long().to(dev)
    with torch.no_grad():
        hs = plm(input_ids=ids, attention_mask=am).last_hidden_state
    valid = am.bool().cpu()
    np.testing.assert_allclose(hs.cpu()[valid].numpy(), t(g["last_hidden_state"])[valid].numpy(), rtol=1e-4, atol=1e-4)


def test_reference_modules_golden(dev):
    import gmlm_amd
    g = load_golden("g5_funcs")
    for tag, dim in (("small", 64), ("p768", 768)):
        ca = gmlm_amd.CrossAttention(dim, num_heads=8, dropout=0.0)
        ca.load_state_dict(recipe_state_dict(ca.state_dict(), 21))
        ca = ca.to(dev)
        x = t(g[f"ca_{tag}_x"]).to(dev).unsqueeze(0).requires_grad_(True)
        y = t(g[f"ca_{tag}_y"]).to(dev).unsqueeze(0).requires_grad_(True)
        o = ca(x, y)
        np.testing.assert_allclose(o.detach().cpu()[0].numpy(), g[f"ca_{tag}_out"], rtol=1e-4, atol=5e-5)
        (o * t(g[f"ca_{tag}_gout"]).to(dev).unsqueeze(0)).sum().backward()
        np.testing.assert_allclose(x.grad.cpu()[0].numpy(), g[f"ca_{tag}_gx"], rtol=2e-3, atol=5e-5)
        np.testing.assert_allclose(y.grad.cpu()[0].numpy(), g[f"ca_{tag}_gy"], rtol=2e-3, atol=5e-5)
        np.testing.assert_allclose(ca.q_proj.weight.grad.cpu().numpy(), g[f"ca_{tag}_gwq"], rtol=2e-3, atol=1e-4)
        np.testing.assert_allclose(ca.v_proj.bias.grad.cpu().numpy(), g[f"ca_{tag}_gbv"], rtol=2e-3, atol=1e-4)
    msf = gmlm_amd.MultiScaleFusion([8, 16, 32, 64], 48)
    msf.load_state_dict(recipe_state_dict(msf.state_dict(), 22))
    out = msf.to(dev)([t(g[f"msf_in{i}"]).to(dev) for i in range(4)])
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["msf_out"], rtol=1e-4, atol=2e-5)


def _oracle_vs_hip(dev, name, hc, plm, cd, atol, max_len=24, bs=32):
    import gmlm_amd
    n, e, f_in, c = O.WORKLOADS[name]
    cfg = dict(n=n, e=e, f_in=f_in, hc=hc, c=c, plm=plm, seed=500 + n % 97, max_len=max_len)
    data = O.synthetic_graph(name)
    ids, am = O.synthetic_tokens(n, max_len, plm["vocab"], seed=n, min_len=4)
    om, _ = oracle_model_from_config(cfg)
    mask = data["active_mask"]
    xm_ref = O.soft_masking_gnn_input(data["x"], mask, om.gnn_mask_token_embed, 0.7)
    ref = om(xm_ref, data["edge_index"], ids, am, mask, plm_batch_size=bs)
    loss_ref = F.cross_entropy(ref[mask], data["y"][mask], label_smoothing=0.2)
    loss_ref.backward()
    m = build_model(cfg, dev, compute_dtype=cd).train()
    x, ei, mk = data["x"].to(dev), data["edge_index"].to(dev), mask.to(dev)
    tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
    xm = m.soft_mask_input(x, mk, 0.7)
    logits = m(xm, ei, tokens, mk, plm_batch_size=bs)
    loss = F.cross_entropy(logits[mk], data["y"].to(dev)[mk], label_smoothing=0.2)
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=atol)
    # fp32: gradient norms to 2e-3.  bf16 (8-bit mantissa operands): every parameter's gradient must keep its SIZE to 6 %
    # and its DIRECTION (cosine >= 0.97 against the fp32 oracle gradient) - a norm-only check would pass a wrong gradient
    rel = 2e-3 if cd == torch.float32 else 0.06
    og = {k: p.grad for k, p in om.named_parameters()}
    for k, p in m.named_parameters():
        ok = k
        if k.startswith("plm_encoder."):
            ok = "plm_params." + k[len("plm_encoder."):].replace(".", "/")
        r = og.get(ok)
        if r is None or p.grad is None:
            assert (r is None or float(r.abs().max()) == 0) and (p.grad is None or float(p.grad.abs().max()) == 0), k
            continue
        rn, gn = float(r.double().norm()), float(p.grad.double().norm())
        # gradients that are analytically ~0 (e.g. a key bias under softmax) are pure rounding noise: floor the scale
        floor = 1e-4 if cd == torch.float32 else 2e-3
        assert abs(rn - gn) <= rel * max(rn, floor) + 1e-6, (k, gn, rn)
        if cd != torch.float32 and rn > 10 * floor:
            a, b = p.grad.detach().double().cpu().reshape(-1), r.double().reshape(-1)
            cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
            assert cos >= 0.97, (k, cos)
    if cd == torch.bfloat16:
        # the budget above is what 8-bit mantissas cost against fp32; the CHECK of the bf16 path is the oracle's bf16-emulating
        # mode (oracle/bf16_emulation.py: fp32 arithmetic, rounded where the kernels round): logits (bf16 numbers: grid 2^-8 ..
        # 2^-7 around 1) within two ulps and far closer on average than either side is to fp32, loss to 1e-3
        import bf16_emulation as E
        with torch.no_grad():
            le = E.forward(om, data["x"], data["edge_index"], ids, am, mask)
        de = (logits.detach().float().cpu() - le).abs()
        d32 = (logits.detach().float().cpu() - ref.detach()).abs()
        loss_e = F.cross_entropy(le[mask], data["y"][mask], label_smoothing=0.2)
        print(f"\n{name} bf16 HIP vs emulation: logits max {float(de.max()):.3e} mean {float(de.mean()):.3e} (vs fp32 oracle: max "
              f"{float(d32.max()):.3e} mean {float(d32.mean()):.3e}); dloss {abs(float(loss) - float(loss_e)):.2e}")
        assert float(de.max()) <= 1.6e-2 and float(de.mean()) <= 2.5e-3 and abs(float(loss) - float(loss_e)) <= 1e-3
        assert float(de.mean()) < 0.7 * float(d32.mean())
    return float(loss), float(loss_ref)


def test_chameleon_size_fp32_vs_oracle(dev):
    """BASELINE configs[1] geometry (N=2,277, E=36,101, hc=256, BERT-mini) in fp32: logits within 1e-4."""
    plm = dict(hidden=256, layers=4, heads=4, inter=1024, max_pos=64, vocab=200)
    l, lr = _oracle_vs_hip(dev, "chameleon", 256, plm, torch.float32, 1e-4)
    assert abs(l - lr) < 1e-4


def test_chameleon_size_bf16_vs_oracle(dev):
    """Same config with bf16 GEMM/attention operands (fp32 accumulation and statistics).  bf16 has an
    8-bit mantissa: logits O(1) agree to ~3e-2 after 4 GNN + 4 BERT layers; this is the bench dtype."""
    plm = dict(hidden=256, layers=4, heads=4, inter=1024, max_pos=64, vocab=200)
    l, lr = _oracle_vs_hip(dev, "chameleon", 256, plm, torch.bfloat16, 3e-2)            # measured 1.3e-2 against fp32; the check proper is the emulation inside
    assert abs(l - lr) < 3e-2


def test_eval_mode_and_empty_mask(dev):
    import gmlm_amd
    cfg = dict(f_in=32, hc=16, c=5, plm=dict(hidden=64, layers=1, heads=4, inter=128, max_pos=64, vocab=200), seed=3)
    m = build_model(cfg, dev).eval()
    n = 50
    x = torch.randn(n, 32, device=dev)
    ei = torch.randint(0, n, (2, 120), device=dev)
    ids, am = O.synthetic_tokens(n, 12, 200, 1, 2)
    tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
    with torch.no_grad():
        a = m(x, ei, tokens, torch.zeros(n, dtype=torch.bool, device=dev))        # no active node (main.py:333 guard)
        assert a.shape == (n, 5) and torch.isfinite(a).all()
        # explicit edge_type incl. the never-generated relation 4, single-node graph (GraphNorm skipped: main.py:273)
        et = torch.randint(0, 5, (120,), device=dev)
        b = m(x, ei, tokens, torch.ones(n, dtype=torch.bool, device=dev), edge_type=et, plm_batch_size=7)
        assert torch.isfinite(b).all()
        one = m.get_graph_embeddings(x[:1], torch.zeros(2, 0, dtype=torch.long, device=dev))
        assert one.shape == (1, 64) and torch.isfinite(one).all()


def test_large_power_law_graph_embeddings_vs_oracle(dev):
    """100k-node / 1M-edge power-law graph (hubs with thousands of edges -> chunked aggregation path, all four
    degree buckets occur): get_graph_embeddings forward + backward vs the CPU oracle, fp32, 1e-4."""
    import gmlm_amd
    n, e, f_in, hc = 100_000, 1_000_000, 64, 32
    g = torch.Generator().manual_seed(77)
    w = (torch.arange(n, dtype=torch.float32) + 1).pow(-1.0 / 1.2)
    perm = torch.randperm(n, generator=g)
    ei = torch.stack([perm[torch.multinomial(w, e, True, generator=g)], perm[torch.multinomial(w, e, True, generator=g)]])
    x = torch.randn(n, f_in, generator=g)
    plm = dict(hidden=64, layers=1, heads=4, inter=128, max_pos=64, vocab=200)
    cfg = dict(n=n, e=e, f_in=f_in, hc=hc, c=5, plm=plm, seed=9)
    om, _ = oracle_model_from_config(cfg)
    go = torch.randn(n, 64, generator=g)
    ref = om.get_graph_embeddings(x, ei)
    ref.backward(go)
    m = build_model(cfg, dev).train()
    csr = m.graph(ei.to(dev), n)
    assert csr.r_active == 4 and csr.split is not None and csr.t_split is not None
    assert np.array_equal(csr.edge_type.cpu().numpy(), O.edge_types_from_degree(ei, n).numpy())
    out = m.get_graph_embeddings(x.to(dev), ei.to(dev))
    out.backward(go.to(dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    og = {k: p.grad for k, p in om.named_parameters()}
    for k, p in m.named_parameters():
        r = og.get(k)
        if p.grad is None or r is None:
            continue
        rn, gn = float(r.double().norm()), float(p.grad.double().norm())
        assert abs(rn - gn) <= 2e-3 * max(rn, 1e-4) + 1e-6, (k, gn, rn)


def test_text_list_drop_in_path(dev):
    """forward(..., all_node_texts: list[str], ...) exactly as the reference calls it (main.py:545): host
    tokenisation with the HF tokenizer (once, cached), then the device pipeline.  Texts / vocabulary come
    from the golden fixture, whose logits were produced by main.GraphTextLM on the same strings."""
    from transformers import BertTokenizer
    import gmlm_amd
    g = load_golden("g1_toy")
    cfg = g["config"]
    tok = BertTokenizer(vocab={w: i for i, w in enumerate(g["vocab"].tolist())})
    m = gmlm_amd.GraphTextLM(cfg["f_in"], cfg["hc"], cfg["c"], dropout_rate=0.0, plm_encoder=hf_bert(cfg["plm"]),
                             plm_tokenizer=tok, plm_max_length=cfg["max_len"])
    m.load_state_dict(recipe_state_dict(model_state_template(cfg["f_in"], cfg["hc"], cfg["c"], cfg["plm"]), cfg["seed"]))
    m = m.to(dev).eval()
    texts = g["texts"].tolist()
    tt = m.tokenize(texts)
    assert np.array_equal(tt.lens.cpu().numpy(), g["attention_mask"].sum(-1))
    l = tt.input_ids.shape[1]
    assert np.array_equal(tt.input_ids.cpu().numpy(), g["input_ids"][:, :l])          # token ids: bit-exact
    with torch.no_grad():
        logits = m(t(g["x_soft_masked"]).to(dev), t(g["edge_index"]).to(dev), texts, t(g["node_mask"]).to(dev),
                   plm_batch_size=cfg["plm_batch_size"])
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=0, atol=1e-4)
    assert m.tokenize(texts) is tt                                                       # cached by list identity


def test_packed_bert_matches_hf_golden(dev):
    """Variable-length (packed) encoder pass vs HF BertModel outputs of the golden fixture (BERT-base geometry)."""
    from gmlm_amd import bert, ops
    g = load_golden("g4_bert_base")
    c = g["config"]
    plm = hf_bert(dict(vocab=c["vocab"], hidden=c["hidden"], layers=c["layers"], heads=c["heads"], inter=c["inter"],
                       max_pos=c["max_pos"]))
    plm.load_state_dict(recipe_state_dict(bert_state_template(c["hidden"], c["layers"], c["inter"], c["vocab"], c["max_pos"]),
                                          c["seed"]))
    plm = plm.to(dev).train()
    ids, am = t(g["input_ids"]).to(dev), t(g["attention_mask"]).to(dev).bool()
    lens = am.sum(-1)
    b, l = ids.shape
    cu = torch.zeros(b + 1, dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(lens, 0)
    pos = torch.arange(l, device=dev)[None].expand(b, l)[am]
    hs = bert.bert_encode_packed(plm, ids[am], pos, cu, int(lens.max()), torch.float32, training=True)
    np.testing.assert_allclose(hs.detach().cpu().numpy(), t(g["last_hidden_state"])[am.cpu()].numpy(), rtol=1e-4, atol=1e-4)
    pooled = ops.MeanPoolScatter.apply(torch.zeros(b, c["hidden"], device=dev), hs, None, torch.arange(b, device=dev), cu)
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), g["pooled"], rtol=1e-4, atol=5e-5)
    (pooled * t(g["grad_pooled"]).to(dev)).sum().backward()
    grads = {k: v.grad for k, v in plm.named_parameters()}
    for k, ref in g["grad_norms"].items():
        nrm = float(grads[k].double().norm())
        assert abs(nrm - ref) <= 2e-3 * max(ref, 1e-3) + 1e-6, (k, nrm, ref)


def test_activation_checkpointing_replays_dropout(dev):
    """Reference-style activation checkpointing (main.py:278-314) with dropout ON: the recompute must draw the
    same dropout masks as the first forward (seeds come from torch's CPU generator, whose state the checkpoint
    restores), so gradients equal those of the non-checkpointed run with the same seed."""
    plm = dict(hidden=128, layers=1, heads=2, inter=256, max_pos=64, vocab=200)
    n, e = 400, 3000
    cfg = dict(n=n, e=e, f_in=40, hc=32, c=4, plm=plm, seed=13)
    g = torch.Generator().manual_seed(4)
    x, ei = torch.randn(n, 40, generator=g).to(dev), torch.randint(0, n, (2, e), generator=g).to(dev)
    go = torch.randn(n, 128, generator=g).to(dev)
    grads = []
    for ckpt in (False, True):
        m = build_model(cfg, dev).train()
        for k in range(1, 5):
            getattr(m, f"dropout{k}").p = 0.3
        m.activation_checkpointing = ckpt
        torch.manual_seed(1234)
        out = m.get_graph_embeddings(x, ei)
        out.backward(go)
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 20
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * (1 + float(a.abs().max()))), k
    # and dropout really was active: a different seed gives different gradients
    m = build_model(cfg, dev).train()
    for k in range(1, 5):
        getattr(m, f"dropout{k}").p = 0.3
    torch.manual_seed(99)
    m.get_graph_embeddings(x, ei).backward(go)
    assert not torch.allclose(m.rgcn1.root.grad, grads[0]["rgcn1.root"], rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("t", [4096 + 5, 65536 + 37])
def test_splitk_wgrad_ragged_token_count(dev, t):
    """Weight gradient of the packed PLM path at a token count no slice count divides (slices + tail)."""
    from gmlm_amd.bert import _splitk_wgrad
    g = torch.Generator(device=dev).manual_seed(t)
    dy = torch.randn(t, 96, device=dev, generator=g) * 0.05
    x = torch.randn(t, 160, device=dev, generator=g)
    ref = dy.double().t() @ x.double()
    out32 = _splitk_wgrad(dy, x)
    assert out32.dtype == torch.float32 and torch.allclose(out32.double(), ref, rtol=1e-4, atol=1e-3)
    outb = _splitk_wgrad(dy.bfloat16(), x.bfloat16())
    refb = dy.bfloat16().double().t() @ x.bfloat16().double()
    assert outb.dtype == torch.bfloat16
    assert (outb.double() - refb).abs().max() <= 2 ** -7 * refb.abs().max()


@pytest.mark.parametrize("case", ["no_edges", "self_loops_and_duplicates", "one_active_one_token", "hub"])
def test_edge_case_graphs_vs_oracle(dev, case):
    """Degenerate inputs the reference accepts, logits within 1e-4 of the oracle in fp32: a graph without edges
    (every aggregation segment empty), only self loops + repeated edges (collisions inside a segment), a single
    active node whose text is one token (shortest packed sequence), one hub that every edge points to and from."""
    import gmlm_amd
    plm = dict(hidden=128, layers=2, heads=2, inter=256, max_pos=64, vocab=200)
    n = 40
    cfg = dict(n=n, e=0, f_in=24, hc=16, c=3, plm=plm, seed=77, max_len=16)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, 24, generator=g)
    y = torch.randint(0, 3, (n,), generator=g)
    mask = torch.rand(n, generator=g) < 0.5
    ids, am = O.synthetic_tokens(n, 16, 200, 5, 2)
    if case == "no_edges":
        ei = torch.zeros(2, 0, dtype=torch.long)
    elif case == "self_loops_and_duplicates":
        a = torch.arange(n)
        ei = torch.cat([torch.stack([a, a]), torch.tensor([[3, 3, 3, 3, 7, 7], [5, 5, 5, 5, 5, 5]])], 1)
    elif case == "hub":
        a = torch.arange(1, n)
        ei = torch.cat([torch.stack([a, torch.zeros_like(a)]), torch.stack([torch.zeros_like(a), a])], 1)
    else:
        ei = torch.randint(0, n, (2, 90), generator=g)
        mask = torch.zeros(n, dtype=torch.bool)
        mask[11] = True
        am[11] = 0
        am[11, 0] = 1
    om, _ = oracle_model_from_config(cfg)
    xm_ref = O.soft_masking_gnn_input(x, mask, om.gnn_mask_token_embed, 0.7)
    ref = om(xm_ref, ei, ids, am, mask, plm_batch_size=8)
    loss_ref = F.cross_entropy(ref[mask], y[mask], label_smoothing=0.2)
    loss_ref.backward()
    m = build_model(cfg, dev, compute_dtype=torch.float32).train()
    tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
    xm = m.soft_mask_input(x.to(dev), mask.to(dev), 0.7)
    logits = m(xm, ei.to(dev), tokens, mask.to(dev), plm_batch_size=8)
    loss = F.cross_entropy(logits[mask.to(dev)], y.to(dev)[mask.to(dev)], label_smoothing=0.2)
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=1e-4)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-4
    for k in ("rgcn1.weight", "rgcn4.root", "gnorm2.weight", "gnn_mask_token_embed", "classifier.3.weight"):
        r, p = dict(om.named_parameters())[k].grad, dict(m.named_parameters())[k].grad
        if r is None:
            assert p is None or float(p.abs().max()) == 0
            continue
        np.testing.assert_allclose(p.cpu().numpy(), r.numpy(), rtol=2e-3, atol=2e-5, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
def test_multi_scale_fusion_one_gemm_form_equals_streaming_form(dev, cd):
    """MultiScaleFusion (main.py:167-180): the one-GEMM form over the concatenated scales (nn._ScaledProjectionCat) against the
    streaming form (nn._ScaledProjectionSum, what the 10M-node graph uses) and, in fp32, against the module's plain definition."""
    import gmlm_amd
    g = torch.Generator().manual_seed(5)
    dims, p, n = [32, 64, 128, 256], 96, 777
    ref = torch.nn.ModuleList([torch.nn.Linear(d, p) for d in dims]).to(dev)
    embs = [torch.randn(n, d, generator=g).to(dev, cd) for d in dims]
    go = torch.randn(n, p, generator=g).to(dev)

    def run(limit):
        m = gmlm_amd.MultiScaleFusion(dims, p).to(dev)
        m.compute_dtype = cd
        m.cat_bytes_limit = limit
        with torch.no_grad():
            m.scale_weights.copy_(torch.tensor([0.1, 0.4, -0.3, 0.2]))
            for a, b in zip(m.projections, ref):
                a.weight.copy_(b.weight); a.bias.copy_(b.bias)
        xs = [e.clone().requires_grad_(True) for e in embs]
        y = m(xs)
        y.backward(go)
        return y.detach(), [x.grad for x in xs], {k: v.grad for k, v in m.named_parameters()}

    y1, dx1, dp1 = run(1 << 30)
    y0, dx0, dp0 = run(0)
    tol = 2e-5 if cd == torch.float32 else 2e-2
    assert float((y1 - y0).abs().max()) <= tol * float(y0.abs().max())
    for a, b in zip(dx1, dx0):
        assert float((a.float() - b.float()).abs().max()) <= tol * float(b.float().abs().max())
    for k in dp0:
        assert float((dp1[k] - dp0[k]).abs().max()) <= tol * float(dp0[k].abs().max()) + 1e-7, k
    if cd == torch.float32:
        w = torch.softmax(torch.tensor([0.1, 0.4, -0.3, 0.2], device=dev), 0)
        plain = sum(w[i] * ref[i](embs[i]) for i in range(4))
        m = gmlm_amd.MultiScaleFusion(dims, p).to(dev)
        plain = torch.nn.functional.layer_norm(plain, (p,), m.layer_norm.weight, m.layer_norm.bias, m.layer_norm.eps)
        assert float((y1 - plain).abs().max()) <= 2e-5 * float(plain.abs().max())
