"""hipGraph capture of the GNN + head regions (gmlm_amd/graphs.py) and the device-side dropout seed it needs.

* seed plumbing: every dropout kernel takes (seed, seed_dev) and must behave exactly like the plain seed ``seed + *seed_dev``
  (forward AND backward) -- bit-exact against the same op called with that host seed;
* replay == eager, bit for bit, with dropout off (same kernels, same order, same inputs);
* with dropout on, consecutive replays draw different masks and stay finite."""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_model import build_model, dev  # noqa: F401

pytestmark = pytest.mark.gpu


def _with_counter(value, dev, fn):
    from gmlm_amd import ops
    ops.SEED_DEVICE = torch.full((1,), value, dtype=torch.int64, device=dev)
    try:
        return fn()
    finally:
        ops.SEED_DEVICE = None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_device_seed_equals_host_seed_sum(dev, dtype):
    from gmlm_amd import ops
    g = torch.Generator(device=dev).manual_seed(11)
    s0, c = 123456789, 1000003

    def run_all(seed):
        outs = []
        x = torch.randn(300, 256, device=dev, generator=g, dtype=dtype).requires_grad_(True)
        b = torch.randn(256, device=dev).requires_grad_(True)
        y = ops.BiasGelu.apply(x, b, 0.3, seed)
        y.backward(torch.ones_like(y))
        outs += [y.detach(), x.grad.clone(), b.grad.clone()]
        x2 = torch.randn(300, 256, device=dev, dtype=dtype).requires_grad_(True)
        res = torch.randn(300, 256, device=dev, dtype=dtype)
        gm, bt = torch.ones(256, device=dev, requires_grad=True), torch.zeros(256, device=dev, requires_grad=True)
        y2 = ops.BiasResLayerNorm.apply(x2, b.detach(), res, gm, bt, 1e-5, True, 0.3, seed)
        y2.backward(torch.ones_like(y2))
        outs += [y2.detach(), x2.grad.clone(), gm.grad.clone()]
        z = torch.randn(300, 256, device=dev, dtype=dtype).requires_grad_(True)
        w, bb, ms = (torch.ones(256, device=dev, requires_grad=True), torch.zeros(256, device=dev, requires_grad=True),
                     torch.ones(256, device=dev, requires_grad=True))
        y3 = ops.GraphNormAct.apply(z, w, bb, ms, 1e-5, True, 0.3, seed, dtype, None, 300)
        y3.backward(torch.ones_like(y3))
        outs += [y3.detach(), z.grad.clone(), w.grad.clone()]
        q, k, v = (torch.randn(2, 200, 4 * 64, device=dev, dtype=dtype).requires_grad_(True) for _ in range(3))
        o = ops.Attention.apply(q, k, v, None, 4, 0.125, 0.2, seed)
        o.backward(torch.ones_like(o))
        outs += [o.detach(), q.grad.clone(), k.grad.clone(), v.grad.clone()]
        return outs

    g.manual_seed(11); torch.manual_seed(5)
    ref = run_all(s0 + c)
    g.manual_seed(11); torch.manual_seed(5)
    got = _with_counter(c, dev, lambda: run_all(s0))
    g.manual_seed(11); torch.manual_seed(5)
    other = _with_counter(c + 1, dev, lambda: run_all(s0))
    for i, (a, b_) in enumerate(zip(ref, got)):
        assert torch.equal(a, b_), f"tensor {i}: (seed, counter) differs from the plain seed seed + counter"
    assert not torch.equal(ref[0], other[0]) and not torch.equal(ref[9], other[9])      # the counter moves the masks


def _cfg(drop):
    plm = dict(hidden=128, layers=2, heads=2, inter=256, max_pos=64, vocab=200)
    return dict(n=301, e=2500, f_in=40, hc=32, c=5, plm=plm, seed=91, max_len=12)


def _data(cfg, dev):
    import gmlm_oracle as O
    import gmlm_amd
    g = torch.Generator().manual_seed(9)
    n = cfg["n"]
    x = torch.randn(n, cfg["f_in"], generator=g).to(dev)
    ei = torch.randint(0, n, (2, cfg["e"]), generator=g).to(dev)
    y = torch.randint(0, cfg["c"], (n,), generator=g).to(dev)
    ids, am = O.synthetic_tokens(n, 12, 200, 5, 2)
    tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
    masks = [(torch.rand(n, generator=g) < 0.5).to(dev) for _ in range(3)]
    return x, ei, y, tokens, masks


def _step(m, x, ei, y, tokens, mask, plm_batch=64):
    m.zero_grad(set_to_none=True)
    logits = m(m.soft_mask_input(x, mask, 0.7), ei, tokens, mask, plm_batch_size=plm_batch)
    loss = F.cross_entropy(logits[mask], y[mask], label_smoothing=0.2)
    loss.backward()
    return logits.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def _same_gradients(g0, g1, branched=False):
    """Every gradient bit for bit, in every replay mode (linear recordings, concurrent replay, whole-step graph with branches)."""
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("encoder", [False, True, "concurrent", "whole"])
def test_replay_equals_eager_without_dropout(dev, cd, encoder):
    """GNN + head regions (and, with ``encoder``, the text encoder recorded per size bucket) against the SAME computation run
    eagerly.  "concurrent": the same linear recordings, GNN and encoder replayed on two streams at once - still bit for bit.
    The recorded encoder works on the bucket-padded batch (``bucketed_layout``): the eager twin runs that very
    batch through ``plm_bucketed``; the next test ties the padded batch to the plain one."""
    cfg = _cfg(0.0)
    x, ei, y, tokens, masks = _data(cfg, dev)
    eager = build_model(cfg, dev, compute_dtype=cd).train()             # build_model: dropout_rate = 0
    eager.plm_bucketed = bool(encoder)
    graphed = build_model(cfg, dev, compute_dtype=cd).train()
    g = graphed.capture_hip_graphs(graphed.soft_mask_input(x, masks[0], 0.7), ei, encoder=bool(encoder),
                                   whole_step=encoder == "whole", concurrent=encoder == "concurrent")
    masks = masks + [masks[0]] + masks[::-1] + [masks[1]]                # ... and eight replays of buckets that are already recorded
    for mask in masks:                                                   # the active set (hence the packed PLM batch) changes per step
        pb = 512 if encoder else 64                                      # one packed batch (recordable) / several micro-batches (eager)
        l0, g0 = _step(eager, x, ei, y, tokens, mask, pb)
        l1, g1 = _step(graphed, x, ei, y, tokens, mask, pb)
        assert torch.equal(l0, l1)
        assert set(g0) == set(g1)
        _same_gradients(g0, g1, branched=encoder == "whole")
    assert (len(g._encoders) >= 1) == (encoder in (True, "concurrent")) and (len(g._steps) >= 1) == (encoder == "whole")
    graphed.eval()                                                       # evaluation falls back to the eager path
    eager.eval()
    with torch.no_grad():
        a = graphed(graphed.soft_mask_input(x, masks[0], 0.7), ei, tokens, masks[0], plm_batch_size=64)
        b = eager(eager.soft_mask_input(x, masks[0], 0.7), ei, tokens, masks[0], plm_batch_size=64)
    assert torch.equal(a, b)


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
def test_bucket_padding_changes_nothing_but_summation_order(dev, cd):
    """Dummy [PAD] sequences reach no output and add exact zeros to every weight gradient: the padded batch differs from the
    plain one only through the order of the split-K / column sums (and, in bf16, through which rows share an attention work
    item)."""
    cfg = _cfg(0.0)
    x, ei, y, tokens, masks = _data(cfg, dev)
    plain = build_model(cfg, dev, compute_dtype=cd).train()
    padded = build_model(cfg, dev, compute_dtype=cd).train()
    padded.plm_bucketed = True
    l0, g0 = _step(plain, x, ei, y, tokens, masks[1], 512)
    l1, g1 = _step(padded, x, ei, y, tokens, masks[1], 512)
    tol = 2e-5 if cd == torch.float32 else 2e-2
    assert (l0 - l1).abs().max() <= tol * max(1.0, float(l0.abs().max()))
    assert set(g0) == set(g1)
    gmax = max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        if k.endswith("key.bias"):                                       # mathematically zero gradient: rounding noise on both sides
            assert float(g0[k].abs().max()) <= 1e-4 * gmax and float(g1[k].abs().max()) <= 1e-4 * gmax, k
            continue
        scale = max(float(g0[k].abs().max()), 1e-4 * gmax)
        assert float((g0[k] - g1[k]).abs().max()) <= tol * scale, k


def test_bucket_lru_and_layout_change(dev):
    """More buckets than the LRU holds, then the first one again: recordings are dropped and re-made, results stay right."""
    import gmlm_amd.model as gm
    cfg = _cfg(0.0)
    x, ei, y, tokens, _ = _data(cfg, dev)
    gen = torch.Generator().manual_seed(3)
    n = cfg["n"]
    fracs = (0.05, 0.3, 0.55, 0.8, 0.95, 0.05)
    masks = [(torch.rand(n, generator=gen) < f).to(dev) for f in fracs[:-1]]
    masks.append(masks[0])
    eager = build_model(cfg, dev, compute_dtype=torch.float32).train()
    eager.plm_bucketed = True
    graphed = build_model(cfg, dev, compute_dtype=torch.float32).train()
    old = gm.ENCODER_BUCKET
    gm.ENCODER_BUCKET = (16, 256, 8)                                     # small quanta: every mask above lands in its own bucket
    try:
        g = graphed.capture_hip_graphs(graphed.soft_mask_input(x, masks[0], 0.7), ei, whole_step=False)
        g.encoder_buckets = 2
        for mask in masks:
            l0, g0 = _step(eager, x, ei, y, tokens, mask, 512)
            l1, g1 = _step(graphed, x, ei, y, tokens, mask, 512)
            assert torch.equal(l0, l1)
            for k in g0:
                assert torch.equal(g0[k], g1[k]), k
            assert len(g._encoders) <= 2
    finally:
        gm.ENCODER_BUCKET = old


def test_replays_draw_fresh_dropout_masks(dev):
    import gmlm_amd
    from test_gpu_model import hf_bert
    from helpers import model_state_template
    from param_recipe import recipe_state_dict
    cfg = _cfg(0.3)
    x, ei, y, tokens, masks = _data(cfg, dev)
    m = gmlm_amd.GraphTextLM(cfg["f_in"], cfg["hc"], cfg["c"], dropout_rate=0.3, plm_encoder=hf_bert(cfg["plm"]),
                             plm_max_length=12, compute_dtype=torch.bfloat16)
    m.load_state_dict(recipe_state_dict(model_state_template(cfg["f_in"], cfg["hc"], cfg["c"], cfg["plm"]), cfg["seed"]))
    m = m.to(dev).train()
    xm = m.soft_mask_input(x, masks[0], 0.7)
    g = m.capture_hip_graphs(xm, ei, whole_step=True)
    c0 = int(g.counter.item())
    e1 = g.gnn(xm).detach().clone()
    e2 = g.gnn(xm).detach().clone()
    assert int(g.counter.item()) == c0 + 2                              # bumped inside the graph, once per forward replay
    assert torch.isfinite(e1).all() and torch.isfinite(e2).all() and not torch.equal(e1, e2)
    losses = []
    for mask in masks:
        logits, grads = _step(m, x, ei, y, tokens, mask)
        assert torch.isfinite(logits).all() and all(torch.isfinite(v).all() for v in grads.values())
        losses.append(float(F.cross_entropy(logits[mask], y[mask])))
    assert len(set(losses)) == len(losses)
    # the whole-step recording (one micro-batch): the SAME mask twice draws two different dropout masks, a third call with
    # another active set records / replays another bucket or the same one with other tables; gradients stay finite and the
    # active index follows the mask
    outs = []
    for mask in (masks[0], masks[0], masks[1]):
        logits, grads = _step(m, x, ei, y, tokens, mask, 512)
        assert torch.isfinite(logits).all() and all(torch.isfinite(v).all() for v in grads.values())
        assert torch.equal(m.active_index, mask.nonzero(as_tuple=True)[0])
        outs.append(logits)
    assert len(g._steps) >= 1
    assert not torch.equal(outs[0], outs[1])


def test_mask_tensor_cache_notices_in_place_writes(dev):
    """The host copy of the active set is reused only for the very same, unwritten mask tensor (identity + version counter)."""
    cfg = _cfg(0.0)
    x, ei, y, tokens, masks = _data(cfg, dev)
    m = build_model(cfg, dev, compute_dtype=torch.float32).train()
    ref = build_model(cfg, dev, compute_dtype=torch.float32).train()
    mask = masks[0].clone()
    l0, _ = _step(m, x, ei, y, tokens, mask, 512)
    l1, _ = _step(m, x, ei, y, tokens, mask, 512)                       # cached active set
    assert torch.equal(l0, l1)
    mask.copy_(masks[1])                                                 # in-place write: version counter moves
    l2, _ = _step(m, x, ei, y, tokens, mask, 512)
    r2, _ = _step(ref, x, ei, y, tokens, masks[1], 512)
    assert torch.equal(l2, r2) and torch.equal(m.active_index, masks[1].nonzero(as_tuple=True)[0])
    l3, _ = _step(m, x, ei, y, tokens, masks[2], 512)                    # another tensor
    r3, _ = _step(ref, x, ei, y, tokens, masks[2], 512)
    assert torch.equal(l3, r3)


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
def test_stream_overlap_changes_nothing(dev, cd):
    """``model.overlap_streams``: the text encoder on a second HIP stream beside the GNN (forward and, through autograd's
    per-node streams, backward).  Same kernels, same inputs, same dropout seeds: logits and every gradient bit for bit,
    with dropout on, over changing masks."""
    import gmlm_amd
    from test_gpu_model import hf_bert
    from helpers import model_state_template
    from param_recipe import recipe_state_dict
    cfg = _cfg(0.3)
    x, ei, y, tokens, masks = _data(cfg, dev)

    def make():
        m = gmlm_amd.GraphTextLM(cfg["f_in"], cfg["hc"], cfg["c"], dropout_rate=0.3, plm_encoder=hf_bert(cfg["plm"]),
                                 plm_max_length=12, compute_dtype=cd)
        m.load_state_dict(recipe_state_dict(model_state_template(cfg["f_in"], cfg["hc"], cfg["c"], cfg["plm"]), cfg["seed"]))
        return m.to(dev).train()

    plain, over = make(), make()
    over.overlap_streams = True
    for mask in masks + masks:
        torch.manual_seed(1234)                                          # dropout seeds come from torch's CPU generator
        l0, g0 = _step(plain, x, ei, y, tokens, mask, 512)
        torch.manual_seed(1234)
        l1, g1 = _step(over, x, ei, y, tokens, mask, 512)
        assert torch.equal(l0, l1)
        assert set(g0) == set(g1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize("mode", ["concurrent", "whole"])
def test_recorded_modes_fall_back_when_the_text_batch_needs_micro_batches(dev, mode):
    """A step whose active texts do not fit one micro-batch cannot replay a recorded encoder: the whole-step mode then replays
    the GNN and head recordings around an eager encoder, the concurrent mode runs that eager encoder on the second stream beside
    the GNN recording.  Same logits and gradients as the eager model, bit for bit; a later step that fits replays again."""
    cfg = _cfg(0.0)
    x, ei, y, tokens, masks = _data(cfg, dev)
    eager = build_model(cfg, dev, compute_dtype=torch.bfloat16).train()
    graphed = build_model(cfg, dev, compute_dtype=torch.bfloat16).train()
    g = graphed.capture_hip_graphs(graphed.soft_mask_input(x, masks[0], 0.7), ei, whole_step=mode == "whole",
                                   concurrent=mode == "concurrent")
    for mask, pb in ((masks[0], 64), (masks[1], 64), (masks[0], 512), (masks[2], 64)):
        eager.plm_bucketed = pb == 512                                   # the recorded encoder works on the bucket-padded batch
        l0, g0 = _step(eager, x, ei, y, tokens, mask, pb)
        l1, g1 = _step(graphed, x, ei, y, tokens, mask, pb)
        assert torch.equal(l0, l1)
        assert set(g0) == set(g1)
        _same_gradients(g0, g1)
    assert len(g._encoders) + len(g._steps) >= 1
