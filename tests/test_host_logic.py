"""CPU tests of host-side logic that needs no GPU: NT-Xent restatement vs the golden value produced by
main.nt_xent_loss, the .npz reader's layout / split, optimizer grouping helper."""
import numpy as np
import pytest
import torch

from helpers import load_golden, t


def test_nt_xent_matches_reference_golden():
    from gmlm_amd.harness import nt_xent_loss
    g = load_golden("g5_funcs")
    z1, z2 = t(g["ntx_z1"]), t(g["ntx_z2"])                     # 21 rows: two full chunks of 8 + a ragged chunk of 5
    assert abs(float(nt_xent_loss(z1, z2, 0.5, 8)) - float(g["ntx_loss"])) < 1e-5
    assert float(nt_xent_loss(z1[:0], z2[:0])) == 0.0
    assert float(nt_xent_loss(z1[:1], z2[:1])) == 0.0            # single-row chunk is skipped (main.py:117)
    z1.requires_grad_(True)
    nt_xent_loss(z1, z2, 0.5, 8).backward()
    assert torch.isfinite(z1.grad).all()


def test_npz_reader_layout_and_split(tmp_path):
    from gmlm_amd.data import load_npz_dataset
    n = 50
    rng = np.random.RandomState(0)
    path = tmp_path / "toy.npz"
    np.savez(path, node_features=rng.randn(n, 7).astype(np.float32), edges=rng.randint(0, n, (2, 120)),
             node_labels=rng.randint(0, 3, n), node_texts=np.array([f"text {i}" for i in range(n)]),
             label_texts=np.array(["a", "b", "c"]), train_masks=np.zeros(n, bool), val_masks=np.zeros(n, bool),
             test_masks=np.ones(n, bool))
    data, nf, nc = load_npz_dataset(str(path), split_ratios=(0.48, 0.32, 0.20), seed=42)
    assert (nf, nc, data.num_nodes) == (7, 3, n) and data.node_texts[3] == "text 3"
    idx = np.arange(n)
    np.random.RandomState(42).shuffle(idx)                       # main.py:793-795
    assert np.array_equal(np.sort(data.train_mask.nonzero().flatten().numpy()), np.sort(idx[:24]))
    assert int(data.train_mask.sum() + data.val_mask.sum() + data.test_mask.sum()) == n
    assert not bool((data.train_mask & data.val_mask).any())
    data2, _, _ = load_npz_dataset(str(path))
    assert bool(data2.test_mask.all()) and not bool(data2.train_mask.any())
    # object-dtype text arrays (what main.py:782's allow_pickle=True implies real files hold): refused by default, read
    # when the caller opts in for a file it wrote itself (this one)
    obj = tmp_path / "toy_obj.npz"
    np.savez(obj, node_features=rng.randn(n, 7).astype(np.float32), edges=rng.randint(0, n, (2, 120)),
             node_labels=rng.randint(0, 3, n), node_texts=np.array([f"text {i}" for i in range(n)], dtype=object),
             label_texts=np.array(["a", "b", "c"], dtype=object), train_masks=np.zeros(n, bool), val_masks=np.zeros(n, bool),
             test_masks=np.ones(n, bool))
    with pytest.raises(ValueError, match="allow_pickle"):
        load_npz_dataset(str(obj))
    data3, _, _ = load_npz_dataset(str(obj), allow_pickle=True)
    assert data3.node_texts == data2.node_texts and data3.label_texts == ["a", "b", "c"]


def test_tokenize_cache_is_constant_time_per_step_and_notices_changes():
    """GraphTextLM.tokenize (host side of main.py:342-345): the same text list costs no tokeniser call and no O(N) pass on
    later steps; a list that grew, shrank or was rewritten is tokenised again."""
    from transformers import BertConfig, BertModel
    import gmlm_amd

    calls = []

    class Tok:
        def __call__(self, texts, **kw):
            calls.append(len(texts))
            l = max(len(s.split()) for s in texts)
            ids = torch.zeros(len(texts), l, dtype=torch.long)
            am = torch.zeros_like(ids)
            for i, s in enumerate(texts):
                k = len(s.split())
                ids[i, :k] = torch.arange(1, k + 1)
                am[i, :k] = 1
            return {"input_ids": ids, "attention_mask": am}

    enc = BertModel(BertConfig(vocab_size=50, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
                               intermediate_size=64, max_position_embeddings=32))
    m = gmlm_amd.GraphTextLM(8, 4, 3, plm_encoder=enc, plm_tokenizer=Tok())
    texts = [f"node {i} has text" for i in range(1000)]
    t1 = m.tokenize(texts)
    assert calls == [1000] and tuple(t1.lens_host[:3].tolist()) == (4, 4, 4)
    assert m.tokenize(texts) is t1 and m.tokenize(texts) is t1 and calls == [1000]
    assert m.tokenize(t1) is t1                                           # pre-tokenised input passes through
    texts.append("one more")
    t2 = m.tokenize(texts)
    assert t2 is not t1 and calls == [1000, 1001]
    texts[:] = [s + " changed" for s in texts]                            # rewritten in place, same length
    t3 = m.tokenize(texts)
    assert t3 is not t2 and calls[-1] == 1001 and int(t3.lens_host[0]) == 5
    m.clear_caches()
    assert m.tokenize(texts) is not t3


def test_bucketed_layout_invariants():
    """Static sizes of a captured text-encoder batch (gmlm_amd/model.py::bucketed_layout): multiples of the quanta, every
    dummy sequence within 1..cap tokens, work items cover all sequences within the kernels' limits."""
    from gmlm_amd import ops
    from gmlm_amd.model import bucketed_layout
    rng = np.random.default_rng(0)
    for n_real, cap, quanta in ((0, 12, None), (1, 128, None), (548, 128, None), (700, 40, (16, 256, 8)), (63, 16, (64, 2048, 32)),
                                (5, 3, (8, 64, 4))):
        lens = rng.integers(1, cap + 1, n_real).tolist()
        la, s_b, t_b, groups = bucketed_layout(lens, cap, True, quanta)
        sq, tq, gq = quanta or (64, 2048, 32)
        c = max(cap, 16)
        assert la[:n_real] == lens and len(la) == s_b and sum(la) == t_b
        assert s_b % sq == 0 and t_b % tq == 0 and s_b > n_real
        assert all(1 <= v <= c for v in la[n_real:])
        b = groups.tolist()
        assert b[0] == 0 and b[-1] == s_b and all(x < y for x, y in zip(b[:-1], b[1:]))
        assert (len(b) - 1) % gq == 0 or len(b) - 1 == s_b
        for lo, hi in zip(b[:-1], b[1:]):
            assert hi - lo <= ops.SHORT_GROUP_SEQS and sum(la[lo:hi]) <= ops.SHORT_GROUP_ROWS
        # same bucket for a slightly different active set
        la2, s2, t2, g2 = bucketed_layout(lens[:-1] if lens else lens, cap, True, quanta)
        assert (s2, t2) == (s_b, t_b) or n_real == 0 or abs(s2 - s_b) == sq or abs(t2 - t_b) == tq
    assert bucketed_layout([5, 7], 12, False)[3] is None


def test_param_shadow_layout_and_small_autograd_helpers():
    """nn.ParamShadow (compute-dtype operands of a region in one flat buffer), nn._SplitLast (K|V split whose backward is one
    concatenation): host-checkable pieces of the
    kernel-count work - same values as the plain torch expressions they replace."""
    from gmlm_amd import nn as gnn
    g = torch.Generator().manual_seed(3)
    w1, w2 = torch.randn(5, 7, generator=g), torch.randn(3, 7, generator=g)
    b1, root = torch.randn(5, generator=g), torch.randn(6, 4, generator=g)
    sh = gnn.ParamShadow([("kv", [w1, w2], 0), (id(b1), [b1], 0), (id(root), [root], 2)], torch.bfloat16, torch.device("cpu"))
    assert torch.equal(sh.get("kv"), torch.cat([w1, w2], 0).to(torch.bfloat16))
    assert torch.equal(sh.get(id(b1)), b1.to(torch.bfloat16))
    r = sh.get(id(root))
    assert r.shape == (8, 4) and torch.equal(r[:6], root.to(torch.bfloat16)) and float(r[6:].abs().max()) == 0.0   # pad rows stay zero
    assert sh.get("missing") is None
    for v in sh.views.values():
        assert v.is_contiguous() and v.data_ptr() % 256 == sh.get("kv").data_ptr() % 256                          # 256-byte aligned slots
    x = torch.randn(4, 10, generator=g, requires_grad=True)
    a, b = gnn._SplitLast.apply(x, 6)
    (a.sum() * 2 + (b * b).sum()).backward()
    ref = torch.cat([torch.full((4, 6), 2.0), 2 * x.detach()[:, 6:]], 1)
    assert torch.equal(x.grad, ref)
