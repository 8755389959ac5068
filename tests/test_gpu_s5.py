"""BASELINE configs[4] at FULL size on one GPU: the 10M-node / 100M-edge Chung-Lu power-law graph (SURVEY.md section 8d "S5":
alpha = 2.2, sources and targets sampled proportionally to the weights, ids randomly permuted), built on the GPU exactly as
``bench.py --workload s5`` builds it.  The reference cannot run this size at all (per-edge Python loop, main.py:257-267;
dense N x N scores, main.py:159-160), so the checks are index data bit-exact against numpy, sampled rows against the oracle
arithmetic, and size-independent properties:

 (a) K1: degree histogram and ``edge_type`` bit-exact vs numpy over ALL 100M edges, CSR ``rowptr`` bit-exact vs
     ``np.bincount``, ``col`` of ~1,000 sampled (target, relation) segments - the longest hubs included - vs a numpy stable
     sort of those segments;
 (b) K2: aggregation at F = 768, bf16 and fp32: the same sampled segments vs the mean of their source rows (float64),
     mean-of-ones = 1 on every non-empty segment and 0 on every empty one (bit-exact), bit-determinism;
 (c) ``get_graph_embeddings`` forward + backward at hidden_channels = 96 (the width at which all 10M nodes fit one GPU):
     finite, bit-deterministic, peak memory < 260 GB.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, E, F_IN = 10_000_000, 100_000_000, 768


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def graph(dev):
    g = torch.Generator(device=dev).manual_seed(1005)                     # bench.py s5_strong
    w = (torch.arange(N, device=dev, dtype=torch.float32) + 1.0).pow(-1.0 / 1.2)
    perm = torch.randperm(N, device=dev, generator=g)
    cdf = torch.cumsum(w.double(), 0)
    cdf /= cdf[-1].clone()
    src = perm[torch.searchsorted(cdf, torch.rand(E, device=dev, generator=g, dtype=torch.float64)).clamp_(max=N - 1)]
    dst = perm[torch.searchsorted(cdf, torch.rand(E, device=dev, generator=g, dtype=torch.float64)).clamp_(max=N - 1)]
    ei = torch.stack([src, dst])
    del w, perm, cdf, src, dst
    torch.cuda.empty_cache()
    return ei


def test_s5_graph_build_and_aggregation(dev, graph):
    import gmlm_amd
    from gmlm_amd import ops
    ei = graph
    csr = gmlm_amd.build_rel_csr(ei, N, 5)
    assert csr.active_relations == [0, 1, 2, 3] and csr.num_edges == E
    src_h, dst_h = ei[0].cpu().numpy(), ei[1].cpu().numpy()
    # ---- (a) integer part, bit-exact over all edges
    deg = np.bincount(src_h, minlength=N)
    assert np.array_equal(gmlm_amd.degree(ei[0], N, torch.int64).cpu().numpy(), deg)
    d = deg[src_h]
    et = np.full(E, 3, dtype=np.int64)
    et[d <= 10] = 2
    et[d <= 5] = 1
    et[d <= 2] = 0
    assert np.array_equal(csr.edge_type.cpu().numpy(), et)                 # main.py:260-267 on 100M edges
    hist = np.bincount(et, minlength=4)
    assert hist.min() > E // 100, hist                                    # every degree bucket is populated
    key = dst_h * 4 + et                                                   # segment = target * R_a + relation slot
    seg_len = np.bincount(key, minlength=4 * N)
    rowptr = np.zeros(4 * N + 1, dtype=np.int64)
    np.cumsum(seg_len, out=rowptr[1:])
    assert np.array_equal(csr.rowptr.cpu().numpy(), rowptr.astype(np.int32))
    assert np.array_equal(csr.t_rowptr.cpu().numpy(), np.concatenate([[0], np.cumsum(deg)]).astype(np.int32))
    # sampled segments: the 24 longest (hubs: > 10^5 edges, chunked reduction path) + 1,000 random non-empty ones
    rng = np.random.default_rng(5)
    nonempty = np.flatnonzero(seg_len)
    sample = np.unique(np.concatenate([np.argsort(seg_len)[-24:], rng.choice(nonempty, 1000, replace=False)]))
    assert seg_len[sample].max() > 100_000 and csr.split is not None and csr.split.n_long > 0
    pick = np.flatnonzero(np.isin(key, sample))                            # edges of the sampled segments, original order
    order = np.argsort(key[pick], kind="stable")
    pick = pick[order]                                                     # (segment, original edge id) order = the stable sort
    bounds = np.concatenate([[0], np.cumsum(seg_len[sample])])
    col_h = csr.col.cpu().numpy()
    perm_h = csr.perm.cpu().numpy()
    for i, s in enumerate(sample):
        lo, hi = rowptr[s], rowptr[s + 1]
        mine = pick[bounds[i]:bounds[i + 1]]
        assert np.array_equal(perm_h[lo:hi], mine.astype(np.int32)), s
        assert np.array_equal(col_h[lo:hi], src_h[mine].astype(np.int32)), s
    del col_h, perm_h, key, d
    # ---- (b) aggregation on the full graph, sampled rows vs float64 means of the source rows
    gx = torch.Generator(device=dev).manual_seed(77)
    sample_t = torch.from_numpy(sample).to(dev)
    src_rows = torch.from_numpy(src_h[pick]).to(dev)
    seg_of = torch.repeat_interleave(torch.arange(sample.size, device=dev), torch.from_numpy(seg_len[sample]).to(dev))
    for dt, tol in ((torch.bfloat16, 8e-3), (torch.float32, 2e-5)):
        x = torch.empty(N, F_IN, device=dev, dtype=dt)
        for c0 in range(0, N, 1 << 21):
            x[c0:c0 + (1 << 21)] = torch.randn(min(1 << 21, N - c0), F_IN, device=dev, generator=gx).to(dt)
        h = ops.RGCNAggregate.apply(x, csr)                                # [N, 4 * 768]
        got = h.view(4 * N, F_IN)[sample_t].double()
        ref = torch.zeros(sample.size, F_IN, dtype=torch.float64, device=dev)
        for c0 in range(0, src_rows.numel(), 1 << 20):                     # chunks: the hub segments gather ~10^6 rows
            ref.index_add_(0, seg_of[c0:c0 + (1 << 20)], x[src_rows[c0:c0 + (1 << 20)]].double())
        ref /= torch.from_numpy(seg_len[sample]).to(dev).double()[:, None]
        err = float((got - ref).abs().max())
        # rows ~ N(0,1): a mean of L rows is ~ L^-1/2; bf16 stores it with 8 mantissa bits, fp32 sums ~10^5 terms
        assert err <= tol, (dt, err)
        del got, ref
        if dt == torch.bfloat16:                                           # (the fp32 output alone is 123 GB: one copy at a time)
            h2 = ops.RGCNAggregate.apply(x, csr)
            assert torch.equal(h, h2)                                      # fixed summation order: bit-deterministic
            del h2
            x.fill_(1.0)
            ones = ops.RGCNAggregate.apply(x, csr).view(4 * N, F_IN)
            nz = torch.from_numpy(seg_len > 0).to(dev)
            # mean of ones: exactly 1 wherever a segment has an edge (sum of L ones * (1/L) rounds to 1 in bf16), 0 elsewhere
            assert bool((ones[:, 0].float() == nz.float()).all()) and bool((ones[:, -1].float() == nz.float()).all())
            cols = torch.randint(0, F_IN, (4,), device=dev)
            assert bool((ones[:, cols].float() == nz.float()[:, None]).all())
            del ones
        del x, h
        torch.cuda.empty_cache()


def test_s5_get_graph_embeddings_hc96_fits_one_gpu(dev, graph):
    import gmlm_amd
    from transformers import BertConfig, BertModel
    ei = graph
    torch.cuda.empty_cache()
    enc = BertModel(BertConfig(vocab_size=64, hidden_size=768, num_hidden_layers=1, num_attention_heads=12,
                               intermediate_size=64, max_position_embeddings=16))     # not executed: fixes P = 768
    torch.manual_seed(0)
    m = gmlm_amd.GraphTextLM(F_IN, 96, 16, dropout_rate=0.3, plm_encoder=enc, compute_dtype=torch.bfloat16,
                             activation_checkpointing=True).to(dev).train()
    gx = torch.Generator(device=dev).manual_seed(77)
    x = torch.empty(N, F_IN, device=dev, dtype=torch.bfloat16)
    for c0 in range(0, N, 1 << 21):
        x[c0:c0 + (1 << 21)] = torch.randn(min(1 << 21, N - c0), F_IN, device=dev, generator=gx).to(torch.bfloat16)
    mask = torch.rand(N, device=dev, generator=gx) < 0.3
    torch.cuda.reset_peak_memory_stats()
    runs = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(123)                                              # same dropout seeds (drawn from the CPU generator)
        out = m.get_graph_embeddings(m.soft_mask_input(x, mask, 0.7), ei)
        assert out.shape == (N, 768) and out.dtype == torch.float32
        (out.float().square().sum() / N).backward()
        gsum = torch.stack([p.grad.double().abs().sum() for p in m.parameters() if p.grad is not None])
        assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(gsum).all()) and float(gsum.sum()) > 0
        runs.append((out[::9973].clone(), m.rgcn1.root.grad.clone(), m.gnn_mask_token_embed.grad.clone(), m.rgcn4.comp.grad.clone()))
        del out
    for a, b in zip(*runs):
        assert torch.equal(a, b)                                            # forward rows and gradients bit for bit
    peak = torch.cuda.max_memory_allocated() / 1e9
    print(f"\nS5 get_graph_embeddings hc=96: peak {peak:.1f} GB")
    assert peak < 260.0
