"""Oracle parity of the attention kernel INSTANTIATIONS whose roofline numbers bench.py / DESIGN.md quote:

* CrossAttention geometry (main.py:159-163) h = 8, d = 96, N = 20,804, bf16: 8 heads x 82 query blocks >= 512 workgroups of
  256 rows selects the 8-wave kernels attn_fwd_pipe_kernel<96, 8>, attn_bwd_dq_kernel<bf16, 96, 8> and
  attn_bwd_dkv_kernel<bf16, 96, 8, ., 64>; with and without probability dropout;
* BERT geometry masked MHA (hf:modeling_bert.py:188-201) B = 32, h = 12, L = 512, d = 64, kv_len ~ U[256, 512], bf16:
  attn_fwd_pipe_kernel<64, 4> and the 4-wave dQ / dK-dV kernels with key masks.

The checker is the reference's own arithmetic, softmax(q k^T scale [+ mask]) v with nn.Dropout on the probabilities, in
float64 over the FULL tensors (dense N x N scores per head: 3.5 GB in fp64 at N = 20,804, evaluated head by head on the GPU by
torch, forward and autograd backward), fed with the same bf16 inputs.  The dropout mask is the kernels' replayable hash,
restated in tests/helpers.py.  Tolerances are bf16's: the kernels round P (and dS) to 8 mantissa bits before the second
product and the outputs once."""
import pytest
import torch

from helpers import attn_dropout_scale

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _dense_reference(q, k, v, go, h, scale, kv_len=None, drop=None):
    """q, k, v, go: [b, l, h*d] (any float dtype) -> out, dq, dk, dv in float64, one (batch, head) at a time."""
    b, lq, hd = q.shape
    lk, d = k.shape[1], hd // h
    outs = [torch.zeros(b, t.shape[1], hd, dtype=torch.float64, device=q.device) for t in (q, q, k, v)]
    for bi in range(b):
        for hi in range(h):
            sl = slice(hi * d, (hi + 1) * d)
            qq, kk, vv = (t[bi, :, sl].double().detach().requires_grad_(True) for t in (q, k, v))
            s = (qq @ kk.t()) * scale
            if kv_len is not None:
                s = s.masked_fill(torch.arange(lk, device=q.device)[None, :] >= int(kv_len[bi]), float("-inf"))
            p = torch.softmax(s, -1)
            del s
            if drop is not None:
                p = p * drop(bi, hi)
            o = p @ vv
            o.backward(go[bi, :, sl].double())
            outs[0][bi, :, sl], outs[1][bi, :, sl], outs[2][bi, :, sl], outs[3][bi, :, sl] = o.detach(), qq.grad, kk.grad, vv.grad
            del p, o
    return outs


def _close(name, got, ref, max_rel, l2_rel):
    got = got.double()
    e = (got - ref)
    mx, l2 = float(e.abs().max()) / float(ref.abs().max()), float(e.norm()) / float(ref.norm())
    print(f"  {name}: max|err| / max|ref| = {mx:.2e}, l2 rel = {l2:.2e}")
    assert mx <= max_rel and l2 <= l2_rel, (name, mx, l2)


@pytest.mark.parametrize("p_drop", [0.0, 0.3])
def test_cross_attention_geometry_8wave_kernels_vs_dense_fp64(dev, p_drop):
    from gmlm_amd import ops
    h, d, n = 8, 96, 20804
    g = torch.Generator().manual_seed(96)
    q, k, v, go = (torch.randn(1, n, h * d, generator=g).to(dev, torch.bfloat16) for _ in range(4))
    k = (k.float() * 1.5).to(torch.bfloat16)                      # scores ~ N(0, 1.5): a softmax that is neither flat nor one-hot
    seed = 0x1234_5678_9ABC
    qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = ops.Attention.apply(qd, kd, vd, None, h, d ** -0.5, p_drop, seed)
    out.backward(go)
    drop = None
    if p_drop:
        drop = lambda bi, hi: attn_dropout_scale(seed, (bi * h + hi) * n, n, n, p_drop, dev)
    ro, rq, rk, rv = _dense_reference(q, k, v, go, h, d ** -0.5, None, drop)
    print(f"\ncross-attention h=8 d=96 N={n} bf16, dropout {p_drop}:")
    # the probabilities are rounded to bf16 (2^-9 relative, independent): an output element sums thousands of them
    _close("out", out.detach(), ro, 2e-2, 6e-3)
    _close("dq", qd.grad, rq, 2e-2, 8e-3)
    _close("dk", kd.grad, rk, 2e-2, 8e-3)
    _close("dv", vd.grad, rv, 2e-2, 6e-3)
    if p_drop:                                                    # the mask really was applied, at the stated rate
        keep = float((attn_dropout_scale(seed, 0, 2048, 2048, p_drop, dev) > 0).double().mean())
        assert abs(keep - (1 - round(p_drop * 256) / 256)) < 2e-3


@pytest.mark.parametrize("b,l", [(32, 512), (2, 2048)])
def test_masked_mha_d64_vs_dense_fp64(dev, b, l):
    """B = 32 / L = 512 is the shape of ``attention.masked_mha_fwd`` (HBM side of the ridge); L = 2,048 the geometry of
    ``attention.masked_mha_fwd_L2048`` (the MFMA-bound regime of the same kernels; the bench runs it at B = 16, here B = 2 keeps
    the dense fp64 reference small - the kernels treat (batch, head) pairs independently)."""
    from gmlm_amd import ops
    h, d = 12, 64
    g = torch.Generator().manual_seed(64)
    q, k, v, go = (torch.randn(b, l, h * d, generator=g).to(dev, torch.bfloat16) for _ in range(4))
    kv_len = torch.randint(l // 2, l + 1, (b,), generator=g).to(dev, torch.int32)
    kv_len[0], kv_len[1] = l, l // 2 + 1                          # a full row of tiles and a block that straddles the length
    qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = ops.Attention.apply(qd, kd, vd, kv_len, h, d ** -0.5, 0.0, 0)
    out.backward(go)
    ro, rq, rk, rv = _dense_reference(q, k, v, go, h, d ** -0.5, kv_len.cpu())
    print(f"\nmasked MHA B={b} h={h} L={l} d={d} bf16, kv_len in [{int(kv_len.min())}, {int(kv_len.max())}]:")
    _close("out", out.detach(), ro, 2e-2, 6e-3)
    _close("dq", qd.grad, rq, 2e-2, 8e-3)
    _close("dk", kd.grad, rk, 2e-2, 8e-3)
    _close("dv", vd.grad, rv, 2e-2, 6e-3)
    valid = torch.arange(l, device=dev)[None, :] < kv_len[:, None]
    assert float(kd.grad[~valid].abs().max()) == 0.0 and float(vd.grad[~valid].abs().max()) == 0.0   # masked keys get no gradient
