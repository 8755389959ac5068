"""The two PyG operators through their PyG call signatures (reference call sites
main.py:272-273, 285-286, 298-299, 308-309: ``conv(x, edge_index, edge_type)``, ``norm(x)``).

Two independent anchors, because PyG itself is not importable here (the pair stays "parity unpinned" against the real
library):
 * a HAND-WORKED 4-node, 2-relation example (duplicate edge, a node with no in-edge at all, empty per-relation
   neighbourhoods, mean_scale != 1) whose expected outputs are literal numbers derived on paper from PyG's published
   definitions - not produced by any code of this repository;
 * an in-test numpy restatement written as explicit per-edge loops (float64), used for gradients (central finite
   differences) and for a 5-relation case that includes relation id 4 and an int32 ``edge_index``.
CPU tests pin the oracle to both; GPU tests pin gmlm_amd.RGCNConv / GraphNorm (the HIP path) to both and to the oracle.
"""
import numpy as np
import pytest
import torch

import gmlm_oracle as O

# ---- the hand-worked example ------------------------------------------------------------------------------------
HX = np.array([[1., 2.], [3., 4.], [5., 6.], [7., 8.]])
# (src -> dst, relation): 2 -> 1 appears twice (duplicate edges count twice in the mean); node 2 has no in-edge;
# node 0 has no relation-0 in-edge; nodes 2, 3 have no relation-1 in-edge
HEDGES = [(0, 1, 0), (2, 1, 0), (2, 1, 0), (3, 1, 1), (1, 0, 1), (0, 3, 0)]
HWEIGHT = np.array([[[1., 0.], [0., 1.]], [[0., 1.], [1., 0.]]])      # 2 bases
HCOMP = np.array([[1., 2.], [0.5, -1.]])                                # W_0 = [[1,2],[2,1]], W_1 = [[.5,-1],[-1,.5]]
HROOT = np.array([[1., 1.], [0., -1.]])
HBIAS = np.array([0.1, -0.2])
# on paper: mean_0 = [0; (x0+2*x2)/3 = (11/3, 14/3); 0; x0], mean_1 = [x1; x3; 0; 0];
# out_i = mean_0[i] W_0 + mean_1[i] W_1 + x_i root + bias
HOUT = np.array([[-1.4, -2.2], [11.6, 7.8], [5.1, -1.2], [12.1, 2.8]])

GX = np.array([[1., 2.], [3., 6.], [5., 10.], [7., 14.]])
GW, GB, GMS = np.array([2., 1.]), np.array([0.5, -1.]), np.array([0.5, 2.0])
# column 0: mean 4, o = x - 2 = (-1, 1, 3, 5), var = 9;  column 1: mean 8, o = x - 16 = (-14, -10, -6, -2), var = 84
GOUT = np.stack([2. * np.array([-1., 1., 3., 5.]) / np.sqrt(9. + 1e-5) + 0.5,
                 np.array([-14., -10., -6., -2.]) / np.sqrt(84. + 1e-5) - 1.], 1)


def hand_graph(dtype=torch.long):
    e = np.array(HEDGES)
    return torch.tensor(e[:, :2].T.copy(), dtype=dtype), torch.tensor(e[:, 2].copy(), dtype=torch.long)


# ---- numpy restatement as explicit loops (float64) ----------------------------------------------------------------
def np_rgcn(x, edges, weight, comp, root, bias):
    n, r = x.shape[0], comp.shape[0]
    out = x @ root + bias
    for rel in range(r):
        w = sum(comp[rel, b] * weight[b] for b in range(weight.shape[0]))
        for i in range(n):
            nb = [s for (s, d, t) in edges if d == i and t == rel]
            if nb:
                out[i] += (sum(x[s] for s in nb) / len(nb)) @ w
    return out


def np_graphnorm(x, w, b, ms, eps=1e-5):
    o = x - x.mean(0) * ms
    return w * o / np.sqrt((o * o).mean(0) + eps) + b


def fd_grad(f, a, g, h=1e-6):
    """d sum(f(a) * g) / d a by central differences (float64)."""
    out = np.zeros_like(a)
    it = np.nditer(a, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        ap, am = a.copy(), a.copy()
        ap[i] += h
        am[i] -= h
        out[i] = ((f(ap) - f(am)) * g).sum() / (2 * h)
    return out


def test_numpy_restatement_reproduces_the_paper_numbers():
    np.testing.assert_allclose(np_rgcn(HX, HEDGES, HWEIGHT, HCOMP, HROOT, HBIAS), HOUT, rtol=0, atol=1e-12)
    np.testing.assert_allclose(np_graphnorm(GX, GW, GB, GMS), GOUT, rtol=0, atol=1e-12)


def test_oracle_matches_handworked_example_and_finite_differences():
    ei, et = hand_graph()
    tt = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)   # noqa: E731
    x, w, c, r, b = tt(HX), tt(HWEIGHT), tt(HCOMP), tt(HROOT), tt(HBIAS)
    out = O.rgcn_conv(x, ei, et, w, c, r, b)
    np.testing.assert_allclose(out.detach().numpy(), HOUT, rtol=0, atol=1e-12)
    g = np.arange(8, dtype=np.float64).reshape(4, 2) / 4 - 0.7
    out.backward(torch.tensor(g))
    np.testing.assert_allclose(x.grad.numpy(), fd_grad(lambda a: np_rgcn(a, HEDGES, HWEIGHT, HCOMP, HROOT, HBIAS), HX, g), atol=1e-7)
    np.testing.assert_allclose(w.grad.numpy(), fd_grad(lambda a: np_rgcn(HX, HEDGES, a, HCOMP, HROOT, HBIAS), HWEIGHT, g), atol=1e-7)
    np.testing.assert_allclose(c.grad.numpy(), fd_grad(lambda a: np_rgcn(HX, HEDGES, HWEIGHT, a, HROOT, HBIAS), HCOMP, g), atol=1e-7)
    np.testing.assert_allclose(r.grad.numpy(), fd_grad(lambda a: np_rgcn(HX, HEDGES, HWEIGHT, HCOMP, a, HBIAS), HROOT, g), atol=1e-7)
    np.testing.assert_allclose(b.grad.numpy(), g.sum(0), atol=1e-12)
    gx, gw, gb, gms = tt(GX), tt(GW), tt(GB), tt(GMS)
    y = O.graph_norm(gx, gw, gb, gms)
    np.testing.assert_allclose(y.detach().numpy(), GOUT, rtol=0, atol=1e-12)
    y.backward(torch.tensor(g))
    np.testing.assert_allclose(gx.grad.numpy(), fd_grad(lambda a: np_graphnorm(a, GW, GB, GMS), GX, g), atol=1e-6)
    np.testing.assert_allclose(gms.grad.numpy(), fd_grad(lambda a: np_graphnorm(GX, GW, GB, a), GMS, g), atol=1e-6)
    np.testing.assert_allclose(gw.grad.numpy(), fd_grad(lambda a: np_graphnorm(GX, a, GB, GMS), GW, g), atol=1e-6)


# ---- GPU: the HIP modules through the PyG signatures ---------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("idx_dtype", [torch.long, torch.int32])
def test_hip_rgcnconv_handworked(dev, idx_dtype):
    import gmlm_amd
    ei, et = hand_graph(idx_dtype)
    conv = gmlm_amd.RGCNConv(2, 2, num_relations=2, num_bases=2).to(dev)
    with torch.no_grad():
        conv.weight.copy_(torch.tensor(HWEIGHT)); conv.comp.copy_(torch.tensor(HCOMP))
        conv.root.copy_(torch.tensor(HROOT)); conv.bias.copy_(torch.tensor(HBIAS))
    x = torch.tensor(HX, dtype=torch.float32, device=dev, requires_grad=True)
    out = conv(x, ei.to(dev), et.to(dev))                       # exactly main.py:272's call form
    assert out.dtype == torch.float32 and out.shape == (4, 2)
    np.testing.assert_allclose(out.detach().cpu().numpy(), HOUT, rtol=0, atol=2e-6)
    g = np.arange(8, dtype=np.float64).reshape(4, 2) / 4 - 0.7
    out.backward(torch.tensor(g, dtype=torch.float32, device=dev))
    chk = lambda t, ref: np.testing.assert_allclose(t.detach().cpu().numpy(), ref, rtol=0, atol=2e-5)   # noqa: E731
    chk(x.grad, fd_grad(lambda a: np_rgcn(a, HEDGES, HWEIGHT, HCOMP, HROOT, HBIAS), HX, g))
    chk(conv.weight.grad, fd_grad(lambda a: np_rgcn(HX, HEDGES, a, HCOMP, HROOT, HBIAS), HWEIGHT, g))
    chk(conv.comp.grad, fd_grad(lambda a: np_rgcn(HX, HEDGES, HWEIGHT, a, HROOT, HBIAS), HCOMP, g))
    chk(conv.root.grad, fd_grad(lambda a: np_rgcn(HX, HEDGES, HWEIGHT, HCOMP, a, HBIAS), HROOT, g))
    chk(conv.bias.grad, g.sum(0))


@pytest.mark.gpu
def test_hip_graphnorm_handworked(dev):
    import gmlm_amd
    norm = gmlm_amd.GraphNorm(2).to(dev)
    with torch.no_grad():
        norm.weight.copy_(torch.tensor(GW)); norm.bias.copy_(torch.tensor(GB)); norm.mean_scale.copy_(torch.tensor(GMS))
    x = torch.tensor(GX, dtype=torch.float32, device=dev, requires_grad=True)
    y = norm(x)                                                  # main.py:273's call form
    np.testing.assert_allclose(y.detach().cpu().numpy(), GOUT, rtol=0, atol=2e-6)
    g = np.arange(8, dtype=np.float64).reshape(4, 2) / 4 - 0.7
    y.backward(torch.tensor(g, dtype=torch.float32, device=dev))
    np.testing.assert_allclose(x.grad.cpu().numpy(), fd_grad(lambda a: np_graphnorm(a, GW, GB, GMS), GX, g), atol=2e-5)
    np.testing.assert_allclose(norm.mean_scale.grad.cpu().numpy(), fd_grad(lambda a: np_graphnorm(GX, GW, GB, a), GMS, g), atol=2e-5)
    np.testing.assert_allclose(norm.weight.grad.cpu().numpy(), fd_grad(lambda a: np_graphnorm(GX, a, GB, GMS), GW, g), atol=2e-5)
    np.testing.assert_allclose(norm.bias.grad.cpu().numpy(), g.sum(0), atol=2e-6)


def _five_relation_case(n=37, e=300, f_in=24, f_out=16, seed=5):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    et = torch.randint(0, 5, (e,), generator=g)
    et[:7] = 4                                                   # relation 4: never produced by the degree bucketing (main.py:260-267)
    et[et == 2] = 3                                              # ... and one relation (2) with no edge at all
    x = torch.randn(n, f_in, generator=g)
    return ei, et, x, g


@pytest.mark.gpu
@pytest.mark.parametrize("idx_dtype", [torch.long, torch.int32])
def test_hip_rgcnconv_reference_geometry_explicit_edge_type(dev, idx_dtype):
    """RGCNConv(in, out, num_relations=5, num_bases=30) exactly as the reference constructs it (main.py:189), called
    with an explicit edge_type that uses relation 4, against the oracle (fp32, fwd + all grads) and the numpy loops."""
    import gmlm_amd
    ei, et, x, g = _five_relation_case()
    conv = gmlm_amd.RGCNConv(24, 16, num_relations=5, num_bases=30)
    ref = O.OracleRGCNConv(24, 16, 5, 30)
    ref.load_state_dict(conv.state_dict(), strict=True)          # same parameter names / shapes as PyG's
    with torch.no_grad():
        conv.bias.uniform_(-0.5, 0.5, generator=g); ref.bias.copy_(conv.bias)
    conv = conv.to(dev)
    xr = x.clone().requires_grad_(True)
    xo = x.clone().to(dev).requires_grad_(True)
    out_ref = ref(xr, ei, et)
    out = conv(xo, ei.to(idx_dtype).to(dev), et.to(dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    edges = [(int(s), int(d), int(t)) for s, d, t in zip(ei[0], ei[1], et)]
    sd = {k: v.detach().double().numpy() for k, v in ref.state_dict().items()}
    np.testing.assert_allclose(out.detach().cpu().numpy(), np_rgcn(x.double().numpy(), edges, sd["weight"], sd["comp"], sd["root"], sd["bias"]),
                               rtol=1e-5, atol=1e-5)
    go = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(go)
    out.backward(go.to(dev))
    np.testing.assert_allclose(xo.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)
    for k, p in conv.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), dict(ref.named_parameters())[k].grad.numpy(), rtol=1e-4, atol=2e-5, err_msg=k)


@pytest.mark.gpu
def test_hip_operators_under_autocast(dev):
    """Under the reference's ``torch.amp.autocast('cuda')`` (main.py:543) the operators run in bf16 (the reduced
    precision these kernels implement; fp16 autocast maps to it too, see nn.compute_dtype): outputs within bf16
    rounding of the fp32 oracle (|err| <= 2^-7 relative to the output scale, 3 roundings: x, H, out)."""
    import gmlm_amd
    ei, et, x, g = _five_relation_case(n=300, e=4000, f_in=64, f_out=48)
    conv = gmlm_amd.RGCNConv(64, 48, num_relations=5, num_bases=30)
    norm = gmlm_amd.GraphNorm(48)
    with torch.no_grad():
        norm.mean_scale.uniform_(0.5, 1.5, generator=g); norm.weight.uniform_(0.5, 1.5, generator=g); norm.bias.uniform_(-1, 1, generator=g)
    ref = O.OracleRGCNConv(64, 48, 5, 30)
    ref.load_state_dict(conv.state_dict(), strict=True)
    z_ref = ref(x, ei, et).detach()
    y_ref = O.graph_norm(z_ref, norm.weight.detach(), norm.bias.detach(), norm.mean_scale.detach())
    conv, norm = conv.to(dev), norm.to(dev)
    for dt in (torch.bfloat16, torch.float16):
        with torch.amp.autocast("cuda", dtype=dt):
            z = conv(x.to(dev), ei.to(dev), et.to(dev))
            y = norm(z)
        assert z.dtype == torch.bfloat16
        scale = float(z_ref.abs().max())
        assert float((z.float().cpu() - z_ref).abs().max()) <= 2.0 ** -7 * scale
        assert float((y.float().cpu() - y_ref).abs().max()) <= 2.0 ** -6 * float(y_ref.abs().max())


@pytest.mark.gpu
def test_hip_graphnorm_pyg_signature_vs_oracle(dev):
    import gmlm_amd
    g = torch.Generator().manual_seed(9)
    x = torch.randn(501, 96, generator=g) * 3 + 1
    norm = gmlm_amd.GraphNorm(96)
    with torch.no_grad():
        norm.mean_scale.uniform_(0.2, 1.8, generator=g); norm.weight.uniform_(0.5, 1.5, generator=g); norm.bias.uniform_(-1, 1, generator=g)
    ref = O.OracleGraphNorm(96)
    ref.load_state_dict(norm.state_dict(), strict=True)
    xr = x.clone().requires_grad_(True)
    xo = x.clone().to(dev).requires_grad_(True)
    norm = norm.to(dev)
    y_ref, y = ref(xr), norm(xo)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    go = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(go)
    y.backward(go.to(dev))
    np.testing.assert_allclose(xo.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)
    for k, p in norm.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), dict(ref.named_parameters())[k].grad.numpy(), rtol=1e-4, atol=1e-4, err_msg=k)
