"""CPU checks of the oracle's bf16-emulating mode (oracle/bf16_emulation.py): with the rounding switched off it must BE the
fp32 oracle (same logits, same gradients: the restated order of operations is the oracle's algorithm), with it on it
must stay inside a bf16 budget of the fp32 oracle.  The GPU tests then hold the bf16 HIP path against this mode tightly."""
import torch
import torch.nn.functional as F

import bf16_emulation as E
import gmlm_oracle as O
from helpers import oracle_model_from_config


def _case():
    plm = dict(hidden=128, layers=2, heads=2, inter=256, max_pos=64, vocab=200)
    cfg = dict(n=183, e=2400, f_in=61, hc=32, c=5, plm=plm, seed=7)          # f_in = 61: column padding of the stored input
    g = torch.Generator().manual_seed(0)
    x = torch.randn(cfg["n"], cfg["f_in"], generator=g)
    w = (torch.arange(cfg["n"], dtype=torch.float32) + 1).pow(-0.8)          # skewed out-degrees: all four edge types occur
    ei = torch.stack([torch.multinomial(w, cfg["e"], True, generator=g), torch.randint(0, cfg["n"], (cfg["e"],), generator=g)])
    y = torch.randint(0, 5, (cfg["n"],), generator=g)
    mask = torch.rand(cfg["n"], generator=g) < 0.4
    ids, am = O.synthetic_tokens(cfg["n"], 16, 200, 1, 2)
    om, _ = oracle_model_from_config(cfg)
    assert sorted(set(O.edge_types_from_degree(ei, cfg["n"]).tolist())) == [0, 1, 2, 3]
    return om, x, ei, y, mask, ids, am


def _grads(om):
    return {k: p.grad.clone() for k, p in om.named_parameters() if p.grad is not None}


def test_emulation_without_rounding_is_the_fp32_oracle(monkeypatch):
    om, x, ei, y, mask, ids, am = _case()
    ref = om(O.soft_masking_gnn_input(x, mask, om.gnn_mask_token_embed, 0.7), ei, ids, am, mask, plm_batch_size=16)
    loss = F.cross_entropy(ref[mask], y[mask], label_smoothing=0.2)
    loss.backward()
    g32 = _grads(om)
    om.zero_grad(set_to_none=True)
    monkeypatch.setattr(E, "r", lambda t: t)
    lg = E.forward(om, x, ei, ids, am, mask)
    l2 = F.cross_entropy(lg[mask], y[mask], label_smoothing=0.2)
    l2.backward()
    assert float((lg - ref).abs().max()) < 1e-5 and abs(float(l2) - float(loss)) < 1e-6
    ge = _grads(om)
    assert set(ge) == set(g32)
    gmax = max(float(v.norm()) for v in g32.values())
    for k, b in g32.items():
        # relative to the tensor's own norm; the key biases are analytically zero (softmax is shift invariant): absolute
        assert float((ge[k] - b).norm()) <= 1e-4 * float(b.norm()) + 1e-7 * gmax, k


def test_emulation_with_rounding_stays_in_the_bf16_budget():
    om, x, ei, y, mask, ids, am = _case()
    ref = om(O.soft_masking_gnn_input(x, mask, om.gnn_mask_token_embed, 0.7), ei, ids, am, mask, plm_batch_size=16)
    F.cross_entropy(ref[mask], y[mask], label_smoothing=0.2).backward()
    g32 = _grads(om)
    om.zero_grad(set_to_none=True)
    lg = E.forward(om, x, ei, ids, am, mask)
    F.cross_entropy(lg[mask], y[mask], label_smoothing=0.2).backward()
    d = float((lg - ref).abs().max())
    assert 1e-4 < d < 5e-2                                               # really rounded, and not more than bf16 explains
    assert bool((E.r(lg) == lg).all())                                   # the logits are bf16 values
    gmax = max(float(v.norm()) for v in g32.values())
    for k, p in om.named_parameters():
        b = g32.get(k)
        if b is None or float(b.norm()) < 1e-6 * gmax:
            continue
        cos = float(torch.dot(p.grad.flatten(), b.flatten()) / (p.grad.norm() * b.norm()))
        assert cos > 0.99, (k, cos)


def test_pipelined_forward_reference_max_schedule():
    """The emulated streaming forward re-references a whole 32-query wave when one of its queries jumps by more than 2^6
    (log2 domain); whatever the schedule, the result is the softmax: against a dense fp64 evaluation."""
    g = torch.Generator().manual_seed(3)
    h, n, d = 2, 200, 16
    q, k, v = (E.r(torch.randn(h, n, d, generator=g)) for _ in range(3))
    k[0, 150] = q[0, 5] * 9.0                                           # a late, large score for one query
    o, o_lo, lse = E._pipe_forward(q, k, v, d ** -0.5)
    assert float(o_lo.abs().max()) <= 2 ** -8 * float(o.abs().max())
    s = (q.double() @ k.double().transpose(-1, -2)) * d ** -0.5
    ref = torch.softmax(s, -1) @ v.double()
    assert float((o.double() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
    assert float((lse.double() - torch.logsumexp(s, -1)).abs().max()) < 5e-2   # q * scale * log2e is rounded to bf16: |s| * 2^-9
