"""Training-step harness (gmlm_amd.harness; counterpart of main.py:528-563) vs the same loop on the CPU
oracle: three optimiser steps in fp32, dropout 0, masks captured as inputs -> same loss trajectory."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import gmlm_oracle as O
from helpers import oracle_model_from_config

pytestmark = pytest.mark.gpu


def test_three_training_steps_match_oracle():
    import gmlm_amd
    from gmlm_amd import harness
    from test_gpu_model import build_model
    dev = torch.device("cuda:0")
    plm = dict(hidden=128, layers=2, heads=2, inter=256, max_pos=64, vocab=200)
    n, e = 150, 700
    cfg = dict(n=n, e=e, f_in=48, hc=32, c=4, plm=plm, seed=31)
    g = torch.Generator().manual_seed(8)
    x, ei = torch.randn(n, 48, generator=g), torch.randint(0, n, (2, e), generator=g)
    y = torch.randint(0, 4, (n,), generator=g)
    masks = [torch.rand(n, generator=g) < 0.4 for _ in range(3)]
    ids, am = O.synthetic_tokens(n, 12, 200, 3, 2)
    lrs = dict(lr_graph=1e-3, lr_bert=1e-4, lr_other=5e-4, weight_decay=0.01)

    om, _ = oracle_model_from_config(cfg)
    # the oracle registers PLM weights under plm_params.*: build the same three groups by hand
    groups = [[], [], []]
    for name, p in om.named_parameters():
        if name.startswith("plm_params."):
            groups[1].append(p)
        elif any(s in name for s in harness.GNN_PARAM_NAMES):
            groups[0].append(p)
        else:
            groups[2].append(p)
    oopt = torch.optim.AdamW([{"params": groups[0], "lr": 1e-3, "weight_decay": 0.01},
                              {"params": groups[1], "lr": 1e-4, "weight_decay": 0.01},
                              {"params": groups[2], "lr": 5e-4, "weight_decay": 0.01}])
    osched = harness.linear_warmup_schedule(oopt, 1, 10)
    ref_losses = []
    for mk in masks:
        oopt.zero_grad()
        xm = O.soft_masking_gnn_input(x, mk, om.gnn_mask_token_embed, 0.7)
        loss = F.cross_entropy(om(xm, ei, ids, am, mk, plm_batch_size=64)[mk], y[mk], label_smoothing=0.2)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(om.parameters(), 1.0)
        oopt.step(); osched.step()
        ref_losses.append(float(loss.detach()))

    m = build_model(cfg, dev)
    opt = harness.setup_optimizer(m, **lrs)
    assert [len(gp["params"]) for gp in opt.param_groups] == [len(gr) for gr in groups]
    sched = harness.linear_warmup_schedule(opt, 1, 10)
    tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
    losses = []
    for mk in masks:
        r = harness.train_step(m, opt, sched, x.to(dev), ei.to(dev), tokens, y.to(dev), mk.to(dev), beta=0.7,
                               plm_batch_size=64, autocast=False)
        assert not r.skipped
        losses.append(r.loss)
    # step 1 is a pure forward comparison (1e-4); later steps also carry AdamW's sign-sensitive first updates
    assert abs(losses[0] - ref_losses[0]) < 1e-4, (losses, ref_losses)
    np.testing.assert_allclose(losses, ref_losses, rtol=0, atol=2e-3)
    loss, acc, f1 = harness.eval_step(m, x.to(dev), ei.to(dev), tokens, y.to(dev), masks[0].to(dev), plm_batch_size=64,
                                      autocast=False)
    assert np.isfinite(loss) and 0.0 <= acc <= 1.0 and 0.0 <= f1 <= 1.0
    # degree-proportional mask sampling: right count, only base nodes
    base = torch.zeros(n, dtype=torch.bool, device=dev)
    base[:60] = True
    mk = harness.generate_active_node_mask(x.to(dev), ei.to(dev), 0.5, base)
    assert int(mk.sum()) == 30 and not bool(mk[60:].any())


def test_pretrain_step_nt_xent():
    """pretrain_contrastive_gnn iteration (main.py:438-456): two views -> get_graph_embeddings x2 -> NT-Xent,
    loss equal to the reference's chunk loop evaluated on the same embeddings."""
    import gmlm_amd
    from gmlm_amd import harness
    from test_gpu_model import build_model
    dev = torch.device("cuda:0")
    plm = dict(hidden=128, layers=1, heads=2, inter=256, max_pos=64, vocab=200)
    n, e = 203, 900
    cfg = dict(n=n, e=e, f_in=48, hc=32, c=4, plm=plm, seed=5)
    g = torch.Generator().manual_seed(2)
    x, ei = torch.randn(n, 48, generator=g).to(dev), torch.randint(0, n, (2, e), generator=g).to(dev)
    m1, m2 = (torch.rand(n, generator=g) < 0.3).to(dev), (torch.rand(n, generator=g) < 0.3).to(dev)
    m = build_model(cfg, dev).train()
    with torch.no_grad():
        g1 = m.get_graph_embeddings(m.soft_mask_input(x, m1, 0.7), ei)
        g2 = m.get_graph_embeddings(m.soft_mask_input(x, m2, 0.7), ei)
    # reference loop (main.py:113-131) restated here as the checker
    total, ref = g1.size(0), 0.0
    for i in range(0, total, 8):
        a, b = F.normalize(g1[i:i + 8], dim=1), F.normalize(g2[i:i + 8], dim=1)
        bc = a.size(0)
        if bc <= 1:
            continue
        emb = torch.cat([a, b], 0)
        sim = (emb @ emb.t() / 0.5).masked_fill(torch.eye(2 * bc, dtype=torch.bool, device=dev), -float("inf"))
        lab = torch.cat([torch.arange(bc, device=dev) + bc, torch.arange(bc, device=dev)])
        ref += float(F.cross_entropy(sim, lab)) * (bc / total)
    assert abs(float(harness.nt_xent_loss(g1, g2, 0.5, 8)) - ref) < 1e-5
    opt = torch.optim.AdamW([p for nme, p in m.named_parameters() if not nme.startswith("plm_encoder.")], lr=1e-3)
    l0 = harness.pretrain_step(m, opt, x, ei, m1, m2, autocast=False)
    assert abs(l0 - ref) < 1e-4
    l1 = harness.pretrain_step(m, opt, x, ei, m1, m2, autocast=False)
    assert np.isfinite(l1) and l1 < l0            # same views, one optimiser step: the contrastive loss drops


def test_bf16_training_reduces_loss_on_learnable_labels():
    """End-to-end sanity of the bf16 training path (autocast, dropout ON, AdamW, clip, warm-up): labels are a
    function of the node features, so a few dozen full-batch steps must cut the training loss."""
    import gmlm_amd
    from gmlm_amd import harness
    from transformers import BertConfig, BertModel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n, e, f_in, c = 600, 4000, 64, 4
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, f_in, generator=g)
    proto = torch.randn(c, f_in, generator=g)
    y = (x @ proto.t()).argmax(1)
    ei = torch.randint(0, n, (2, e), generator=g)
    ids, am = O.synthetic_tokens(n, 16, 200, 3, 4)
    ids[:, 1] = 5 + y                                   # the text carries the label too
    enc = BertModel(BertConfig(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                               intermediate_size=256, max_position_embeddings=64))
    m = gmlm_amd.GraphTextLM(f_in, 32, c, dropout_rate=0.1, plm_encoder=enc).to(dev)
    tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
    train = torch.zeros(n, dtype=torch.bool)
    train[: n // 2] = True
    xd, eid, yd, td = x.to(dev), ei.to(dev), y.to(dev), train.to(dev)
    opt = harness.setup_optimizer(m, 3e-3, 3e-4, 3e-3, 0.01)
    sched = harness.linear_warmup_schedule(opt, 3, 60)
    losses = []
    for step in range(40):
        mk = harness.generate_active_node_mask(xd, eid, 0.6, td)
        r = harness.train_step(m, opt, sched, xd, eid, tokens, yd, mk, plm_batch_size=4096, autocast=True)
        assert not r.skipped and np.isfinite(r.loss)
        losses.append(r.loss)
    assert np.mean(losses[-5:]) < 0.8 * np.mean(losses[:5]), losses
    # eval mode (dropout off, no soft mask) on the nodes it was trained on: the labels have been fitted.  Held-out
    # accuracy depends on the (random) init on this graph, 0.24-0.92 for the same protocol, so it is not asserted HERE;
    # test_forty_step_trajectory_matches_oracle_and_bf16_tracks_fp32 below pins the protocol instead: loss trajectory
    # and train / held-out accuracy equal to the CPU oracle's, bf16 within a stated budget of fp32.
    loss, acc, f1 = harness.eval_step(m, xd, eid, tokens, yd, td, plm_batch_size=4096)
    assert np.isfinite(loss) and acc > 0.8 and f1 > 0.8


def test_forty_step_trajectory_matches_oracle_and_bf16_tracks_fp32():
    """The evidence behind the sanity test above, as an assertion: the SAME 40-step protocol (learnable labels, fixed
    masks, AdamW with the reference's three groups, clip, warm-up; dropout 0 so that the CPU oracle can follow) gives
    the same loss trajectory on the fp32 HIP path as on the CPU oracle, the same train / held-out accuracy afterwards,
    and the bf16 path tracks the fp32 one within a bf16 budget.  A bf16 / optimiser / backward bug that merely slows
    learning down would pass ``loss drops``; it does not pass this."""
    import gmlm_amd
    from gmlm_amd import harness
    from test_gpu_model import build_model
    dev = torch.device("cuda:0")
    plm = dict(hidden=128, layers=2, heads=2, inter=256, max_pos=64, vocab=200)
    n, e, f_in, c = 600, 4000, 64, 4
    cfg = dict(n=n, e=e, f_in=f_in, hc=32, c=c, plm=plm, seed=31, max_len=16)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, f_in, generator=g)
    proto = torch.randn(c, f_in, generator=g)
    y = (x @ proto.t()).argmax(1)
    ei = torch.randint(0, n, (2, e), generator=g)
    ids, am = O.synthetic_tokens(n, 16, 200, 3, 4)
    ids[:, 1] = 5 + y
    train = torch.zeros(n, dtype=torch.bool)
    train[: n // 2] = True
    masks = []
    for _ in range(40):
        mk = torch.zeros(n, dtype=torch.bool)
        mk[torch.randperm(n // 2, generator=g)[:180]] = True
        masks.append(mk)
    lrs = dict(lr_graph=3e-3, lr_bert=3e-4, lr_other=3e-3, weight_decay=0.01)

    # CPU oracle: the same loop as main.py:528-563
    om, _ = oracle_model_from_config(cfg)
    groups = [[], [], []]
    for name, p in om.named_parameters():
        groups[1 if name.startswith("plm_params.") else (0 if any(s in name for s in harness.GNN_PARAM_NAMES) else 2)].append(p)
    oopt = torch.optim.AdamW([{"params": groups[0], "lr": 3e-3, "weight_decay": 0.01}, {"params": groups[1], "lr": 3e-4, "weight_decay": 0.01},
                              {"params": groups[2], "lr": 3e-3, "weight_decay": 0.01}])
    osched = harness.linear_warmup_schedule(oopt, 3, 60)
    ref = []
    for mk in masks:
        oopt.zero_grad()
        loss = F.cross_entropy(om(O.soft_masking_gnn_input(x, mk, om.gnn_mask_token_embed, 0.7), ei, ids, am, mk, plm_batch_size=4096)[mk],
                               y[mk], label_smoothing=0.2)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(om.parameters(), 1.0)
        oopt.step(); osched.step()
        ref.append(float(loss.detach()))
    with torch.no_grad():
        om.eval()
        ref_acc = {}
        for nm, msk in (("train", train), ("held", ~train)):
            ref_acc[nm] = float((om(x, ei, ids, am, msk, plm_batch_size=4096)[msk].argmax(1) == y[msk]).float().mean())

    tokens = gmlm_amd.TokenizedTexts.from_mask(ids.to(dev), am.to(dev))
    xd, eid, yd = x.to(dev), ei.to(dev), y.to(dev)

    def run(autocast):
        m = build_model(cfg, dev)
        opt = harness.setup_optimizer(m, **lrs)
        sched = harness.linear_warmup_schedule(opt, 3, 60)
        traj = [harness.train_step(m, opt, sched, xd, eid, tokens, yd, mk.to(dev), plm_batch_size=4096, autocast=autocast).loss for mk in masks]
        acc = {nm: harness.eval_step(m, xd, eid, tokens, yd, msk.to(dev), plm_batch_size=4096, autocast=autocast)[1]
               for nm, msk in (("train", train), ("held", ~train))}
        return np.array(traj), acc

    t32, a32 = run(False)
    print(f"\n40 steps: oracle {ref[0]:.4f} -> {ref[-1]:.4f}, fp32 HIP max|d| {np.abs(t32 - np.array(ref)).max():.2e}; acc oracle {ref_acc} fp32 {a32}")
    assert ref[-1] < 0.8 * ref[0]                                     # the protocol does learn
    np.testing.assert_allclose(t32, ref, rtol=0, atol=1e-3)          # same trajectory (measured: 1.3e-6 over all 40 steps)
    assert abs(a32["train"] - ref_acc["train"]) <= 0.01 and abs(a32["held"] - ref_acc["held"]) <= 0.02, (a32, ref_acc)
    tbf, abf = run(True)
    print(f"bf16 vs fp32 trajectory max|d| {np.abs(tbf - t32).max():.2e}; acc bf16 {abf}")
    np.testing.assert_allclose(tbf, t32, rtol=0, atol=2e-2)           # 8-bit mantissa operands, 40 optimiser steps (measured: 6.6e-3)
    assert abs(abf["train"] - a32["train"]) <= 0.03 and abs(abf["held"] - a32["held"]) <= 0.05, (abf, a32)
