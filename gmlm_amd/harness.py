"""Training / evaluation step around the hot path — the thin counterpart of ``train_model`` /
``evaluate_model`` (main.py:488-720) that turns the kernel path into an end-to-end step.

Only the step logic is mirrored (mask sampling -> soft mask -> forward -> CE(label_smoothing=0.2) ->
backward -> clip_grad_norm_(1.0) -> AdamW(3 name-based groups) -> linear warm-up), with stock
``torch.optim`` / schedulers exactly like the reference; logging, wandb, early stopping bookkeeping and the
experiment driver stay with the caller.
"""
from __future__ import annotations

import random
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F
import torch.optim as optim

from . import ops

GNN_PARAM_NAMES = ['rgcn1', 'rgcn2', 'rgcn3', 'gnorm1', 'gnorm2', 'gnorm3', 'residual_proj1', 'residual_proj2',
                   'residual_proj3']   # main.py:379 (rgcn4 / gnorm4 fall into "other", as in the reference)


def setup_optimizer(model, lr_graph, lr_bert, lr_other, weight_decay):
    """main.py:375-398: AdamW with three parameter groups selected by NAME."""
    graph_params, bert_params, other_params = [], [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if name.startswith('plm_encoder.'):
            bert_params.append(param)
        elif any(g in name for g in GNN_PARAM_NAMES):
            graph_params.append(param)
        else:
            other_params.append(param)
    return optim.AdamW([
        {'params': graph_params, 'lr': lr_graph, 'weight_decay': weight_decay},
        {'params': bert_params, 'lr': lr_bert, 'weight_decay': 0.01},
        {'params': other_params, 'lr': lr_other, 'weight_decay': weight_decay}])


def linear_warmup_schedule(optimizer, num_warmup_steps, num_training_steps):
    """``transformers.get_linear_schedule_with_warmup`` (main.py:504) as a plain LambdaLR."""
    def lr_lambda(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)


def generate_active_node_mask(x, edge_index, mask_ratio, base_mask: Optional[torch.Tensor] = None):
    """main.py:47-89 on the device: base nodes -> degree-proportional sampling without replacement.
    The degree histogram is the K1 kernel (bit-exact); the sampling itself is torch's RNG, as in the
    reference (not reproducible across devices, so parity tests pass masks in as inputs)."""
    n = x.size(0)
    dev = x.device
    base = base_mask.nonzero(as_tuple=False).reshape(-1) if base_mask is not None else torch.arange(n, device=dev)
    nb = base.numel()
    out = torch.zeros(n, dtype=torch.bool, device=dev)
    if nb == 0:
        return out
    num_select = max(1, min(int(mask_ratio * nb), nb))
    deg = ops.degree(edge_index[0].to(torch.long), n)[base]
    tot = deg.sum()
    if float(tot) == 0:
        sel = base[torch.randperm(nb, device=dev)[:num_select]]
    else:
        probs = torch.nan_to_num(deg / tot, nan=1.0 / nb)
        sel = base[torch.multinomial(probs, num_select, replacement=False)]
    out[sel] = True
    return out


def nt_xent_loss(z1, z2, temperature=0.5, batch_size=8):
    """main.py:102-136 (chunked NT-Xent) without the Python loop over chunks: all full chunks of
    ``batch_size`` rows go through one batched [chunks, 2b, 2b] similarity + cross-entropy; the ragged
    last chunk (if any) is handled the same way on its own.  Chunks of a single row are skipped and every
    chunk is weighted by its share of the rows, exactly like the reference."""
    total = z1.size(0)
    if total == 0:
        return torch.tensor(0.0, device=z1.device, requires_grad=True)
    b = batch_size if batch_size is not None else total

    def chunk_losses(a1, a2, bc):                       # a*: [chunks, bc, P] -> mean CE per chunk
        e = torch.cat([F.normalize(a1, dim=-1), F.normalize(a2, dim=-1)], dim=1)            # [chunks, 2bc, P]
        sim = torch.bmm(e, e.transpose(1, 2)) / temperature
        eye = torch.eye(2 * bc, dtype=torch.bool, device=z1.device)
        sim = sim.masked_fill(eye, -float('inf'))
        pos = torch.arange(bc, device=z1.device)
        labels = torch.cat([pos + bc, pos]).expand(sim.size(0), -1)
        return F.cross_entropy(sim.reshape(-1, 2 * bc), labels.reshape(-1), reduction='none').view(sim.size(0), -1).mean(1)

    nfull, rem = divmod(total, b)
    loss, used = None, False
    if nfull and b > 1:
        lf = chunk_losses(z1[:nfull * b].view(nfull, b, -1).float(), z2[:nfull * b].view(nfull, b, -1).float(), b)
        loss, used = lf.sum() * (b / total), True
    if rem > 1:
        lr = chunk_losses(z1[nfull * b:].unsqueeze(0).float(), z2[nfull * b:].unsqueeze(0).float(), rem)
        loss = lr.sum() * (rem / total) + (loss if loss is not None else 0.0)
        used = True
    if not used:
        return torch.tensor(0.0, device=z1.device, requires_grad=True)
    return loss


def _all_flags(model, *flags: bool):
    """AND of per-rank boolean decisions over the partition group (identity on a single GPU): every rank must take
    the same branch, otherwise the ranks that go on sit in the halo / GraphNorm / K|V collectives forever."""
    if model.dist is None:
        return flags
    t = model.dist.all_reduce_min(torch.tensor([1.0 if f else 0.0 for f in flags]))
    return tuple(bool(v > 0.5) for v in t.tolist())


def pretrain_step(model, optimizer, x, edge_index, mask1, mask2, *, beta=0.7, temperature=0.5, autocast=True):
    """One iteration of ``pretrain_contrastive_gnn`` (main.py:438-456): two soft-masked views ->
    ``get_graph_embeddings`` twice (the graph preprocessing is cached, not redone per view) -> NT-Xent.

    Node partition (``model.dist``): the reference chunks the N embeddings into groups of 8 consecutive rows
    (main.py:112-131); the embeddings of both views are therefore all-gathered (rows in global order) and every
    rank evaluates the SAME global loss, so the gradient that flows back through the reduce-scatter is the
    single-GPU gradient and the all-reduced parameter gradients need no rescaling."""
    model.train()
    optimizer.zero_grad(set_to_none=True)
    with torch.amp.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
        g1 = model.get_graph_embeddings(model.soft_mask_input(x, mask1, beta), edge_index, edge_type=None)
        g2 = model.get_graph_embeddings(model.soft_mask_input(x, mask2, beta), edge_index, edge_type=None)
        if model.dist is not None:
            g1 = model.dist.all_gather_rows(g1.unsqueeze(0)).squeeze(0)
            g2 = model.dist.all_gather_rows(g2.unsqueeze(0)).squeeze(0)
        loss = nt_xent_loss(g1, g2, temperature=temperature, batch_size=8)
        if model.dist is not None:
            loss = loss / model.dist.plan.world          # every rank holds the same loss; the gradient all-reduce SUMS them
    (finite,) = _all_flags(model, bool(torch.isfinite(loss)))
    if not finite:
        return float(loss)
    if model.dist is not None:
        model.dist.grad_buckets(model).prepare()          # gradients as views of flat buckets, reduced under backward
    loss.backward()
    scale = 1.0
    if model.dist is not None:
        model.dist.grad_buckets(model).finish()
        scale = float(model.dist.plan.world)
    optimizer.step()
    return float(loss.detach()) * scale


@dataclass
class StepResult:
    loss: float
    accuracy: float
    skipped: bool = False


def train_step(model, optimizer, scheduler, x, edge_index, texts, y, active_mask, *, beta=0.7, plm_batch_size=32,
               grad_clip_norm=1.0, autocast: bool = True, label_smoothing=0.2) -> StepResult:
    """One iteration of the epoch loop main.py:528-563 (full-batch: one forward + backward per epoch).

    Node partition (``model.dist``; x / texts / y / active_mask are this rank's rows): the loss is the GLOBAL mean
    over all active nodes (local sum / all-reduced count), so the summed gradients equal the single-GPU ones and
    ``clip_grad_norm_`` sees the same norm; the skip decisions (no active node anywhere, non-finite loss anywhere)
    are taken collectively; a rank without a local active node still runs forward and backward with a zero loss
    term, because its peers need it in the halo / GraphNorm / K|V exchanges."""
    model.train()
    optimizer.zero_grad(set_to_none=True)
    dist_ctx = model.dist
    n_local = int(active_mask.sum()) if dist_ctx is not None else None
    if dist_ctx is None:
        if not bool(active_mask.any()):
            return StepResult(float('nan'), 0.0, True)
        n_active = None
    else:
        cnt = torch.tensor([float(n_local)])
        dist_ctx.all_reduce_sum(cnt)
        n_active = int(cnt.item())
        if n_active == 0:                                  # the same decision on every rank
            return StepResult(float('nan'), 0.0, True)
    with torch.amp.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
        xm = model.soft_mask_input(x, active_mask, beta)
        logits = model(xm, edge_index, texts, active_mask, edge_type=None, plm_batch_size=plm_batch_size)
        idx = model.active_index                # ascending ids of the active nodes, built by the forward (same rows as
        if idx is None:                         # logits[active_mask], without its sync); None = no local active node
            idx = torch.zeros(0, dtype=torch.long, device=logits.device)
        la, ya = logits.index_select(0, idx), y.index_select(0, idx)
        if dist_ctx is None:
            loss = F.cross_entropy(la, ya, label_smoothing=label_smoothing)
        else:
            loss = F.cross_entropy(la, ya, label_smoothing=label_smoothing, reduction='sum') / n_active if idx.numel() \
                else logits.sum() * 0.0
    with torch.no_grad():
        hits = (la.argmax(1) == ya).float().sum() if idx.numel() else torch.zeros((), device=logits.device)
        if dist_ctx is None:
            acc, loss_value = float(hits / max(idx.numel(), 1)), loss.detach()
        else:
            agg = torch.stack([hits.cpu(), loss.detach().float().cpu()])
            dist_ctx.all_reduce_sum(agg)
            acc, loss_value = float(agg[0]) / n_active, agg[1]
    (finite,) = _all_flags(model, bool(torch.isfinite(loss)))
    if not finite:
        return StepResult(float(loss_value), acc, True)
    if dist_ctx is not None:
        dist_ctx.grad_buckets(model).prepare()            # gradients as views of flat buckets, reduced under backward
    loss.backward()
    if dist_ctx is not None:
        dist_ctx.grad_buckets(model).finish()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=grad_clip_norm)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return StepResult(float(loss_value), acc)


@torch.no_grad()
def eval_step(model, x, edge_index, texts, y, mask, *, plm_batch_size=32, autocast: bool = True):
    """main.py:584-616 / 669-720: eval forward on ``mask`` -> (loss, accuracy, macro-F1)."""
    model.eval()
    with torch.amp.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
        logits = model(x, edge_index, texts, mask, edge_type=None, plm_batch_size=plm_batch_size)
    lg, lab = logits[mask], y[mask]
    if lab.numel() == 0:
        return float('nan'), 0.0, 0.0
    loss = float(F.cross_entropy(lg, lab))
    pred = lg.argmax(1)
    acc = float((pred == lab).float().mean())
    c = int(max(int(lab.max()), int(pred.max()))) + 1
    f1s = []
    for k in range(c):
        tp = float(((pred == k) & (lab == k)).sum())
        fp = float(((pred == k) & (lab != k)).sum())
        fn = float(((pred != k) & (lab == k)).sum())
        if tp + fp + fn > 0:
            f1s.append(2 * tp / (2 * tp + fp + fn))
    return loss, acc, float(sum(f1s) / max(len(f1s), 1))


def train_model(model, x, edge_index, texts, y, train_mask, val_mask=None, *, num_epochs=500,
                active_node_mask_ratio_min=0.2, active_node_mask_ratio_max=0.4, beta_soft_mask_gnn=0.7, lr_graph=1e-3,
                lr_bert=1e-5, lr_other=1e-4, weight_decay=0.01, patience=20, warmup_ratio=0.1, grad_clip_norm=1.0,
                plm_batch_size=32, autocast=True, on_epoch=None):
    """Epoch loop with the reference's hyper-parameter names (main.py:488-493); returns the loss list."""
    optimizer = setup_optimizer(model, lr_graph, lr_bert, lr_other, weight_decay)
    scheduler = linear_warmup_schedule(optimizer, int(num_epochs * warmup_ratio), num_epochs)
    losses, best_f1, bad = [], 0.0, 0
    for epoch in range(num_epochs):
        ratio = random.uniform(active_node_mask_ratio_min, active_node_mask_ratio_max)
        mask = generate_active_node_mask(x, edge_index, ratio, train_mask)
        res = train_step(model, optimizer, scheduler, x, edge_index, texts, y, mask, beta=beta_soft_mask_gnn,
                         plm_batch_size=plm_batch_size, grad_clip_norm=grad_clip_norm, autocast=autocast)
        losses.append(res.loss)
        if val_mask is not None and bool(val_mask.any()) and (epoch % 5 == 0 or epoch == num_epochs - 1):
            _, _, f1 = eval_step(model, x, edge_index, texts, y, val_mask, plm_batch_size=plm_batch_size, autocast=autocast)
            if f1 > best_f1:
                best_f1, bad = f1, 0
            else:
                bad += 1
            if bad >= patience:
                break
        if on_epoch is not None:
            on_epoch(epoch, res)
    return losses
