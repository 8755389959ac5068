"""hipGraph capture of the static-shape parts of a ``GraphTextLM`` step (launch-bound small configurations).

A full-batch step on a fixed graph launches the same GNN kernels (4 x RGCN block + multi-scale fusion: main.py:250-320) and
the same head kernels (2 x CrossAttention + fusion MLP + classifier: main.py:360-372) with the same shapes every step; on
Cornell / Chameleon-size graphs those are a few hundred launches of 3-30 us each and the step is bound by the host's launch
rate, not by the GPU.  ``capture(model, x_sample, edge_index)`` records both regions (forward AND backward) once with
``torch.cuda.make_graphed_callables`` — the ctypes launches of libgmlm_hip go to the capturing stream like any other kernel —
and ``model.forward`` replays them.

The text encoder in between sees a different active set every step (main.py:532-534 draws the mask per epoch), so its packed
token count moves.  It is recorded per SIZE BUCKET instead: ``model.bucketed_layout`` pads the packed batch with dummy [PAD]
sequences to a multiple of 64 sequences / 2,048 tokens / 32 attention work items, every index the pass needs is computed on
the device from three small per-step tables (sequence lengths, node of each sequence, cumulative lengths), and
``GraphedStep.encoder`` keeps one hipGraph (forward + backward, weight casts included) per bucket in a small LRU.  Replaying
copies the step's tables into the graph's static inputs.  Batches that need more than one micro-batch run eagerly.

By default the three pieces are ONE recording per bucket (``GraphedStep.step``): text encoder, GNN and the second
CrossAttention are issued on side streams inside the capture, so the graph has parallel branches and the device runs their
3-30 us kernels side by side (Chameleon-size: 6.3 instead of 8.2 ms/step); ``whole_step=False`` keeps the three linear recordings
and ``concurrent=True`` replays two of them on two streams at once.  All modes equal the eager step bit for bit.

Dropout under replay: a captured kernel argument is a constant, so the host seeds drawn at capture time would repeat the same
masks forever.  Every dropout kernel therefore takes ``(seed, seed_dev)`` and uses ``seed + *seed_dev`` (include/gmlm_hip.h);
the capture points ``seed_dev`` at a device counter that the FIRST kernel of the GNN forward graph increments.  Forward,
backward (and a checkpoint recompute) of one replay read the same counter value, the next replay a new one.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops


class _Region(nn.Module):
    """One capturable region: a callable over tensors + the parameters it touches (shared with the model, so the captured
    backward accumulates into the model's own ``.grad``)."""

    def __init__(self, fn, params, counter: torch.Tensor, bump: bool):
        super().__init__()
        self._fn = fn
        self.params = nn.ParameterList(params)
        self._counter = counter
        self._bump = bump

    def forward(self, *args):
        prev = ops.SEED_DEVICE
        ops.SEED_DEVICE = self._counter            # baked into the captured kernels' arguments as a pointer
        try:
            if self._bump:
                self._counter.add_(1)              # captured: every forward replay moves the seed
            return self._fn(*args)
        finally:
            ops.SEED_DEVICE = prev


def _touched_params(model: nn.Module, run) -> list:
    """Parameters that receive a gradient from ``run()`` (one eager forward + backward)."""
    saved = {n: p.grad for n, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    run()
    used = [p for p in model.parameters() if p.grad is not None]
    for n, p in model.named_parameters():
        p.grad = saved[n]
    return used


class GraphedStep:
    """Holds the two captured regions of one (model, graph) pair.  ``gnn(xm)`` -> fp32 [N, P] graph embeddings,
    ``head(gnn_embeds, plm_embeds)`` -> logits; both are autograd-aware graph replays."""

    def __init__(self, model, xm_sample: torch.Tensor, edge_index: torch.Tensor, edge_type: Optional[torch.Tensor] = None,
                 encoder: bool = True, whole_step: bool = True, concurrent: bool = False):
        if model.dist is not None:
            raise ValueError("hipGraph capture is single-GPU: collectives of the node partition are not captured")
        if not model.training:
            raise ValueError("capture in training mode (evaluation steps are not launch-bound enough to matter)")
        dev = xm_sample.device
        self.key = (tuple(xm_sample.shape), xm_sample.dtype, edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape))
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.encoder_enabled = bool(encoder)
        self.encoder_buckets = 4                         # recordings kept (LRU)
        self._encoders = {}                              # bucket key -> (token set, graphed region)
        self.whole_step = bool(whole_step) and self.encoder_enabled
        # replay the GNN recording and the encoder recording on two streams at once (model.forward): the recordings themselves
        # stay linear; the seed bump then happens eagerly AHEAD of the fork instead of inside the GNN recording
        self.concurrent = bool(concurrent) and self.encoder_enabled and not self.whole_step
        self._steps = {}                                 # bucket key -> (token set, graphed whole-forward region)
        self._side = torch.cuda.Stream(device=dev)
        self._side2 = torch.cuda.Stream(device=dev)
        self._graph_args = (edge_index, edge_type)
        model.graph(edge_index, xm_sample.size(0), edge_type).active_index    # CSR build (sort, host syncs) happens before capture
        p_dim = model.plm_encoder.config.hidden_size
        n = xm_sample.size(0)

        def gnn_fn(xm):
            return model.get_graph_embeddings(xm, edge_index, edge_type)

        def head_fn(g, t):
            return model.head(g, t)

        xs = xm_sample.detach().clone().requires_grad_(True)
        gs = torch.randn(n, p_dim, device=dev).requires_grad_(True)
        ts = torch.randn(n, p_dim, device=dev).requires_grad_(True)
        gnn_params = _touched_params(model, lambda: gnn_fn(xs).sum().backward())
        head_params = _touched_params(model, lambda: head_fn(gs, ts).sum().backward())
        xs.grad = gs.grad = ts.grad = None
        regions = (_Region(gnn_fn, gnn_params, self.counter, not self.concurrent), _Region(head_fn, head_params, self.counter, False))
        self.gnn, self.head = torch.cuda.make_graphed_callables(regions, ((xs,), (gs, ts)), num_warmup_iters=2)

    def encoder(self, model, key, tokens, tables):
        """Replay (recording it first if needed) the encoder pass of bucket ``key`` over this step's index tables."""
        hit = self._encoders.get(key)
        if hit is None or hit[0] is not tokens:
            def enc_fn(*t):
                return model.encode_packed_static(tokens, key, *t)

            samples = tuple(t.clone() for t in tables)
            used = _touched_params(model, lambda: enc_fn(*samples).sum().backward())
            region = torch.cuda.make_graphed_callables(_Region(enc_fn, used, self.counter, False), samples, num_warmup_iters=2)
            while len(self._encoders) >= self.encoder_buckets:
                self._encoders.pop(next(iter(self._encoders)))              # oldest recording: its graph and activations are freed
            hit = (tokens, region)
        else:
            self._encoders.pop(key)
        self._encoders[key] = hit                                           # most recently used last
        return hit[1](*tables)

    def step(self, model, key, tokens, xm, tables):
        """Replay (recording it first if needed) the WHOLE forward of bucket ``key``: text encoder and GNN as two branches that
        fork after the seed bump and join ahead of the head, so the device runs their (small) kernels side by side; autograd
        runs each backward node on its forward stream, so the recorded backward has the same two branches."""
        hit = self._steps.get(key)
        if hit is None or hit[0] is not tokens:
            side = self._side
            edge_index, edge_type = self._graph_args

            def step_fn(x, *t):
                model._branch_stream = self._side2
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)                                        # fork (behind the seed bump of _Region.forward)
                # the branch on the side stream is issued LAST: autograd then runs its backward FIRST (latest forward nodes
                # first), so when the engine reaches the GNN's backward the encoder's is already queued on `side` and waits for
                # nothing but the head's gradient
                gnn = model.get_graph_embeddings(x, edge_index, edge_type)
                with torch.cuda.stream(side):
                    plm = model.encode_packed_static(tokens, key, *t)
                cur.wait_stream(side)                                        # join
                # plm's block belongs to `side`'s pool but its last readers (the head, and the head's backward through its saved
                # tensors) run on `cur`: without this the allocator hands the block to the next allocation on `side` - an encoder
                # backward temporary, on the branch that runs BESIDE the head's backward - as soon as the tensor dies
                plm.record_stream(cur)
                try:
                    return model.head(gnn, plm)
                finally:
                    model._branch_stream = None

            xs = xm.detach().clone().requires_grad_(True)
            samples = (xs,) + tuple(t.clone() for t in tables)
            used = _touched_params(model, lambda: step_fn(*samples).sum().backward())
            xs.grad = None
            region = torch.cuda.make_graphed_callables(_Region(step_fn, used, self.counter, True), samples, num_warmup_iters=2)
            while len(self._steps) >= self.encoder_buckets:
                self._steps.pop(next(iter(self._steps)))
            hit = (tokens, region)
        else:
            self._steps.pop(key)
        self._steps[key] = hit
        return hit[1](xm, *tables)

    def matches(self, xm: torch.Tensor, edge_index: torch.Tensor) -> bool:
        return self.key == (tuple(xm.shape), xm.dtype, edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape))


def capture(model, xm_sample: torch.Tensor, edge_index: torch.Tensor, edge_type: Optional[torch.Tensor] = None,
            encoder: bool = True, whole_step: bool = True, concurrent: bool = False) -> GraphedStep:
    """Record the GNN and head regions for this (soft-masked input shape, graph) and attach them to ``model``:
    ``model.forward`` replays them whenever it is called in training mode with an input of the same shape / dtype and the
    same ``edge_index`` tensor; any other call runs eagerly.  ``encoder``: also record the text encoder, per size bucket, the
    first time a step needs that bucket; ``whole_step`` (default): record encoder + GNN + head as ONE graph per bucket whose
    encoder and GNN branches (and the second CrossAttention) run side by side (``GraphedStep.step``; a step whose text batch
    needs several micro-batches falls back to the separate regions); ``whole_step=False``: three linear recordings replayed one
    after the other, or - ``concurrent`` - GNN and encoder recordings replayed on two streams at once.  Every mode returns the
    eager step's logits and gradients bit for bit (tests/test_gpu_graphs.py; 800-step soaks per mode with
    tools/dev/replay_diff.py).  (Rounds 2-3 carried an open issue here: with two streams busy, ``rgcn4.comp``'s gradient came
    out wrong in about one step of thirty.  Cause: packed-fp32 code that the SLP vectoriser made of the basis backward kernel's
    accumulation - csrc/Makefile)  ``model.release_hip_graphs()`` drops the recordings."""
    model._graphed = None
    g = GraphedStep(model, xm_sample, edge_index, edge_type, encoder, whole_step, concurrent)
    model._graphed = g
    return g
