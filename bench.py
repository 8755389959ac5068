#!/usr/bin/env python3
"""bench.py — nodes/sec of one full-batch GraphTextLM forward+backward (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]        # N>1: launched by torch.distributed.run

Workload (SURVEY.md §8d "S3", BASELINE configs[2], the config the metric is quoted on): synthetic
Squirrel-size graph (5,201 nodes, 217,073 edges, F_in=2089, 5 classes), hidden_channels=768,
BERT-base-geometry text encoder (768 x 12 layers x 12 heads, random init), dropout as in the reference
(0.3 model / 0.1 encoder), ~1,250 active text nodes of 16..128 tokens, bf16 GEMM/attention operands with
fp32 accumulation and statistics.  A "step" = soft-mask -> forward -> CE(label_smoothing=0.2) -> backward
(the optimiser is outside the metric: SURVEY.md §8d).  With N > 1 every rank owns a Squirrel-size
partition of an N x 5,201-node graph (1-D node partition, halo exchange + RCCL collectives): weak scaling.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (the RGCN aggregation kernel, HBM-bound,
timed with HIP events on the launch stream inside the timed region), "kernels" (the other hand-written
kernels, same method), "cpu_baseline" (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md): 8.0 TB/s
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16
MFMA_F32_PEAK_TF = 157.3

WORKLOADS = {  # name: (N, E, F_in, C)
    "cornell": (183, 298, 1703, 5),
    "chameleon": (2277, 36101, 2325, 5),
    "squirrel": (5201, 217073, 2089, 5),
}


def synthetic(name, n_parts=1):
    """Seeded synthetic graph of the named size; with n_parts > 1 the node/edge counts scale by n_parts."""
    n, e, f_in, c = WORKLOADS[name]
    n, e = n * n_parts, e * n_parts
    g = torch.Generator().manual_seed(1000 + list(WORKLOADS).index(name) + 2)
    x = torch.randn(n, f_in, generator=g)
    ei = torch.randint(0, n, (2, e), generator=g, dtype=torch.long)
    y = torch.randint(0, c, (n,), generator=g)
    perm = torch.randperm(n, generator=g)
    train = torch.zeros(n, dtype=torch.bool)
    train[perm[: int(0.48 * n)]] = True
    active = train & (torch.rand(n, generator=g) < 0.5)
    return dict(n=n, e=e, f_in=f_in, c=c, x=x, edge_index=ei, y=y, active=active)


def synthetic_tokens(n, max_len, vocab, seed, min_len=16):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(min(min_len, max_len), max_len + 1, (n,), generator=g)
    ids = torch.randint(5, vocab, (n, max_len), generator=g)
    am = (torch.arange(max_len)[None, :] < lens[:, None]).long()
    return ids * am, am


def build_model(args, data, dev):
    from transformers import BertConfig, BertModel
    import gmlm_amd
    cfg = BertConfig(vocab_size=args.vocab, hidden_size=args.plm_hidden, num_hidden_layers=args.plm_layers,
                     num_attention_heads=args.plm_hidden // 64, intermediate_size=4 * args.plm_hidden,
                     max_position_embeddings=max(512, args.max_len))
    torch.manual_seed(0)
    enc = BertModel(cfg)
    cd = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    m = gmlm_amd.GraphTextLM(data["f_in"], args.hc, data["c"], dropout_rate=0.3, plm_encoder=enc,
                             plm_max_length=args.max_len, compute_dtype=cd)
    return m.to(dev).train()


def cpu_baseline(args, data, ids, am):
    """Oracle (oracle/gmlm_oracle.py, CPU fp32) forward+backward on the same graph; the PLM leg runs on a
    bounded sample of active nodes and is scaled linearly to all of them (it is row-wise independent)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gmlm_oracle as O
    from helpers import bert_state_template, model_state_template
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))                  # the 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(threads)
    say = lambda msg: print(f"[cpu_baseline] {msg}", file=sys.stderr, flush=True)
    hc = min(args.hc, args.cpu_hc)
    plm = dict(hidden=args.plm_hidden, layers=args.plm_layers, heads=args.plm_hidden // 64, inter=4 * args.plm_hidden,
               vocab=args.vocab, max_pos=max(512, args.max_len))
    tmpl = model_state_template(data["f_in"], hc, data["c"], plm)
    g = torch.Generator().manual_seed(1)

    def init(k, shape):
        if len(shape) > 1:
            return torch.randn(*shape, generator=g) * 0.02
        norm_gain = k.endswith("mean_scale") or (k.endswith("weight") and any(t in k for t in ("LayerNorm", "layer_norm", "gnorm", "fusion_network.1")))
        return torch.ones(*shape) if norm_gain else torch.zeros(*shape)

    say(f"building the oracle model (hidden_channels={hc}, {threads} threads)")
    sd = {k: init(k, s) for k, s in tmpl.items()}
    plm_sd = {k[len("plm_encoder."):]: v for k, v in sd.items() if k.startswith("plm_encoder.")}
    om = O.OracleGraphTextLM(data["f_in"], hc, data["c"], plm_sd, plm["heads"])
    om.load_reference_state(sd)
    del sd, plm_sd
    mask = data["active"]
    idx = mask.nonzero(as_tuple=True)[0]
    sample = idx[: args.cpu_plm_sample]
    smask = torch.zeros_like(mask)
    smask[sample] = True
    # leg 1: everything except the text encoder, on the full graph (empty text mask => plm_embeds = 0)
    t0 = time.time()
    xm = O.soft_masking_gnn_input(data["x"], mask, om.gnn_mask_token_embed, 0.7)
    logits = om(xm, data["edge_index"], ids, am, torch.zeros_like(mask), plm_batch_size=32)
    say(f"GNN + cross-attention + head forward done ({time.time() - t0:.1f}s)")
    loss = F.cross_entropy(logits[mask], data["y"][mask], label_smoothing=0.2)
    loss.backward()
    t_rest = time.time() - t0
    say(f"... backward done ({t_rest:.1f}s)")
    # leg 2: BERT + pooling on a bounded sample of the active nodes, scaled linearly (row-wise independent)
    t1 = time.time()
    pe = om.encode_texts(ids, am, smask, 32)
    pe.sum().backward()
    t_plm = time.time() - t1
    say(f"BERT leg on {sample.numel()} nodes done ({t_plm:.1f}s)")
    est = t_rest + t_plm * (idx.numel() / max(sample.numel(), 1))
    # the reference's per-edge Python loop for edge typing (main.py:257-267), timed on a bounded sample of edges and
    # scaled linearly: the "faithful" variant of SURVEY.md section 8d
    t2 = time.time()
    ne = min(data["e"], 20000)
    O.edge_types_loop(data["edge_index"][:, :ne], data["n"])
    t_loop = (time.time() - t2) * (data["e"] / ne)
    say(f"per-edge typing loop: {t_loop:.1f}s for all edges (scaled from {ne})")
    return dict(value=round(data["n"] / est, 3), unit="nodes/s", cores=threads, kind="port",
                faithful_value=round(data["n"] / (est + t_loop), 3),
                sample=(f"oracle (CPU fp32 restatement) fwd+bwd on the same {data['n']}-node graph, vectorised edge typing, "
                        f"hidden_channels={hc}{'' if hc == args.hc else ' (bench uses %d)' % args.hc}: GNN + cross-attention + head "
                        f"on the full graph measured ({t_rest:.1f}s); BERT leg measured on {sample.numel()} of {idx.numel()} "
                        f"active nodes ({t_plm:.1f}s) and scaled linearly"),
                seconds_measured=round(t_rest + t_plm, 1))


def _time_events(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return sum(ts) / len(ts), ts[len(ts) // 2]


def micro(dev, args):
    """Kernel micro-benchmarks at sizes where the named roofline is the binding one:
    * aggregation on a power-law graph whose feature matrix is far larger than the 256 MiB Infinity Cache
      (a per-GPU shard of the 10M-node S5 config), F=768;
    * BERT-geometry masked MHA at L=512 (B=32, h=12, d=64) and CrossAttention geometry (h=8, d=96)."""
    import gmlm_amd
    from gmlm_amd import ops
    out = {}
    groups = set(args.micro_select.split(","))
    if "spmm" in groups:
        _micro_spmm(dev, args, out)
    if "attn" in groups:
        _micro_attn(dev, args, out)
    return out


def _micro_spmm(dev, args, out):
    import gmlm_amd
    from gmlm_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    n, e, f = args.micro_nodes, args.micro_edges, 768
    w = (torch.arange(n, device=dev, dtype=torch.float32) + 1.0).pow(-1.0 / 1.2)          # Chung-Lu, alpha = 2.2
    perm = torch.randperm(n, device=dev, generator=g)
    src = perm[torch.multinomial(w, e, replacement=True, generator=g)]
    dst = perm[torch.multinomial(w, e, replacement=True, generator=g)]
    ei = torch.stack([src, dst])
    del w, perm, src, dst
    csr = gmlm_amd.build_rel_csr(ei, n, 5)
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
        x = torch.randn(n, f, device=dev, dtype=dt)
        s_fwd = n * csr.r_active
        a_fwd = ops.spmm_algorithmic_bytes(e, s_fwd, s_fwd, f, x.element_size())
        avg, med = _time_events(lambda: ops.RGCNAggregate.apply(x, csr), 5)
        out[f"spmm_fwd_{name}"] = {"nodes": n, "edges": e, "f": f, "r_active": csr.r_active, "avg_ms": round(avg, 3),
                                   "algorithmic_GB": round(a_fwd / 1e9, 2), "GBps": round(a_fwd / avg / 1e6, 1),
                                   "frac_hbm_peak": round(a_fwd / avg / 1e6 / HBM_PEAK_GBS, 4)}
        gh = torch.randn(n, csr.r_active * f, device=dev, dtype=dt)
        gx = torch.empty(n, f, device=dev, dtype=dt)
        a_bwd = ops.spmm_algorithmic_bytes(e, n, n, f, x.element_size(), True)
        avg, med = _time_events(lambda: ops._spmm(gh.view(n * csr.r_active, f), csr.t_rowptr, csr.t_seg, csr.inv_cnt, False, n, f, gx, csr.t_split), 5)
        out[f"spmm_bwd_{name}"] = {"avg_ms": round(avg, 3), "algorithmic_GB": round(a_bwd / 1e9, 2),
                                   "GBps": round(a_bwd / avg / 1e6, 1), "frac_hbm_peak": round(a_bwd / avg / 1e6 / HBM_PEAK_GBS, 4)}
        del x, gh, gx
    del csr, ei
    torch.cuda.empty_cache()


def _micro_attn(dev, args, out):
    from gmlm_amd import ops
    for tag, b, h, l, d, masked in (("mha_L512", 32, 12, 512, 64, True), ("mha_L128", 256, 12, 128, 64, True),
                                    ("xattn_N5201", 1, 8, 5201, 96, False), ("xattn_N20804", 1, 8, 20804, 96, False)):
        q, k, v = (torch.randn(b, l, h * d, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
        kv_len = torch.randint(l // 2, l + 1, (b,), device=dev, dtype=torch.int32) if masked else None
        go = torch.randn(b, l, h * d, device=dev, dtype=torch.bfloat16)
        fl = 4.0 * b * h * l * l * d
        avg, _ = _time_events(lambda: ops.attention(q.detach(), k.detach(), v.detach(), kv_len, h, d ** -0.5), 5)
        y = ops.attention(q, k, v, kv_len, h, d ** -0.5)
        avg_b, _ = _time_events(lambda: torch.autograd.grad(y, (q, k, v), go, retain_graph=True), 5)
        out[tag] = {"b": b, "h": h, "l": l, "d": d, "fwd_ms": round(avg, 3), "fwd_TFLOPs": round(fl / avg / 1e9, 1),
                    "fwd_frac_mfma_peak": round(fl / avg / 1e9 / MFMA_BF16_PEAK_TF, 4), "bwd_ms": round(avg_b, 3),
                    "bwd_TFLOPs": round(2.5 * fl / avg_b / 1e9, 1),
                    "bwd_frac_mfma_peak": round(2.5 * fl / avg_b / 1e9 / MFMA_BF16_PEAK_TF, 4),
                    "note": "padded flops (masked keys counted)" if masked else "no mask"}
        del q, k, v, go, y


def gnn_large(dev, args):
    """get_graph_embeddings forward+backward on one GPU's share of the 10M-node S5 config (SURVEY.md §8d:
    reported separately because the N x N cross-attention, not the GNN, bounds the full model at that size):
    power-law graph, F_in = 768, hidden_channels = args.hc, bf16, reference dropout, activation checkpointing
    like the reference (main.py:278-314)."""
    import gmlm_amd
    from transformers import BertConfig, BertModel
    n, e, f_in = args.micro_nodes, args.micro_edges, 768
    g = torch.Generator(device=dev).manual_seed(6)
    w = (torch.arange(n, device=dev, dtype=torch.float32) + 1.0).pow(-1.0 / 1.2)
    perm = torch.randperm(n, device=dev, generator=g)
    ei = torch.stack([perm[torch.multinomial(w, e, replacement=True, generator=g)],
                      perm[torch.multinomial(w, e, replacement=True, generator=g)]])
    del w, perm
    x = torch.randn(n, f_in, device=dev)
    mask = torch.rand(n, device=dev) < 0.3
    enc = BertModel(BertConfig(vocab_size=64, hidden_size=args.plm_hidden, num_hidden_layers=1,
                               num_attention_heads=args.plm_hidden // 64, intermediate_size=64, max_position_embeddings=16))
    torch.manual_seed(0)
    m = gmlm_amd.GraphTextLM(f_in, args.hc, 16, dropout_rate=0.3, plm_encoder=enc, compute_dtype=torch.bfloat16,
                             activation_checkpointing=True).to(dev).train()

    def step():
        m.zero_grad(set_to_none=True)
        out = m.get_graph_embeddings(m.soft_mask_input(x, mask, 0.7), ei)
        out.float().square().mean().backward()

    torch.cuda.reset_peak_memory_stats()
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    csr = m.graph(ei, n)
    return {"workload": f"get_graph_embeddings fwd+bwd, power-law graph N={n} E={e} F_in={f_in} hidden_channels={args.hc} bf16, "
                        f"activation checkpointing, R_a={csr.r_active}",
            "nodes_per_s": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 2),
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}


def _pmc_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_spmm_traffic.json); None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_spmm_traffic.json")) as f:
            return json.load(f).get(key, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="squirrel", choices=list(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--hc", type=int, default=768)
    ap.add_argument("--plm-hidden", type=int, default=768)
    ap.add_argument("--plm-layers", type=int, default=12)
    ap.add_argument("--vocab", type=int, default=30522)
    ap.add_argument("--max-len", type=int, default=128)
    ap.add_argument("--plm-batch", type=int, default=4096, help="text micro-batch (nodes); default = all active nodes in ONE packed batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-plm-sample", type=int, default=16)
    ap.add_argument("--cpu-hc", type=int, default=768)
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-micro", action="store_true", help="skip the kernel micro-benchmarks (rank 0, N=1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo = 1-GPU rehearsal of the N>1 path)")
    ap.add_argument("--micro-only", action="store_true")
    ap.add_argument("--gnn-large", action="store_true", help="GNN-only step on a 1.25M-node S5 shard (single GPU)")
    ap.add_argument("--micro-select", default="spmm,attn", help="comma list of micro-benchmark groups: spmm, attn")
    ap.add_argument("--micro-nodes", type=int, default=1_250_000)
    ap.add_argument("--micro-edges", type=int, default=12_500_000)
    args = ap.parse_args()

    import gmlm_amd
    from gmlm_amd import ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)     # gloo rehearsal: several ranks on one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    if args.gnn_large:
        print(json.dumps({"gnn_large": gnn_large(dev, args)}))
        return
    if args.micro_only:
        print(json.dumps({"micro": micro(dev, args)}))
        return
    data = synthetic(args.workload, n_parts=world)
    ids, am = synthetic_tokens(data["n"], args.max_len, args.vocab, seed=data["n"])
    model = build_model(args, data, dev)
    if distributed:
        from gmlm_amd.dist import attach_partition
        part = attach_partition(model, data["edge_index"], data["n"], dev)
        lo, hi = part.plan.lo, part.plan.hi
    else:
        part, lo, hi = None, 0, data["n"]
    x = data["x"][lo:hi].to(dev)
    y = data["y"][lo:hi].to(dev)
    active = data["active"][lo:hi].to(dev)
    ei = data["edge_index"].to(dev)
    tokens = gmlm_amd.TokenizedTexts.from_mask(ids[lo:hi].to(dev), am[lo:hi].to(dev))
    n_active_total = int(data["active"].sum())

    def step():
        model.zero_grad(set_to_none=True)
        xm = model.soft_mask_input(x, active, 0.7)
        logits = model(xm, ei, tokens, active, plm_batch_size=args.plm_batch)
        idx = model.active_index                     # the forward's own active-node index (no second mask -> index sync)
        loss = F.cross_entropy(logits.index_select(0, idx), y.index_select(0, idx), label_smoothing=0.2,
                               reduction="sum") / n_active_total
        loss.backward()
        if part is not None:
            part.all_reduce_grads(model)
        return loss

    def fence():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timer = None if args.no_kernel_timers else ops.KernelTimer()
    ops.TIMER = timer
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    ops.TIMER = None
    if distributed:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = data["n"] * args.steps / dt

    out = {
        "metric": "nodes/sec fwd+bwd (full-batch GraphTextLM step)", "value": round(value, 2), "unit": "nodes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.workload}-size synthetic graph x{world} (N={data['n']}, E={data['e']}, F_in={data['f_in']}), "
                               f"hidden_channels={args.hc}, BERT geometry {args.plm_hidden}x{args.plm_layers}, "
                               f"{n_active_total} active text nodes, 16..{args.max_len} tokens, plm_batch_size={args.plm_batch}",
                   "global_nodes": data["n"], "parallelism": f"1-D node partition x{world}" if distributed else "single GPU",
                   "loss": round(float(loss.detach()), 5)},
    }
    if rank == 0 and timer is not None:
        summ = timer.summary()
        kern = {}
        for name, d in summ.items():
            ms = d["ms"] / max(d["launches"], 1)
            k = {"launches_per_step": d["launches"] / args.steps, "avg_ms": round(ms, 4)}
            if d["bytes"]:
                k["algorithmic_GBps"] = round(d["bytes"] / d["ms"] / 1e6, 1)
                k["frac_hbm_peak"] = round(d["bytes"] / d["ms"] / 1e6 / HBM_PEAK_GBS, 4)
            if d["flops"]:
                peak = MFMA_BF16_PEAK_TF if args.dtype == "bf16" else MFMA_F32_PEAK_TF
                k["TFLOPs"] = round(d["flops"] / d["ms"] / 1e9, 2)
                k["frac_mfma_peak"] = round(d["flops"] / d["ms"] / 1e9 / peak, 4)
            kern[name] = k
        s = summ.get("spmm_fwd")
        if s:
            ach = s["bytes"] / s["ms"] / 1e6
            out["roofline"] = {"kernel": "seg_reduce_vec_kernel (RGCN mean aggregation, forward, 4 layers/step)",
                               "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": _pmc_traffic("spmm_fwd_in_step") if world == 1 else None,
                               "algorithmic_bytes_per_launch": round(s["bytes"] / s["launches"]),
                               "avg_launch_ms": round(s["ms"] / s["launches"], 4)}
        out["kernels"] = kern
    if rank == 0 and world == 1 and not args.no_micro:
        del model
        torch.cuda.empty_cache()
        out["micro"] = mi = micro(dev, args)
        # the HBM roofline binds only when the feature matrix is far larger than the 256 MiB Infinity Cache
        sp = mi.get("spmm_fwd_bf16")
        if sp:
            out["roofline_hbm_regime"] = {
                "kernel": "seg_reduce_vec_kernel (RGCN mean aggregation, forward) on a %d-node / %d-edge power-law shard, F=768 bf16"
                          % (sp["nodes"], sp["edges"]),
                "bound": "hbm", "achieved": sp["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sp["frac_hbm_peak"],
                "traffic": _pmc_traffic("spmm_fwd_bf16"), "algorithmic_bytes_per_launch": int(sp["algorithmic_GB"] * 1e9),
                "avg_launch_ms": sp["avg_ms"]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args, data, ids, am)
        except Exception as exc:  # the baseline is a reported number, never a reason to lose the bench line
            out["cpu_baseline"] = {"value": None, "unit": "nodes/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {exc!r}"}
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
