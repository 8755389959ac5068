#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide prescribes; collected by
tools/collect_traffic_pmc.sh) of `bench.py --micro-only --micro-select spmm` - and, optionally, of the bench step itself -
into profiles/rNN_spmm_traffic.json.
usage: python tools/pmc_traffic.py <fetch dir> <write dir> <out.json> [<in-step fetch dir> <in-step write dir>]

Units / corrections (MI355X_MICROARCH.md §HBM): counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import sys

fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "seg_reduce" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].replace("void gmlm::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
groups = {"spmm_fwd_bf16": ("unsigned short", "false"), "spmm_bwd_bf16": ("unsigned short", "true"),
          "spmm_fwd_f32": ("float", "false"), "spmm_bwd_f32": ("float", "true")}
res = {"_note": "per launch of the aggregation (main + long-segment chunk + combine kernels); hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
                "(gfx950 FETCH_SIZE half-count correction); 1.25M-node / 12.5M-edge power-law shard, F=768"}
for g, (ty, ew) in groups.items():
    ks = [k for k in fe if f"<{ty}," in k and f", {ew}," in k]
    comb = [k for k in fe if "combine" in k and f"<{ty}>" in k]
    fetch_kib = sum(fe[k] for k in ks) + sum(fe[k] for k in comb) / 2   # combine runs for fwd and bwd: half each
    write_kib = sum(wr.get(k, 0) for k in ks) + sum(wr.get(k, 0) for k in comb) / 2
    res[g] = {"kernels": ks + comb, "FETCH_SIZE_KiB": round(fetch_kib), "WRITE_SIZE_KiB": round(write_kib),
              "hbm_bytes_per_launch": int((2 * fetch_kib + write_kib) * 1024)}
if len(sys.argv) > 5:
    def totals(d, counter):                              # (sum, launches) over the forward bf16 aggregation launches
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if "seg_reduce_vec_kernel<unsigned short," in r["Kernel_Name"] and ", false," in r["Kernel_Name"] and r["Counter_Name"] == counter]
        return sum(vals), len(vals)
    (fs, fn), (ws, wn) = totals(sys.argv[4], "FETCH_SIZE"), totals(sys.argv[5], "WRITE_SIZE")
    ks = sorted({r["Kernel_Name"].split("(")[0].replace("void gmlm::", "") for r in csv.DictReader(open(glob.glob(sys.argv[4] + "/**/*counter_collection.csv", recursive=True)[0]))
                 if "seg_reduce_vec_kernel<unsigned short," in r["Kernel_Name"] and ", false," in r["Kernel_Name"]})
    fetch_kib, write_kib = fs / max(fn, 1), ws / max(wn, 1)   # per launch (4 launches per step: one per RGCN layer)
    res["spmm_fwd_in_step"] = {"kernels": ks, "FETCH_SIZE_KiB": round(fetch_kib), "WRITE_SIZE_KiB": round(write_kib),
                               "hbm_bytes_per_launch": int((2 * fetch_kib + write_kib) * 1024),
                               "note": "average over the forward aggregation kernel variants of a bench.py step (Squirrel-size graph, X cache resident: "
                                       "traffic << algorithmic bytes)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
