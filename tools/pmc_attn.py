#!/usr/bin/env python3
"""Summarise a rocprofv3 PMC pass over the attention micro-benchmarks into profiles/rNN_attn_pmc.{csv,json}.
usage: python tools/pmc_attn.py <rocprof output dir> <out prefix>        (writes <prefix>.csv and <prefix>.json)
Collected with (counters only, no other trace domain, as the pool requires):
    rocprofv3 -i tools/ubench/pmc_attn_in.txt --kernel-trace -d <dir> -o pmc --output-format csv -- tools/ubench/attn_bench 1 1
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)   (GUI_ACTIVE is summed over the 8 XCDs;
            BUSY counts 32 cycles per v_mfma_f32_32x32x16_bf16: guide, cycle constants); averages per dispatch.
For attn_fwd_pipe_kernel one MFMA in 1 + D/16 + 2*D/32 carries the reference max (no useful flops): mfma_util_useful
scales it out."""
import collections, csv, glob, json, sys
d, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_" not in k:
            continue
        k = k.split("(")[0].replace("void gmlm::", "") + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
        "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
res = {}
with open(out + ".csv", "w") as fo:
    fo.write("# rocprofv3 -i tools/ubench/pmc_attn_in.txt --kernel-trace -- tools/ubench/attn_bench 1 1   (two counter passes)\n")
    fo.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs); averages per dispatch; wait_* / active = fraction of SQ_WAVE_CYCLES\n")
    w = csv.writer(fo)
    w.writerow(["kernel", "dispatches"] + cols + ["mfma_util", "mfma_util_useful", "valu_per_mfma", "wait_any", "wait_inst", "active"])
    for k, c in sorted(agg.items()):
        m = {x: (sum(c[x]) / len(c[x]) if c.get(x) else 0.0) for x in cols}
        n = len(c.get("GRBM_GUI_ACTIVE", []))
        util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 256 * 4) if m["GRBM_GUI_ACTIVE"] else 0
        useful = util
        if "attn_fwd_pipe_kernel<96" in k:
            useful = util * 12 / 13
        elif "attn_fwd_pipe_kernel<64" in k:
            useful = util * 8 / 9
        wc = max(m["SQ_WAVE_CYCLES"], 1)
        row = dict(mfma_util=round(util, 3), mfma_util_useful=round(useful, 3), valu_per_mfma=round(m["SQ_INSTS_VALU"] / max(m["SQ_INSTS_MFMA"], 1), 1),
                   wait_any=round(m["SQ_WAIT_ANY"] / wc, 3), wait_inst=round(m["SQ_WAIT_INST_ANY"] / wc, 3), active=round(m["SQ_ACTIVE_INST_ANY"] / wc, 3),
                   lds_conflict_frac=round(m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1), 3), dispatches=n)
        res[k] = row
        w.writerow([k, n] + [round(m[x]) for x in cols] + [row["mfma_util"], row["mfma_util_useful"], row["valu_per_mfma"], row["wait_any"], row["wait_inst"], row["active"]])
json.dump(res, open(out + ".json", "w"), indent=1)
print(open(out + ".csv").read())
