#!/usr/bin/env python3
"""Summarise a rocprofv3 PMC pass over `bench.py --micro-only --micro-select attn` into profiles/r01_attn_pmc.csv.
usage: python tools/pmc_attn.py <rocprof output dir> <out.csv>
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs); averages per dispatch."""
import collections, csv, glob, sys
d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "attn_" not in k:
        continue
    k = k.split("(")[0].replace("void gmlm::", "")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES"]
with open(out, "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- python3 bench.py --micro-only --micro-select attn\n")
    fo.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs)  (GUI_ACTIVE is summed over the 8 XCDs); averages per dispatch\n")
    w = csv.writer(fo)
    w.writerow(["kernel", "dispatches"] + cols + ["mfma_util", "valu_per_mfma"])
    for k, c in agg.items():
        n = len(c["GRBM_GUI_ACTIVE"])
        m = {x: sum(c[x]) / max(len(c[x]), 1) for x in cols}
        util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 256 * 4) if m["GRBM_GUI_ACTIVE"] else 0
        vpm = m["SQ_INSTS_VALU"] / max(m["SQ_INSTS_MFMA"], 1)
        w.writerow([k, n] + [round(m[x]) for x in cols] + [round(util, 3), round(vpm, 1)])
print(open(out).read())
