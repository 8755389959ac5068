#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (counter_collection.csv files under a directory): mean counter value per kernel name."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void gmlm::", "").split("(")[0]
        agg[k + " grid=" + r.get("Grid_Size", "?") + " vgpr=" + r.get("VGPR_Count", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    m = {n: sum(v) / len(v) for n, v in c.items()}
    for n in sorted(m):
        print(f"   {n:28s} {m[n]:16.0f}  (n={len(c[n])})")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        print(f"   mfma_util = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}   valu/mfma = {m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_INSTS_MFMA', 1), 1):.2f}")
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        print("   of wave cycles: " + "  ".join(f"{n[3:]}={m[n] / w:.3f}" for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") if n in m))
