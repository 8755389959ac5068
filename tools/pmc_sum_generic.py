#!/usr/bin/env python3
"""Average every counter per kernel over the dispatches of a rocprofv3 --pmc run (csv output).  usage: pmc_sum_generic.py <dir> [kernel substring]"""
import collections, csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void gmlm::", "")
        if flt and flt not in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    print(k)
    for name, v in sorted(c.items()):
        print(f"    {name:44s} {sum(v) / len(v):16.1f}   (n={len(v)})")
