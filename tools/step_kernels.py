"""Per-kernel totals of the LAST timed step in a rocprofv3 kernel-trace database (rocpd .db).
usage: python tools/step_kernels.py <results.db> [top_n]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'softmask_fwd_kernel' in r[0]]
sel = rows[idx[-1]:]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in sel:
    a = agg[n]
    a[0] += 1
    a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"last step: {len(sel)} kernels, {tot / 1e3:.3f} ms of kernel time")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{c:5d} {t:9.1f} us {t / c:8.1f} us/call  {n[:120]}")
