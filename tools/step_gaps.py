"""Idle gaps between consecutive kernels in the timed steps of a rocprofv3 kernel-trace database.
usage: python tools/step_gaps.py <results.db> [n_steps] [min_gap_us]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'softmask_fwd_kernel' in r[0]]
sel = rows[idx[-1]:]            # last step only
tot = 0.0
prev_end = sel[0][2]
for i in range(1, len(sel)):
    gap = (sel[i][1] - prev_end) / 1e3
    if gap > thr:
        print(f"gap {gap:8.1f} us after #{i-1} {sel[i-1][0][:70]}  -> before {sel[i][0][:60]}")
    if gap > 0:
        tot += gap
    prev_end = max(prev_end, sel[i][2])
print(f"last step: {len(sel)} kernels, total idle {tot/1e3:.2f} ms, span {(prev_end - sel[0][1])/1e6:.2f} ms")
