"""Kernel sequence of the last timed step in a rocprofv3 kernel-trace db: index, duration, gap before, name.
usage: python tools/step_sequence.py <results.db> [first] [last]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'softmask_fwd_kernel' in r[0]]
sel = rows[idx[-1]:]
a = int(sys.argv[2]) if len(sys.argv) > 2 else 0
b = int(sys.argv[3]) if len(sys.argv) > 3 else len(sel)
for i in range(a, min(b, len(sel))):
    gap = (sel[i][1] - sel[i - 1][2]) / 1e3 if i else 0.0
    print(f"{i:4d} {(sel[i][2]-sel[i][1])/1e3:8.1f} us  gap {gap:7.1f}  {sel[i][0][:100]}")
