"""Per-queue run segments of the LAST timed step of a rocprofv3 kernel-trace database: which hardware queue ran what, when -
shows whether the branches of a hipGraph overlapped.  usage: python tools/queue_runs.py <results.db> [min_kernels_per_run]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
minrun = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'softmask_fwd_kernel' in r[0]]
sel = rows[idx[-1]:]
t0 = sel[0][1]
runs = []
for i, (n, s, e, q) in enumerate(sel):
    if runs and runs[-1][0] == q:
        runs[-1][2] = max(runs[-1][2], e); runs[-1][3] += 1; runs[-1][4] += e - s
    else:
        runs.append([q, s, e, 1, e - s, i])
for q, s, e, c, d, i in runs:
    if c >= minrun:
        print(f"queue {q} kernels #{i:3d}..{i + c - 1:3d} from {(s - t0) / 1e3:8.1f} to {(e - t0) / 1e3:8.1f} us, busy {d / 1e3:7.1f} us")
ev = []
for n, s, e, q in sel:
    ev += [(s, 1), (e, -1)]
ev.sort()
cur, last, by = 0, ev[0][0], {}
for t, d in ev:
    by[cur] = by.get(cur, 0) + (t - last); last = t; cur += d
print("time by number of kernels in flight (us):", {k: round(v / 1e3, 1) for k, v in sorted(by.items())},
      "| span", round((max(r[2] for r in sel) - t0) / 1e6, 3), "ms, sum of kernel durations", round(sum(r[2] - r[1] for r in sel) / 1e6, 3), "ms")
