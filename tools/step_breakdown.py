"""Per-step kernel breakdown from a rocprofv3 --kernel-trace run of `bench.py --no-micro --no-cpu-baseline --no-fp32-leg`
(CSV kernel trace, or the rocpd database of older runs).
usage: python tools/step_breakdown.py <kernel_trace.csv | results.db> [n_timed_steps] [top]"""
import collections, csv, sqlite3, sys
path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if path.endswith(".csv"):
    rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"))
            for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: r[1])
else:
    rows = list(sqlite3.connect(path).execute("select name, start, end, vgpr_count, lds_size from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'softmask_fwd_kernel' in r[0]]
sel = rows[idx[-steps]:]
t0, t1 = sel[0][1], max(r[2] for r in sel)
busy = sum(r[2] - r[1] for r in sel)
print(f"span {(t1 - t0) / steps / 1e6:.3f} ms/step, kernel-busy {busy / steps / 1e6:.3f} ms/step")
agg = collections.defaultdict(lambda: [0, 0, None])
for r in sel:
    k = r[0][:110]
    agg[k][0] += 1; agg[k][1] += r[2] - r[1]; agg[k][2] = r[3:]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{v[1] / steps / 1e6:7.3f} ms/step {v[0] / steps:6.1f} calls {v[1] / v[0] / 1e3:8.1f} us vgpr{v[2][0]} lds{v[2][1]} | {k}")
