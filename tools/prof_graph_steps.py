#!/usr/bin/env python
"""Steady-state steps of a rocprofv3 kernel trace of bench.py, split at a once-per-step marker kernel:
   python tools/prof_graph_steps.py <dir> [marker-substring] [max_period_us]
Steps whose period is below max_period_us (default 1.3x the shortest) are the hipGraph replays; prints their mean
period, busy time, idle time, and per-kernel mean durations / launches per step."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "knn_tile_kernel<9, true"
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
marks = [i for i, r in enumerate(rows) if marker in r[2]]
periods = [(rows[marks[j + 1]][0] - rows[marks[j]][0]) / 1e3 for j in range(len(marks) - 1)]
lim = float(sys.argv[3]) if len(sys.argv) > 3 else 1.3 * min(periods)
sel = [j for j, p in enumerate(periods) if p <= lim]
cat = {}
busy = 0.0
for j in sel:
    for s, e, n in rows[marks[j]:marks[j + 1]]:
        a = cat.setdefault(n[:90], [0.0, 0]); a[0] += (e - s) / 1e3; a[1] += 1
        busy += (e - s) / 1e3
n = len(sel)
mean_p = sum(periods[j] for j in sel) / n
print(f"{n} steps with period <= {lim:.0f} us: mean period {mean_p:.1f} us, busy {busy / n:.1f} us, idle {mean_p - busy / n:.1f} us, "
      f"{sum(v[1] for v in cat.values()) / n:.1f} launches/step")
for k, (t, c) in sorted(cat.items(), key=lambda x: -x[1][0])[:70]:
    print(f"{k:92s} {t / n:8.1f} us/step x{c / n:5.1f} avg {t / c:7.1f}")
if len(sys.argv) > 4:                      # launch sequence of the median selected step
    j = sel[len(sel) // 2]
    print("--- launch sequence of one step (duration us, grid) ---")
    import csv as _csv
    full = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""))
                   for r in _csv.DictReader(open(f))))
    for s, e, n, gsz, l in full[marks[j]:marks[j + 1]]:
        print(f"{(e - s) / 1e3:8.1f}  {gsz:>9s}  {l:>6s}  {n[:110]}")
