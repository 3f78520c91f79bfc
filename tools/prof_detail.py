#!/usr/bin/env python
"""Per-kernel totals over the LAST `window_ms` of a rocprofv3 kernel_trace.csv (steady state only).
   python tools/prof_detail.py <dir> <window_ms> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
win = float(sys.argv[2]) * 1e6
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
end = max(r[1] for r in rows)
cat = {}
for s, e, n in rows:
    if s < end - win:
        continue
    a = cat.setdefault(n[:100], [0, 0]); a[0] += e - s; a[1] += 1
tot = sum(v[0] for v in cat.values())
for k, (t, c) in sorted(cat.items(), key=lambda x: -x[1][0])[:top]:
    print(f"{k:102s} {t / 1e3:9.1f} us x{c:5d} avg {t / c / 1e3:7.1f} {100 * t / tot:5.1f}%")
print(f"busy {tot / 1e6:.2f} ms in the last {win / 1e6:.0f} ms")
