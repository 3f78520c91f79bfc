#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a per-step table."""
import csv, glob, sys
path, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(path + "/**/*_kernel_stats.csv", recursive=True)[0] if not path.endswith(".csv") else path
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time / step: {tot / steps / 1e3:.1f} us ; kernels / step: {sum(int(r['Calls']) for r in rows) / steps:.1f}")
mine = sum(float(r["TotalDurationNs"]) for r in rows if "gkg::" in r["Name"])
print(f"gkg:: kernels / step: {mine / steps / 1e3:.1f} us")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:78]:78s} x{int(r['Calls']) / steps:5.1f} avg {float(r['AverageNs']) / 1e3:8.1f}us {float(r['Percentage']):5.2f}%")
