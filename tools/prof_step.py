#!/usr/bin/env python
"""Print the last `n` kernels of a rocprofv3 kernel_trace.csv in launch order with duration and gap to the previous
kernel.   python tools/prof_step.py <dir> <n>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2])
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"))
              for r in csv.DictReader(open(f)))[-n:]
prev = rows[0][0]
for s, e, name, gx, wx in rows:
    short = name.split("gkg::")[1][:50] if "gkg::" in name else name[:50]
    print(f"{(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:7.1f}  grid {gx:>9s}/{wx:<4s} {short}")
    prev = e
