#!/usr/bin/env python
"""Every launch of the kernels whose name contains `pattern` in the LAST window_ms of a rocprofv3 kernel trace: duration, grid.
   python tools/prof_kernel_launches.py <dir> <window_ms> <pattern>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
win = float(sys.argv[2]) * 1e6
rows = [r for r in csv.DictReader(open(f))]
end = max(int(r["End_Timestamp"]) for r in rows)
for r in rows:
    if int(r["Start_Timestamp"]) > end - win and sys.argv[3] in r["Kernel_Name"]:
        print(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?")):>10}  {r["Kernel_Name"][:80]}')
