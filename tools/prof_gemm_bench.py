#!/usr/bin/env python
"""Kernel durations of tools/ubench/gemm_bench under rocprofv3 (rocpd sqlite): median per distinct (kernel, grid) in
first-seen order.  python tools/prof_gemm_bench.py <results.db>"""
import sqlite3, sys, statistics
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start"))
seen, order = {}, []
for n, s, e, gx, gy, gz in rows:
    key = (n.split("(")[0].replace("void gkg::", ""), gx, gy, gz)
    if key not in seen:
        seen[key] = []
        order.append(key)
    seen[key].append((e - s) / 1e3)
for k in order:
    v = seen[k]
    if "fill" in k[0]:
        continue
    print(f"{statistics.median(v):8.1f} us (min {min(v):7.1f}, n={len(v):3d})  grid {k[1]:>8}  {k[0][:110]}")
