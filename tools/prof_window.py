#!/usr/bin/env python
"""Aggregate a rocprofv3 kernel_trace.csv over its LAST `window_ms` milliseconds (steady state, excludes warm-up /
MIOpen find).   python tools/prof_window.py <dir> <window_ms>"""
import csv, glob, sys
path, win = sys.argv[1], float(sys.argv[2]) * 1e6
f = glob.glob(path + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
end = max(r[1] for r in rows)
cat = {}
for s, e, n in rows:
    if s < end - win:
        continue
    if n.startswith("Cijk"): k = "GEMM (rocBLAS/hipBLASLt)"
    elif "gkg::" in n: k = "gkg:" + n.split("gkg::")[1].split("(")[0].split("<")[0]
    elif "BatchNorm" in n: k = "MIOpen BN"
    elif "batched_transpose" in n or "SubTensor" in n or "Im2d2Col" in n or "Col2Im" in n: k = "MIOpen transpose/im2col"
    elif "miopen" in n.lower() or "igemm" in n or "ck::" in n or "_ZN2ck" in n or "Conv" in n or "gemm" in n.lower(): k = "MIOpen conv: " + n[:48]
    elif "at::native" in n: k = "torch elementwise/reduce"
    else: k = "other:" + n[:40]
    a = cat.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
tot = sum(v[0] for v in cat.values())
for k, (t, c) in sorted(cat.items(), key=lambda x: -x[1][0])[:30]:
    print(f"{k:64s} {t / 1e6:8.2f} ms  x{c:5d}  {100 * t / tot:5.1f}%")
print(f"busy {tot / 1e6:.1f} ms of window {win / 1e6:.0f} ms")
