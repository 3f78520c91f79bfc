#!/usr/bin/env python
"""Average the per-dispatch PMC values of rocprofv3's counter_collection.csv for kernels whose name contains a
substring.   python tools/pmc_knn.py <dir> <substring>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
sub = sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if sub not in name:
        continue
    key = name.split("gkg::")[-1][:40]
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[len(v) // 2:]                      # steady state: second half of the dispatches
        print(f"   {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")
