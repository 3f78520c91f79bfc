#!/usr/bin/env python
"""Print the gkg:: rows (name, calls, avg us) of a rocprofv3 kernel_stats.csv found under a directory."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gkg::" in r["Name"]:
            print(f"  {r['Name'][:58]:58s} x{r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
