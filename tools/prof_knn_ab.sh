#!/bin/bash
# A/B of the k-NN kernels (prefilter + exact re-rank vs fp32 tile kernel) under rocprofv3 --kernel-trace --stats.
#   bash tools/prof_knn_ab.sh [bench args...]     e.g. --workload stage3
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
rm -rf /tmp/pk$mode
GKG_KNN_PREFILTER=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk$mode -o run -- python $R/bench.py --steps 20 --warmup 5 --no-tune --no-cpu-baseline "$@" > /tmp/pk$mode.log 2>&1
echo "== prefilter=$mode  $(grep -o '"ms_per_step": [0-9.]*' /tmp/pk$mode.log | head -1)"
python - <<PY
import csv,glob
f=glob.glob("/tmp/pk$mode/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "knn" in n or "token_prep" in n: print("  ", n[:72], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
