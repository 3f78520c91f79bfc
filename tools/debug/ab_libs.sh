#!/bin/bash
# Same-box A/B of two builds of the library (gkgnet_amd/libgkg_hip_old.so built from the previous sources, see tools/debug/README.md):
#   bash tools/debug/ab_libs.sh cfg3 [bench args]
R=${GRAFT_REPO_ROOT:-$PWD}
W=$1; shift
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export GKG_HIP_LIB=$R/gkgnet_amd/libgkg_hip_old.so; else unset GKG_HIP_LIB; fi
    python $R/bench.py --workload $W --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$W', '$lib', d['ms_per_step'], {k: v.get('us_per_step') for k, v in d.get('hip_kernels', {}).items() if k in ('token_prep', 'knn_tile')})"
  done
done
