#!/bin/bash
# Same-box A/B of one GKG_DISABLE switch over a bench workload:  bash tools/debug/ab_disable.sh prep_fork cfg2 [bench args]
R=${GRAFT_REPO_ROOT:-$PWD}
SW=$1; W=$2; shift; shift
for rep in 1 2 3; do
  for dis in "" $SW; do
    GKG_DISABLE=$dis python $R/bench.py --workload $W --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$W', 'GKG_DISABLE=[$dis]', d['ms_per_step'], d.get('ms_per_step_no_tune'))"
  done
done
