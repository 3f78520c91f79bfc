#!/bin/bash
# Same-box A/B of the cfg2 step at hipGraph-replay level.  Each side "<tag>|<ENV=val> [ENV=val ...]" :
#   bash tools/replay_ab.sh "base|GKG_DISABLE=wgrad_batch" "x6all|GKG_GEMM_MATH=x6all"
# per side: two bench lines (ms_per_step, ms_per_step_no_tune), then a rocprofv3 kernel trace of the library-default leg
# condensed by tools/prof_graph_steps.py (per-kernel us/step + the launch sequence of one replayed step) into
# gpurun_out/replay_<tag>.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
EXTRA=${BENCH_ARGS:-}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
 for side in "$@"; do
  T=${side%%|*}; E=${side#*|}
  env $E python $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T', j['ms_per_step'], j['ms_per_step_no_tune'])"
 done
done
for side in "$@"; do
  T=${side%%|*}; E=${side#*|}
  rm -rf /tmp/rp_$T
  env $E rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_$T -o t -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-tune $EXTRA > /dev/null 2>&1
  python $ROOT/tools/prof_graph_steps.py /tmp/rp_$T "${MARKER:-knn_tile_kernel<9, true}" ${MAXP:-1300} seq > $ROOT/gpurun_out/replay_$T.txt 2>&1
  head -1 $ROOT/gpurun_out/replay_$T.txt
done
