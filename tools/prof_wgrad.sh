#!/bin/bash
# per-launch durations of the weight-gradient kernels inside the cfg2 step: x6all (own streaming kernel) vs the default dispatch
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for mode in x6all x6; do
  export GKG_GEMM_MATH=$mode
  rm -rf $ROOT/gpurun_out/wg_$mode
  rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/wg_$mode -o t -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tune > $ROOT/gpurun_out/wg_$mode.json 2>/dev/null
  tail -1 $ROOT/gpurun_out/wg_$mode.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$mode', j['ms_per_step'])"
  python - <<PY
import csv, glob, collections
f = glob.glob("$ROOT/gpurun_out/wg_$mode/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if "wgrad" in k or "Cijk" in k or "gemm" in k:
        tot[(k[:60], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(tot.items()):
    v.sort()
    print("%4d launches  median %7.1f us  min %7.1f  grid %s x %s  %s" % (len(v), v[len(v)//2], v[0], k[1], k[2], k[0]))
PY
  rm -rf $ROOT/gpurun_out/wg_$mode
done
