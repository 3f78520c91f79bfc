#!/bin/bash
# A/B on the same box: bench line + per-kernel totals of the last replayed steps.  Each side: "<repo subdir or .>|<ENV=val>"
#   bash tools/prof_step_ab.sh ".|GKG_DISABLE=none" "_head|GKG_DISABLE=none"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
 for side in "$@"; do
  R=$ROOT/${side%%|*}; E=${side##*|}
  env $E python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$side', j['ms_per_step'], j['ms_per_step_no_tune'])"
 done
done
n=0
for side in "$@"; do
  R=$ROOT/${side%%|*}; E=${side##*|}; n=$((n+1))
  export $E
  rm -rf $ROOT/gpurun_out/ab_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ab_$n -o t -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-tune > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$ROOT/gpurun_out/ab_$n/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
end = max(int(r["End_Timestamp"]) for r in rows)
win = 20e6
sel = [r for r in rows if int(r["Start_Timestamp"]) > end - win]
span = (max(int(r["End_Timestamp"]) for r in sel) - min(int(r["Start_Timestamp"]) for r in sel)) / 1e3
tot = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r["Kernel_Name"][:70]
    tot[k][0] += 1; tot[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
busy = sum(v[1] for v in tot.values())
print("== $side: window %.0f us, busy %.0f us, launches %d" % (span, busy, len(sel)))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:50]:
    print("%8.1f us %5d  %s" % (v[1], v[0], k))
PY
  unset ${E%%=*}
  rm -rf $ROOT/gpurun_out/ab_$n
done
