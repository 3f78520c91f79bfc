#!/bin/bash
# Weight-gradient kernels inside the cfg2 step, in launch order of one step: the own streaming kernel (GKG_GEMM_MATH=x6all)
# against the vendor kernels of the default dispatch, library-default selection (--no-tune) and TunableOp-selected.
#   bash tools/prof_wgrad.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for mode in "x6all --no-tune" "x6 --no-tune" "x6 "; do
  export GKG_GEMM_MATH=${mode%% *}
  rm -rf /tmp/wg_prof
  rocprofv3 --kernel-trace --output-format csv -d /tmp/wg_prof -o t -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline ${mode#* } > /tmp/wg_prof.json 2>/dev/null
  tail -1 /tmp/wg_prof.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('== $mode: ms_per_step', j['ms_per_step'], 'no_tune', j.get('ms_per_step_no_tune'))"
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/wg_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the Grapher's k-NN launch; take the graph-replayed steps (short period) and report per-position medians
marks = [i for i, r in enumerate(rows) if "knn_tile_kernel<9, true" in r["Kernel_Name"]]
steps = []
for a, b in zip(marks, marks[1:]):
    per = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    if per < 1300:
        steps.append(rows[a:b])
steps = steps[-15:]
def is_wg(n): return "wgrad" in n or "Cijk_Ailk_Bjlk" in n or "Cijk_Ailk_Bljk_SB_MT128" in n or "reduce_kernel" in n
seqs = [[(r["Kernel_Name"][:58], r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in s if is_wg(r["Kernel_Name"])] for s in steps]
n = min(len(q) for q in seqs) if seqs else 0
tot = 0.0
for i in range(n):
    d = sorted(q[i][2] for q in seqs)
    tot += d[len(d) // 2]
    print("  %7.1f us  grid %8s  %s" % (d[len(d) // 2], seqs[-1][i][1], seqs[-1][i][0]))
print("  total %.1f us over %d launches (%d replayed steps)" % (tot, n, len(seqs)))
PY
done
