#!/bin/bash
# per-launch durations of the k-NN kernels inside the cfg4 train step, prefilter rule on / off
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for mode in auto 0; do
rm -rf /tmp/p4$mode
if [ $mode = auto ]; then unset GKG_KNN_PREFILTER; else export GKG_KNN_PREFILTER=0; fi
rocprofv3 --kernel-trace --output-format csv -d /tmp/p4$mode -o run -- python $R/tools/train_step.py --batch 32 --steps 3 --warmup 2 2>&1 | tail -1 | cut -c1-200
echo "== prefilter=$mode"
python $R/tools/prof_kernel_launches.py /tmp/p4$mode 105 knn_ | sort -k4 | awk '{print $1, $4, $6, $7}' | head -40
done
