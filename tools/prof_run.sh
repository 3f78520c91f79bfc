#!/bin/bash
# Profile the default bench command with rocprofv3 and leave compact summaries under gpurun_out/<tag>/ :
#   kernel_stats.csv (whole run), window.txt (per-kernel totals, last 20 ms), step_seq.txt (last 140 launches in order)
# usage (on the GPU box, from the repo root):  bash tools/prof_run.sh <tag> [bench args...]
R=$PWD; TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG -o run -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/bench.log 2>&1
find /tmp/$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$TAG/kernel_stats.csv \;
python $R/tools/prof_detail.py /tmp/$TAG 20 70 > $R/gpurun_out/$TAG/window.txt
python $R/tools/prof_step.py /tmp/$TAG 140 > $R/gpurun_out/$TAG/step_seq.txt
grep "^{" $R/gpurun_out/$TAG/bench.log | cut -c1-220
