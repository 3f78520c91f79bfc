#!/bin/bash
# Collect the judged measurement artifacts of a round on the GPU box (from the repo root):  bash tools/collect_round.sh r04
# Leaves everything under gpurun_out/<tag>_*; copy into profiles/ afterwards.
TAG=${1:-rXX}
R=$PWD
mkdir -p $R/gpurun_out
# 0. hardware counters first (separate passes): the default bench line quotes profiles/<tag>_pmc.json and checks that it was
#    taken on the kernel sources it runs (csrc hash) — so the box's copy of profiles/ gets this collection's file before step 1
bash tools/pmc_refresh.sh ${TAG} > /dev/null 2>&1
[ -s gpurun_out/${TAG}_pmc.json ] && cp gpurun_out/${TAG}_pmc.json profiles/${TAG}_pmc.json
# 1. the default bench line (with CPU baseline), and every other workload through the same entry point
python bench.py > gpurun_out/${TAG}_final_bench_cfg2.json 2> gpurun_out/${TAG}_final_bench_cfg2.err
for w in cfg2ref stage3 stage1; do
  python bench.py --workload $w --no-cpu-baseline > gpurun_out/${TAG}_final_bench_$w.json 2>/dev/null
done
for w in cfg3 cfg5 cfg4; do
  python bench.py --workload $w > gpurun_out/${TAG}_final_bench_$w.json 2>/dev/null
done
# 2. rocprofv3 --kernel-trace --stats of the default command + condensed views (both timed legs; then the library-default leg alone)
bash tools/prof_run.sh ${TAG}_final --steps 20 --warmup 5 > /dev/null
( cd /tmp && python $R/tools/prof_graph_steps.py /tmp/${TAG}_final "knn_tile_kernel<9, true" 1100 seq > $R/gpurun_out/${TAG}_final_bench_cfg2_graph_steps.txt )
bash tools/prof_run.sh ${TAG}_notune --steps 20 --warmup 5 --no-tune > /dev/null
( cd /tmp && python $R/tools/prof_graph_steps.py /tmp/${TAG}_notune "knn_tile_kernel<9, true" 1200 seq > $R/gpurun_out/${TAG}_final_bench_cfg2_graph_steps_library_default_leg.txt )
# 3. whole-backbone configs, per kernel
cd /tmp && export TMPDIR=/tmp
for cfg in cfg3 cfg5 cfg4; do
  rm -rf /tmp/p_$cfg
  knn=""; [ $cfg != cfg4 ] && knn="--knn exact"          # the forward workloads: the default (index-exact) leg only, eager launches
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$cfg -o run -- python $R/bench.py --workload $cfg --no-cpu-baseline --steps 3 --warmup 2 --no-graph $knn > $R/gpurun_out/${TAG}_final_${cfg}_prof.log 2>&1
  win=60; [ $cfg = cfg4 ] && win=95; [ $cfg = cfg5 ] && win=110
  python $R/tools/prof_detail.py /tmp/p_$cfg $win 60 > $R/gpurun_out/${TAG}_final_${cfg}_per_kernel.txt
done
cd $R
# 4. SQ counters of the k-NN kernels (separate passes)
bash tools/pmc_knn_run.sh > gpurun_out/${TAG}_knn_tile_sq_counters.txt 2>&1
# 5. per-shape tables
python tools/bench_x6.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_x6_gemm_shapes.txt
python tools/bench_mr_bwd.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_bench_mr_bwd.txt
python tools/bench_mr_fwd.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_bench_mr_fwd.txt
python tools/bench_knn_model_inputs.py cfg3 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_knn_kernels_on_model_inputs_cfg3.txt
python tools/bench_stem.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_bench_stem.txt
python tools/ubench/hbm_rate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_hbm_rate.txt
cut -c1-300 gpurun_out/${TAG}_final_bench_cfg2.json
