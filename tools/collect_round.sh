#!/bin/bash
# Collect the judged measurement artifacts of a round on the GPU box (from the repo root):  bash tools/collect_round.sh r02
# Leaves everything under gpurun_out/<tag>_*; copy into profiles/ afterwards.
TAG=${1:-rXX}
R=$PWD
mkdir -p $R/gpurun_out
# 1. the default bench line (with CPU baseline), and the two secondary workloads
python bench.py > gpurun_out/${TAG}_final_bench_cfg2.json 2> gpurun_out/${TAG}_final_bench_cfg2.err
python bench.py --workload cfg2ref --no-cpu-baseline > gpurun_out/${TAG}_final_bench_cfg2ref.json 2>/dev/null
python bench.py --workload stage3 --no-cpu-baseline > gpurun_out/${TAG}_final_bench_stage3.json 2>/dev/null
# 2. rocprofv3 --kernel-trace --stats of the same command + condensed views
bash tools/prof_run.sh ${TAG}_final --steps 20 --warmup 5 > /dev/null
cd /tmp && export TMPDIR=/tmp
python $R/tools/prof_graph_steps.py /tmp/${TAG}_final "knn_tile_kernel<9, true" 1100 seq > $R/gpurun_out/${TAG}_final_bench_cfg2_graph_steps.txt
# 3. secondary configs, per kernel
for cfg in cfg3 cfg5; do
  rm -rf /tmp/p_$cfg
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$cfg -o run -- python $R/tools/run_configs.py $cfg > $R/gpurun_out/${TAG}_final_${cfg}.log 2>&1
  python $R/tools/prof_detail.py /tmp/p_$cfg 30 45 > $R/gpurun_out/${TAG}_final_${cfg}_forward_per_kernel.txt
done
rm -rf /tmp/p_t4
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_t4 -o run -- python $R/tools/train_step.py --batch 32 --steps 3 --warmup 2 > $R/gpurun_out/${TAG}_final_cfg4_train_step.log 2>&1
python $R/tools/prof_detail.py /tmp/p_t4 100 60 > $R/gpurun_out/${TAG}_final_cfg4_train_step_per_kernel.txt
cd $R
# 4. un-profiled secondary numbers
python tools/run_configs.py cfg3 2>&1 | tail -2 > gpurun_out/${TAG}_final_cfg3_unprofiled.txt
python tools/run_configs.py cfg5 2>&1 | tail -2 > gpurun_out/${TAG}_final_cfg5_unprofiled.txt
python tools/train_step.py --batch 32 --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${TAG}_final_cfg4_unprofiled.txt
# 5. hardware counters (separate passes)
bash tools/pmc_refresh.sh ${TAG} > /dev/null 2>&1
# 6. per-shape GEMM table
python tools/bench_x6.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_x6_gemm_shapes.txt
python tools/bench_x6.py cfg4 2>&1 | grep -v amdgpu.ids >> gpurun_out/${TAG}_x6_gemm_shapes.txt
./tools/ubench/mfma_bf16_fill > gpurun_out/${TAG}_ubench_mfma_bf16_fill.txt 2>&1
cut -c1-300 gpurun_out/${TAG}_final_bench_cfg2.json
