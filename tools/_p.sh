python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r04_final_tests.txt; cat gpurun_out/r04_final_tests.txt
python bench.py --workload cfg5 > gpurun_out/r04_after_knn_map_bench_cfg5.json 2>/dev/null
python bench.py --workload cfg3 > gpurun_out/r04_after_knn_map_bench_cfg3.json 2>/dev/null
python bench.py --workload stage1 --no-cpu-baseline > gpurun_out/r04_after_knn_map_bench_stage1.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r04_after_knn_map_bench_cfg2.json 2>/dev/null
for w in cfg2 stage1 cfg3 cfg5; do python -c "
import json; d=json.loads(open('gpurun_out/r04_after_knn_map_bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d.get('ms_per_step_no_tune'), d.get('ms_per_step_knn_bf16'), d['roofline']['frac'])"; done
