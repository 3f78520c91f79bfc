mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05c_tests.txt 2>&1; tail -6 gpurun_out/r05c_tests.txt
for w in cfg5 cfg3; do
  for rep in 1 2; do
    for dis in "" knn_compact; do
      GKG_DISABLE=$dis python bench.py --workload $w --no-cpu-baseline --knn exact --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$w', 'disable=[$dis]', d['ms_per_step'], d.get('peak_mem_GiB'))"
    done
  done
done
for rep in 1 2; do
  for dis in "" knn_compact; do
    GKG_DISABLE=$dis python bench.py --workload stage1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('stage1', 'disable=[$dis]', d['ms_per_step'], d['hip_kernels'].get('mr_fwd'))"
  done
done
