for rep in 1 2; do
  for pf in "" force; do
    GKG_KNN_PREFILTER=$pf python bench.py --workload cfg5 --no-cpu-baseline --knn exact --steps 8 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('cfg5', 'GKG_KNN_PREFILTER=[$pf]', d['ms_per_step'], d['hip_kernels'].get('knn_tile'))"
  done
done
