H=$PWD/tools/ubench/_knn_ablate/libgkg_hip_head.so
python -m pytest tests/test_hip_ops.py tests/test_hip_config_shapes.py tests/test_hip_knn_mr_fused.py tests/test_hip_knn_map.py -x -q 2>&1 | tail -3
python tools/fuzz_ops.py --seconds 100 --seed 77 2>&1 | tail -1
for s in s1 s2 s3; do
  for rep in 1 2; do
    GKG_HIP_LIB=$H python tools/ubench/knn_shape_run.py $s 4 2>&1 | grep -v amdgpu.ids | sed 's/^/before /'
    python tools/ubench/knn_shape_run.py $s 4 2>&1 | grep -v amdgpu.ids | sed 's/^/ku6    /'
  done
done
for rep in 1 2; do
  GKG_HIP_LIB=$H python bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('before cfg5', j['ms_per_step'], j.get('ms_per_step_knn_bf16'))"
  python bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ku6    cfg5', j['ms_per_step'], j.get('ms_per_step_knn_bf16'))"
done
