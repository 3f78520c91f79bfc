B=tools/ubench/_knn_ablate/libgkg_hip_base.so
python -m pytest tests/test_hip_knn.py tests/test_hip_knn_mr_fused.py tests/test_hip_config_shapes.py -x -q 2>&1 | tail -3
for s in s1 s2 s3; do
  for rep in 1 2; do
    GKG_HIP_LIB=$PWD/$B python tools/ubench/knn_shape_run.py $s 4 2>&1 | grep -v amdgpu.ids | sed 's/^/base /'
    python tools/ubench/knn_shape_run.py $s 4 2>&1 | grep -v amdgpu.ids | sed 's/^/new  /'
  done
done
python tools/bench_knn_shapes.py 2>&1 | grep -v amdgpu.ids | sed 's/^/new  /'
GKG_HIP_LIB=$PWD/$B python tools/bench_knn_shapes.py 2>&1 | grep -v amdgpu.ids | sed 's/^/base /'
bash tools/replay_ab.sh "base|GKG_HIP_LIB=$PWD/$B" "new|GKG_X=1"
