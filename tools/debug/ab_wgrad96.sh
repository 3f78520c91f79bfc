# 64 x 96 weight-gradient tiles (this build) vs the build before them (HEAD), same box
H=$PWD/tools/ubench/_knn_ablate/libgkg_hip_head.so
python -m pytest tests/test_hip_gemm_x6.py tests/test_hip_wgrad_batch.py -x -q 2>&1 | tail -3
for w in stage1 cfg4; do
  for rep in 1 2; do
    GKG_HIP_LIB=$H python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('before $w', j['ms_per_step'])"
    python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('64x96  $w', j['ms_per_step'])"
  done
done
GKG_HIP_LIB=$H X6_ONLY=1 python tools/bench_x6.py wgradcfg4 2>&1 | grep -v amdgpu.ids | sed 's/^/before /'
X6_ONLY=1 python tools/bench_x6.py wgradcfg4 2>&1 | grep -v amdgpu.ids | sed 's/^/64x96  /' 
