# what the x6 kernels would cost without the in-kernel operand split (upper bound for operand planes written by the producers)
L=$PWD/tools/ubench/_knn_ablate/libgkg_hip_x6nosplit.so
bash tools/replay_ab.sh "shipped|GKG_X=1" "nosplit|GKG_HIP_LIB=$L"
for w in stage1 stage3; do
  for rep in 1 2; do
    python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shipped $w', j['ms_per_step'])"
    GKG_HIP_LIB=$L python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nosplit $w', j['ms_per_step'])"
  done
done
grep -h "gemm_x6\|wgrad_x6" gpurun_out/replay_shipped.txt | head -8
echo ---
grep -h "gemm_x6\|wgrad_x6" gpurun_out/replay_nosplit.txt | head -8
