#!/bin/bash
# SQ counters of the k-NN kernel at pvig_m's stage shapes:  bash tools/pmc_knn_shape.sh s1 s3
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for sh in "$@"; do
  echo "== $sh"
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"; do
    name=$(echo $pass | cut -d' ' -f1)
    rm -rf /tmp/pmcs_$name
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmcs_$name -o run -- python3 $R/tools/ubench/knn_shape_run.py $sh 3 > /tmp/pmcs_$name.log 2>&1
    python $R/tools/pmc_knn.py /tmp/pmcs_$name knn_tile || tail -3 /tmp/pmcs_$name.log
  done
  rm -rf /tmp/kt_$sh
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$sh -o run -- python3 $R/tools/ubench/knn_shape_run.py $sh 3 > /dev/null 2>&1
  grep -h "knn_tile" /tmp/kt_$sh/*/*kernel_stats.csv /tmp/kt_$sh/*kernel_stats.csv 2>/dev/null | cut -c1-200
done
