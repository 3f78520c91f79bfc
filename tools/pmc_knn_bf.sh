#!/bin/bash
# SQ counters of the bf16-contraction k-NN kernel at one stage shape:  bash tools/pmc_knn_bf.sh s1
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVES SQ_INST_LEVEL_LDS"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$name -o run -- python3 $R/tools/bench_knn_bf.py "$@" > /tmp/pmc_$name.log 2>&1
  python3 $R/tools/pmc_knn.py /tmp/pmc_$name knn_tile
done
