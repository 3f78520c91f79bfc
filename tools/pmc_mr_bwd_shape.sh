#!/bin/bash
# HBM-side traffic and L2 hit rate of one aggregation-backward shape under forced kernel flags (separate rocprofv3 passes):
#   bash tools/pmc_mr_bwd_shape.sh s1 0x20800
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmcmb_$name
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmcmb_$name -o run -- python3 $R/tools/ubench/mr_bwd_shape.py "$@" 6 > /tmp/pmcmb_$name.log 2>&1
  python3 $R/tools/pmc_knn.py /tmp/pmcmb_$name mr_bwd
done
