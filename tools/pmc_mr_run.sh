#!/bin/bash
# SQ counters of the aggregation (gather / scatter) kernels for one bench workload:  bash tools/pmc_mr_run.sh [bench args]
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmcmr_$name
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmcmr_$name -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tune --no-graph "$@" > /tmp/pmcmr_$name.log 2>&1
  python $R/tools/pmc_knn.py /tmp/pmcmr_$name mr_
done
