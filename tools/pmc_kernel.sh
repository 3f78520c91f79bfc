#!/bin/bash
# Hardware counters of the kernels whose name contains <substring> over an eager run of bench.py (separate rocprofv3 passes):
#   bash tools/pmc_kernel.sh wgrad_x6_batch [bench args]
R=${GRAFT_REPO_ROOT:-$PWD}
SUB=$1; shift
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$name -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tune --no-graph "$@" > /tmp/pmc_$name.log 2>&1
  python3 $R/tools/pmc_knn.py /tmp/pmc_$name "$SUB"
done
