#!/bin/bash
# Hardware-counter passes for the hand-written kernels of the cfg2 step (run on the GPU box from the repo root):
#   bash tools/pmc_refresh.sh r02
# Three separate rocprofv3 passes (counters never share a run with --stats / other trace domains, and FETCH_SIZE /
# WRITE_SIZE do not fit one pass: MI355X guide, "rocprofv3 PMC slots"), each over an eager (no hipGraph) run of bench.py
# so that every kernel of the step is its own dispatch.  Leaves gpurun_out/<tag>_pmc.json; copy it to profiles/.
TAG=${1:-pmc}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$name -o run -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-tune --no-graph > /tmp/pmc_$name.log 2>&1
done
cd $R
python tools/pmc_json.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_WAVE_CYCLES > gpurun_out/${TAG}_pmc.json
head -c 1500 gpurun_out/${TAG}_pmc.json
