#!/bin/bash
# Round-2: wait-state counters of the two latency-bound pyramid kernels.
set -u
TAG=${1:-r02zq}; mkdir -p gpurun_out/$TAG
G1="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"
G2="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
G3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
for wl in pyrdown_u8_4k pyrup_f32_4k; do
  echo "== $wl" | tee -a gpurun_out/$TAG/summary.txt
  bash scripts/diag/pmc_workload.sh $wl $TAG "$G1" "$G2" "$G3" | tee -a gpurun_out/$TAG/summary.txt
done
