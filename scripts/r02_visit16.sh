#!/bin/bash
# Round-2: L1 access counters of the C5 f32 gathers and the C4 filter (is the L1 access rate their limit too?).
set -u
TAG=${1:-r02zu}; mkdir -p gpurun_out/$TAG
G="TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_VALU"
for wl in undistort_warp_4k nv12_chw_640; do
  echo "== $wl" | tee -a gpurun_out/$TAG/summary.txt
  bash scripts/diag/pmc_workload.sh $wl $TAG "$G" | tee -a gpurun_out/$TAG/summary.txt
done
