#!/bin/bash
# Round-2 visit: SQ counters of the three tiled u8 kernels.
set -u
TAG=${1:-r02zb}
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G3="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
for wl in dilate_u8_4k pyrdown_u8_4k resize_u8_224; do
  echo "== $wl" | tee -a gpurun_out/$TAG/summary.txt 2>/dev/null || { mkdir -p gpurun_out/$TAG; echo "== $wl" > gpurun_out/$TAG/summary.txt; }
  bash scripts/diag/pmc_workload.sh $wl $TAG "$G1" "$G2" "$G3" | tee -a gpurun_out/$TAG/summary.txt
done
