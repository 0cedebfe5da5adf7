#!/bin/bash
# Round-2 closing counters: what bounds the rewritten kernels now (instruction mix, VALU / LDS busy cycles).
set -u
TAG=${1:-r02zp}; mkdir -p gpurun_out/$TAG
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
G3="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"
for wl in dilate_u8_4k pyrdown_u8_4k pyrup_u8_4k pyrup_f32_4k bilateral_1080p; do
  echo "== $wl" | tee -a gpurun_out/$TAG/summary.txt
  bash scripts/diag/pmc_workload.sh $wl $TAG "$G1" "$G3" | tee -a gpurun_out/$TAG/summary.txt
done
