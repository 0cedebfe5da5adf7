#!/bin/bash
# Round-4 visit 2: the store-shape GRID (block size x stores per thread x layout x occupancy), timing + two counter passes.
set -u
TAG=${1:-r04b}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
timeout 600 scripts/ubench/bin/store_gap_r04 1024 7 | tee "$OUT/store_grid.txt"
bash scripts/diag/pmc_cmd.sh $TAG/store_pmc "$REPO/scripts/ubench/bin/store_gap_r04 1024 1" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" \
  "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" \
  "TA_TA_BUSY_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_WRITE_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" 2>&1 | tail -60
