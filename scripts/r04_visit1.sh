#!/bin/bash
# Round-4 visit 1: full-batch parity, the store-shape gap with counters (item 3), FETCH_SIZE calibration
# (item 2c), limiter counters of the two unexplained kernels (item 6).
set -u
TAG=${1:-r04a}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
echo "== full-batch parity"
( time timeout 1200 python -m pytest tests/test_full_batch_gpu.py -m gpu -q -x ) > "$OUT/pytest_full_batch.log" 2>&1; tail -6 "$OUT/pytest_full_batch.log"
echo "== store shapes (timing)"
timeout 300 scripts/ubench/bin/store_gap_r04 1024 7 | tee "$OUT/store_gap.txt"
echo "== store shapes (counters)"
bash scripts/diag/pmc_cmd.sh $TAG/store_pmc "$REPO/scripts/ubench/bin/store_gap_r04 1024 2" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum GRBM_GUI_ACTIVE" \
  "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_LEVEL_sum" \
  "TCC_REQ_sum TCC_WRITE_sum TCC_CYCLE_sum TCC_EA0_RDREQ_sum" \
  "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR" 2>&1 | tail -40
echo "== FETCH calibration"
timeout 300 scripts/ubench/bin/fetch_calib_r04 | tee "$OUT/fetch_calib.txt"
bash scripts/diag/pmc_cmd.sh $TAG/fetch_pmc "$REPO/scripts/ubench/bin/fetch_calib_r04" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
  "FETCH_SIZE" "TCC_BUBBLE_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" 2>&1 | tail -40
echo "== limiter counters: resize_u8_224, pyrdown_f32_4k"
for wl in resize_u8_224 pyrdown_f32_4k; do
  bash scripts/diag/pmc_cmd.sh $TAG/lim_$wl "python $REPO/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --also none" \
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_CYCLES" \
    "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum" \
    "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
    "FETCH_SIZE" "WRITE_SIZE" 2>&1 | tail -30
done
du -sh "$OUT"
