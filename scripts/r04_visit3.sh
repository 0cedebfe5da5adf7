#!/bin/bash
# Round-4 visit 3: one-store-per-wave variants of the north-star kernel (scripts/ubench/nv12_r04.hip), then limiter counters of the
# letterbox kernels (on-grid 640, off-grid 608).
set -u
TAG=${1:-r04c}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
timeout 300 scripts/ubench/bin/nv12_r04 1024 7 | tee "$OUT/ubench_nv12_r04.txt"
for wl in nv12_chw_640 nv12_chw_608; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --also none 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/letterbox.txt"
  bash scripts/diag/pmc_cmd.sh $TAG/lim_$wl "python $REPO/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --also none" \
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES" \
    "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum" \
    "FETCH_SIZE" "WRITE_SIZE" 2>&1 | tail -8
done
