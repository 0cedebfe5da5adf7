#!/bin/bash
# Round-4 visit 12: 64 x 16 destination tiles in 256-thread blocks (six blocks per CU) against 64 x 32 / 512 in the staged u8 gather.
set -u
TAG=${1:-r04z1}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests/test_u8_gpu.py tests/test_dev_options_gpu.py -q -x 2>&1 | tail -2 | tee "$OUT/pytest.log"
for r in 1 2 3; do
  for v in ${VALS:--1 16}; do
    echo "round $r warp_u8_rows=$v  " | tee -a "$OUT/rows_ab.txt"
    timeout 300 python bench.py --workload warp_affine_u8_4k --no-cpu-baseline --also warp_perspective_u8_4k,remap_u8_4k --dev-option warp_u8_rows=$v 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/rows_ab.txt"
  done
done
for v in 16; do
  bash scripts/diag/pmc_cmd.sh $TAG/hbm_$v "python $REPO/bench.py --workload warp_affine_u8_4k --steps 2 --warmup 1 --no-cpu-baseline --also warp_perspective_u8_4k,remap_u8_4k --dev-option warp_u8_rows=$v" \
    "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" 2>&1 | grep -i "gather\|kernel," | cut -c1-400
  cp "$OUT/hbm_$v/pmc_table.txt" "$OUT/hbm_${v}_counters.csv" 2>/dev/null
done
