#!/bin/bash
# Round-4 visit 14: block -> tile order of the staged u8 gather: column-major inside each XCD's band of tile rows (test option
# warp_u8_order = 1) so that the blocks in flight on an XCD form a 2-D patch and vertical neighbours share their box rows in L2.
set -u
TAG=${1:-r04z5}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests/test_u8_gpu.py tests/test_dev_options_gpu.py -q -x 2>&1 | tail -2 | tee "$OUT/pytest.log"
for r in 1 2 3; do
  for v in -1 1; do
    echo "round $r warp_u8_order=$v  " | tee -a "$OUT/order_ab.txt"
    timeout 300 python bench.py --workload warp_affine_u8_4k --no-cpu-baseline --also warp_perspective_u8_4k,remap_u8_4k --dev-option warp_u8_order=$v 2>&1 | grep '^{' | python scripts/bench_table.py | cut -c1-125 | tee -a "$OUT/order_ab.txt"
  done
done
for v in -1 1; do
  bash scripts/diag/pmc_cmd.sh $TAG/hbm_$v "python $REPO/bench.py --workload warp_affine_u8_4k --steps 2 --warmup 1 --no-cpu-baseline --also warp_perspective_u8_4k,remap_u8_4k --dev-option warp_u8_order=$v" \
    "FETCH_SIZE" "WRITE_SIZE" 2>&1 | grep -i "gather\|kernel," | cut -c1-200
  cp "$OUT/hbm_$v/pmc_table.txt" "$OUT/hbm_${v}_counters.csv" 2>/dev/null
done
