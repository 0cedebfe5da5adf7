#!/bin/bash
# Round-4 visit 10: per-row spans in the affine staged gather (test option warp_u8_spans = 0 keeps the whole box): parity at the
# BASELINE batch, interleaved A/B, FETCH / WRITE counters of both.
set -u
TAG=${1:-r04y}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests/test_u8_gpu.py tests/test_dev_options_gpu.py tests/test_full_batch_gpu.py tests/test_bench_workloads_gpu.py -q -x -k "u8 or option" 2>&1 | tail -2 | tee "$OUT/pytest.log"
for r in 1 2 3; do
  for v in ${VALS:-0 -1}; do
    echo -n "round $r warp_u8_spans=$v  " | tee -a "$OUT/spans_ab.txt"
    timeout 300 python bench.py --workload warp_affine_u8_4k --no-cpu-baseline --also none --dev-option warp_u8_spans=$v 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/spans_ab.txt"
  done
done
for v in 0 -1; do
  bash scripts/diag/pmc_cmd.sh $TAG/hbm_$v "python $REPO/bench.py --workload warp_affine_u8_4k --steps 2 --warmup 1 --no-cpu-baseline --also none --dev-option warp_u8_spans=$v" \
    "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" 2>&1 | grep -i "gather\|kernel," | cut -c1-400
  cp "$OUT/hbm_$v/pmc_table.txt" "$OUT/hbm_${v}_counters.csv" 2>/dev/null
done
