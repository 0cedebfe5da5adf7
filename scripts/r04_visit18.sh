#!/bin/bash
# Round-4 visit 18: the LDS-staged f32 resize against the per-pixel kernel (test option resize_staged = 0), interleaved, + counters.
set -u
TAG=${1:-r04zl}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests/test_geom_gpu.py tests/test_dev_options_gpu.py tests/test_full_batch_gpu.py -q -x -k "resize or option" 2>&1 | tail -2 | tee "$OUT/pytest.log"
for r in 1 2 3; do
  for v in 0 -1; do
    echo -n "round $r resize_staged=$v  " | tee -a "$OUT/resize_staged_ab.txt"
    timeout 300 python bench.py --workload resize_bicubic_540 --no-cpu-baseline --also none --dev-option resize_staged=$v 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/resize_staged_ab.txt"
  done
done
bash scripts/diag/pmc_cmd.sh $TAG/staged "python $REPO/bench.py --workload resize_bicubic_540 --steps 2 --warmup 1 --no-cpu-baseline --also none" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum" "FETCH_SIZE" "WRITE_SIZE" 2>&1 | tail -3 | cut -c1-500
