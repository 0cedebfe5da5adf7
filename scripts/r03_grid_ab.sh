#!/bin/bash
# Round-3: bilinear preprocess whose taps all sit on whole source pixels (1080p -> 640 letterbox): one-tap kernel vs the four-tap kernel, one box.
set -u
TAG=${1:-r03_grid}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_preprocess_gpu.py tests/test_abi.py -x -q -m gpu -n 4 2>&1 | tail -3 | tee "$OUT/pytest.log"
for round in 1 2 3; do
  for v in 1 0; do
    echo "== KH_PRE_GRID=$v (round $round)" | tee -a "$OUT/ab.txt"
    KH_PRE_GRID=$v timeout 600 python bench.py --workload nv12_chw_640 --no-cpu-baseline --also none --steps 10 --warmup 3 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/ab.txt"
  done
done
bash scripts/diag/pmc_workload.sh nv12_chw_640 $TAG "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM" 2>&1 | tail -14 | tee "$OUT/counters.txt"
