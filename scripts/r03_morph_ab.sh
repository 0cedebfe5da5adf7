#!/bin/bash
# Round-3: rolling planar RGB dilate / erode against the tile kernel on one box.
set -u
TAG=${1:-r03_morph}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pyramid_morph_gpu.py -x -q -m gpu -k morphology 2>&1 | tail -3 | tee "$OUT/pytest.log"
for round in 1 2; do
  for v in 1 0; do
    echo "== KH_MORPH_ROLL=$v (round $round)" | tee -a "$OUT/ab.txt"
    KH_MORPH_ROLL=$v timeout 600 python bench.py --workload dilate_u8_4k --no-cpu-baseline --also none --steps 10 --warmup 3 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/ab.txt"
  done
done
bash scripts/diag/pmc_workload.sh dilate_u8_4k $TAG "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE" 2>&1 | tail -20 | tee "$OUT/counters.txt"
