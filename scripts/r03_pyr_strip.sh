#!/bin/bash
set -u
TAG=${1:-r03q}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for s in 270 540 180 135 90 60 45 30; do
  echo -n "strip $s: " | tee -a "$OUT/pyr_strip.txt"
  KH_PYR_STRIP=$s timeout 300 python bench.py --workload pyrdown_u8_4k --no-cpu-baseline --also none --steps 20 --warmup 5 2>&1 | grep '^{' | python scripts/bench_table.py | head -1 | cut -c30-110 | tee -a "$OUT/pyr_strip.txt"
done
