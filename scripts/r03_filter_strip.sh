#!/bin/bash
set -u
TAG=${1:-r03l}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for s in 360 90 180 540 1080 2160; do
  for v in 1 0; do
    echo -n "strip $s four_columns=$v: " | tee -a "$OUT/filter_strip.txt"
    KH_FILTER_STRIP=$s KH_FILTER_FOUR_COLUMNS=$v timeout 300 python bench.py --workload gaussian_4k --no-cpu-baseline --also none --steps 20 --warmup 5 2>&1 | grep '^{' | python scripts/bench_table.py | head -1 | cut -c30-110 | tee -a "$OUT/filter_strip.txt"
  done
done
