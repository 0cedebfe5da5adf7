#!/bin/bash
# Round-3 A/B of the C4 gaussian: one column per lane (default) vs four columns per lane (KH_FILTER_FOUR_COLUMNS=1), interleaved.
set -u
TAG=${1:-r03k}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for i in 1 2 3; do
  for v in 0 1; do
    echo -n "run $i four_columns=$v: " | tee -a "$OUT/filter_ab.txt"
    KH_FILTER_FOUR_COLUMNS=$v timeout 300 python bench.py --workload gaussian_4k --no-cpu-baseline --also none --steps 20 --warmup 5 2>&1 | grep '^{' | python scripts/bench_table.py | head -1 | tee -a "$OUT/filter_ab.txt"
  done
done
