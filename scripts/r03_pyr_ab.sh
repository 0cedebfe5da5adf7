#!/bin/bash
# (record of a finished A/B: the environment knob it flips was removed from the library with the variant that lost — see profiles/README.md)
set -u
TAG=${1:-r03x}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for i in 1 2 3; do for v in 0 1; do
  echo -n "run $i dense_stores=$v: " | tee -a "$OUT/pyr_ab.txt"
  KH_PYR_DENSE_STORES=$v timeout 300 python bench.py --workload pyrdown_u8_4k --no-cpu-baseline --also none --steps 20 --warmup 5 2>&1 | grep '^{' | python scripts/bench_table.py | head -1 | cut -c30-110 | tee -a "$OUT/pyr_ab.txt"
done; done
