#!/bin/bash
# (record of a finished A/B: the environment knob it flips was removed from the library with the variant that lost — see profiles/README.md)
set -u
TAG=${1:-r03z}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for i in 1 2 3; do for v in 0 1; do
  echo "run $i dense_stores=$v" | tee -a "$OUT/gather_ab.txt"
  KH_GATHER_DENSE_STORES=$v timeout 300 python bench.py --workload warp_affine_u8_4k --no-cpu-baseline --also warp_perspective_u8_4k,remap_u8_4k --steps 20 --warmup 5 2>&1 | grep '^{' | python scripts/bench_table.py | cut -c1-110 | tee -a "$OUT/gather_ab.txt"
done; done
