#!/bin/bash
# (record of a finished A/B: the environment knob it flips was removed from the library with the variant that lost — see profiles/README.md)
# Round-3: staged u8 gather — one barrier per image (two-box layout) and 16 images per block against the previous configuration, one box.
set -u
TAG=${1:-r03_gather2}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_u8_gpu.py -x -q -m gpu -n 4 -k "warp or remap or gather" 2>&1 | tail -3 | tee "$OUT/pytest.log"
for round in 1 2; do
  for cfg in "1 0" "0 8" "1 8" "0 16"; do
    set -- $cfg
    echo "== KH_GATHER_TWO_BOXES=$1 KH_GATHER_NB=$2 (0 = default: 16 for batches >= 128) (round $round)" | tee -a "$OUT/ab.txt"
    KH_GATHER_TWO_BOXES=$1 KH_GATHER_NB=$2 timeout 600 python bench.py --workload warp_affine_u8_4k --no-cpu-baseline --also warp_perspective_u8_4k,remap_u8_4k --steps 10 --warmup 3 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/ab.txt"
  done
done
