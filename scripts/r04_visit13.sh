#!/bin/bash
# Round-4 visit 13 (instrumented build: git apply scripts/diag/u8_gather_ablation_r04.patch on commit 0f of this round, rebuild; never committed applied): ablation of the staged u8 gather — bits of warp_u8_lds_pitch >> 8:
# 1 no global stores, 2 no blend, 4 no LDS tap reads, 8 no staging loads after the first image, 16 no barriers.
set -u
TAG=${1:-r04z3}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for a in 0 1 2 4 8 16 3 6 7 12 15 31 24 9; do
  echo "ablate=$a" | tee -a "$OUT/ablate.txt"
  timeout 300 python bench.py --workload warp_affine_u8_4k --no-cpu-baseline --also warp_perspective_u8_4k,remap_u8_4k --steps 6 --warmup 2 --dev-option warp_u8_lds_pitch=$((a*256)) 2>&1 | grep '^{' | python scripts/bench_table.py | cut -c1-100 | tee -a "$OUT/ablate.txt"
done
