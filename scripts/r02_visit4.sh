#!/bin/bash
# Round-2 visit 4: tiled morphology (parity + timing against the per-pixel kernel), Lanczos sampler before / after.
set -u
TAG=${1:-r02w}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== morphology parity" | tee "$OUT/log.txt"
timeout 600 python -m pytest tests/test_pyramid_morph_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee -a "$OUT/log.txt"
for v in 0 1; do
  echo "== dilate_u8_4k KH_MORPH_DIRECT=$v" | tee -a "$OUT/log.txt"
  KH_MORPH_DIRECT=$v timeout 300 python bench.py --workload dilate_u8_4k --no-cpu-baseline --steps 10 --warmup 2 2>&1 | grep '^{' | tee -a "$OUT/log.txt"
done
if [ -f gpurun_ab/libkornia_hip_oldlz.so ]; then
  for lib in "" gpurun_ab/libkornia_hip_oldlz.so; do
    echo "== nv12_chw_640_lanczos lib=${lib:-HEAD}" | tee -a "$OUT/log.txt"
    KORNIA_HIP_LIB=${lib:+$(pwd)/$lib} timeout 300 python bench.py --workload nv12_chw_640_lanczos --no-cpu-baseline --steps 10 --warmup 2 2>&1 | grep '^{' | tee -a "$OUT/log.txt"
  done
fi
