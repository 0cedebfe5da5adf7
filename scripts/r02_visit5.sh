#!/bin/bash
# Round-2 visit: kernel-trace split of the multi-kernel / low-fraction opt-in workloads.
set -u
TAG=${1:-r02y}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof" -o kt -- python "$REPO/bench.py" --workload resize_u8_224 --no-cpu-baseline --steps 5 --warmup 2 \
  --also pyrdown_u8_4k,dilate_u8_4k,box_blur_fast_1080p,median5_u8_1080p,bilateral_1080p,lab_from_rgb_4k,spatial_gradient_1080p,resize_norm_chw_224 > "$REPO/$OUT/prof.log" 2>&1
cd "$REPO"
db=$(find "$OUT/prof" -name '*.db' | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py "$db" | grep -v "rocclr" > "$OUT/kernel_stats.csv" && head -30 "$OUT/kernel_stats.csv" | cut -c1-220
find "$OUT" -name '*.db' -delete
