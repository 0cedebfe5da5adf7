#!/bin/bash
# Round-4 visit 17: per-step times of box_blur_4k / sobel_4k / gaussian_4k alone and in the default order (the box blur's step is
# 5.2 ms in the default line, its kernel 4.46 ms in the kernel trace of the same command).
set -u
TAG=${1:-r04zj}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_full.json"))
for r in [d] + d.get("also", []):
    print(sys.argv[1], r["config"]["workload"], r["ms_per_step"], r["roofline"].get("launch_ms"))
PY
}
timeout 300 python bench.py --workload box_blur_4k --steps 12 --no-cpu-baseline --also none > /dev/null 2>&1; show "alone:" | tee -a "$OUT/steps.txt"
timeout 300 python bench.py --workload gaussian_4k --steps 10 --no-cpu-baseline --also box_blur_4k,sobel_4k > /dev/null 2>&1; show "after gaussian:" | tee -a "$OUT/steps.txt"
timeout 300 python bench.py --workload sobel_4k --steps 10 --no-cpu-baseline --also box_blur_4k,gaussian_4k > /dev/null 2>&1; show "after sobel:" | tee -a "$OUT/steps.txt"
timeout 300 python bench.py --workload resize_bicubic_540 --steps 10 --no-cpu-baseline --also box_blur_4k > /dev/null 2>&1; show "after bicubic:" | tee -a "$OUT/steps.txt"
