#!/bin/bash
# Round-4 visit 16: per-step times of the H2D workload, alone / after another workload / with and without the CPU baseline legs.
set -u
TAG=${1:-r04zh}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_full.json"))
for r in [d] + d.get("also", []):
    if "h2d" in r["config"]["workload"]:
        print(sys.argv[1], r["config"]["workload"], r["ms_per_step"], r["roofline"].get("launch_ms"), r["roofline"].get("end_to_end_frac_of_pinned_h2d"))
PY
}
timeout 300 python bench.py --workload nv12_h2d_preprocess --steps 20 --no-cpu-baseline --also none > /dev/null 2>&1; show "alone, no cpu leg:" | tee -a "$OUT/h2d_steps.txt"
timeout 300 python bench.py --workload nv12_h2d_preprocess --steps 20 --also none > /dev/null 2>&1; show "alone, cpu leg:" | tee -a "$OUT/h2d_steps.txt"
timeout 300 python bench.py --workload nv12_chw_640 --steps 10 --no-cpu-baseline --also nv12_h2d_preprocess > /dev/null 2>&1; show "after nv12_chw_640, no cpu leg:" | tee -a "$OUT/h2d_steps.txt"
timeout 300 python bench.py --workload nv12_chw_640 --steps 10 --also nv12_h2d_preprocess > /dev/null 2>&1; show "after nv12_chw_640, cpu legs:" | tee -a "$OUT/h2d_steps.txt"
timeout 300 python bench.py --workload nv12_chw --steps 10 --no-cpu-baseline --also nv12_chw_640,nv12_chw_608,yuyv_chw_640,nv12_h2d_preprocess > /dev/null 2>&1; show "default order, no cpu legs:" | tee -a "$OUT/h2d_steps.txt"
