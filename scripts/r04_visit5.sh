#!/bin/bash
# Round-4 visit 5: upload ring + streaming pyrdown_f32 + normalize on the device: tests, then A/B of pyrdown_f32 (pyr_direct = the
# round-3 per-pixel kernel), the H2D workloads, the touched workloads.
set -u
TAG=${1:-r04e}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_preprocess_gpu.py tests/test_pyramid_morph_gpu.py tests/test_dev_options_gpu.py tests/test_pointwise_gpu.py tests/test_sharding_gpu.py tests/test_bench_workloads_gpu.py -m gpu -q -x ) > "$OUT/pytest.log" 2>&1; tail -5 "$OUT/pytest.log"
for r in 1 2 3; do
  for d in 0 1; do
    echo -n "round $r pyr_direct=$d  " | tee -a "$OUT/pyrdown_ab.txt"
    timeout 300 python bench.py --workload pyrdown_f32_4k --no-cpu-baseline --also none --dev-option pyr_direct=$d 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/pyrdown_ab.txt"
  done
done
timeout 600 python bench.py --workload nv12_h2d_preprocess --no-cpu-baseline --also nv12_h2d_preprocess_pageable,normalize_1080p,nv12_chw_640,yuyv_chw_640 2>&1 | grep '^{' | tee "$OUT/bench_h2d.log" | python scripts/bench_table.py | tee "$OUT/bench_h2d.txt"
python - <<'PY' | tee -a "$OUT/bench_h2d.txt"
import json
j = json.load(open("gpurun_out/bench_full.json"))
for rec in [j] + j.get("also", []):
    r = rec["roofline"]
    ex = {k: v for k, v in r.items() if k.startswith(("h2d_", "kernel_only", "end_to_end", "hidden_by", "serial_sum", "source"))}
    if ex: print(rec["config"]["workload"], ex)
PY
