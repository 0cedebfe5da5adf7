#!/bin/bash
# Round-4 visit 4: the flattened-quads generic preprocess kernel, A/B through the test option pre_quads (0 = per-pixel kernel,
# 1 / 2 = quads per lane), three interleaved rounds on one box; then the changed test files and the new default bench line.
set -u
TAG=${1:-r04d}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for r in 1 2 3; do
  for q in 0 1 2; do
    for wl in nv12_chw_640 nv12_chw_608 yuyv_chw_640; do
      echo -n "round $r pre_quads=$q  " | tee -a "$OUT/quads_ab.txt"
      timeout 300 python bench.py --workload $wl --no-cpu-baseline --also none --dev-option pre_quads=$q 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/quads_ab.txt"
    done
  done
done
( time timeout 1500 python -m pytest tests/test_dev_options_gpu.py tests/test_preprocess_gpu.py tests/test_filter_gpu.py tests/test_resize_u8_gpu.py tests/test_u8_gpu.py tests/test_zz_host_extras_gpu.py tests/test_bench_workloads_gpu.py -m gpu -q -x ) > "$OUT/pytest_changed.log" 2>&1; tail -5 "$OUT/pytest_changed.log"
( time timeout 900 python bench.py ) > "$OUT/bench_raw.log" 2>&1
grep '^{' "$OUT/bench_raw.log" > "$OUT/bench.log"; python scripts/bench_table.py < "$OUT/bench.log" | tee "$OUT/bench_table.txt"; grep "^real" "$OUT/bench_raw.log"
cp gpurun_out/bench_full.json "$OUT/bench_full.json" 2>/dev/null
