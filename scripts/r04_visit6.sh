#!/bin/bash
# Round-4 visit 6: separable u8 resize — dot4 horizontal pass + LDS-staged vertical pass against the round-3 kernels (test option
# resize_u8_gather = 2), tests, and a kernel trace splitting the two passes.
set -u
TAG=${1:-r04f}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
( time timeout 900 python -m pytest tests/test_resize_u8_gpu.py tests/test_workspace_cache_gpu.py tests/test_host_api_gpu.py tests/test_cie.py -m gpu -q -x ) > "$OUT/pytest.log" 2>&1; tail -4 "$OUT/pytest.log"
for r in 1 2 3; do
  for g in -1 2; do
    echo -n "round $r resize_u8_gather=$g  " | tee -a "$OUT/resize_u8_ab.txt"
    timeout 300 python bench.py --workload resize_u8_224 --no-cpu-baseline --also none --dev-option resize_u8_gather=$g 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/resize_u8_ab.txt"
  done
done
cd /tmp
for g in -1 2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/kt_$g" -o kt -- python "$REPO/bench.py" --workload resize_u8_224 --no-cpu-baseline --also none --dev-option resize_u8_gather=$g > "$REPO/$OUT/kt_$g.log" 2>&1
  f=$(find "$REPO/$OUT/kt_$g" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$REPO/$OUT/resize_u8_gather${g}_kernel_stats.csv" && head -5 "$f" | cut -c1-200
  rm -rf "$REPO/$OUT/kt_$g"
done
