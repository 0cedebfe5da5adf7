#!/bin/bash
# Round-3 visit 1: the time-boxed north-star ISA / mapping attempt (scripts/ubench/nv12_r03.hip), the device tests touched by
# this round's host changes (workspace registry, Lanczos weight tables, runtime checks), and the new one-line bench format.
set -u
TAG=${1:-r03a}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== nv12_r03 ubench" | tee "$OUT/ubench_nv12.txt"
timeout 300 scripts/ubench/bin/nv12_r03 1024 7 2>&1 | tee -a "$OUT/ubench_nv12.txt"
echo "== device tests (subset)"
( time timeout 900 python -m pytest tests/test_preprocess_gpu.py tests/test_workspace_cache_gpu.py tests/test_bench_workloads_gpu.py tests/test_sharding_gpu.py tests/test_host_api_gpu.py tests/test_fuzz_gpu.py -m gpu -q -n 4 ) > "$OUT/pytest_subset.log" 2>&1
tail -6 "$OUT/pytest_subset.log"
echo "== default bench"
( time timeout 900 python bench.py ) > "$OUT/bench_raw.log" 2>&1
grep '^{' "$OUT/bench_raw.log" > "$OUT/bench.log"; python scripts/bench_table.py < "$OUT/bench.log" | tee "$OUT/bench_table.txt"
tail -4 "$OUT/bench_raw.log" | grep -v '^{'
cp gpurun_out/bench_full.json "$OUT/bench_full.json" 2>/dev/null
