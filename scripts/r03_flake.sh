#!/bin/bash
# Is test_north_star_workloads flaky under concurrency?  4 processes x N rounds of that one test; failures keep their message.
set -u
TAG=${1:-r03n}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for round in 1 2 3 4; do
  for w in 1 2 3 4; do
    ( timeout 300 python -m pytest "tests/test_bench_workloads_gpu.py::test_north_star_workloads" tests/test_bench_workloads_gpu.py::test_filter_workloads_4k -m gpu -q -x -p no:cacheprovider > "$OUT/r${round}_w${w}.log" 2>&1; echo "round $round worker $w rc $?" >> "$OUT/summary.txt" ) &
  done
  wait
done
cat "$OUT/summary.txt"; grep -h "AssertionError" "$OUT"/r*_w*.log | cut -c1-400 | head -20
