#!/bin/bash
# Dev: same-box A/B of the one-column and two-column rolling filter on C4 (interleaved).
set -u
TAG=${1:-r02zk}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; : > "$OUT/log.txt"
for i in 1 2 3; do
  for v in 0 1; do
    echo "== KH_FILTER_TWO_COLUMNS=$v run $i" | tee -a "$OUT/log.txt"
    KH_FILTER_TWO_COLUMNS=$v timeout 300 python bench.py --workload gaussian_4k --no-cpu-baseline --steps 20 --warmup 3 2>&1 | grep '^{' | python -c 'import json,sys
for l in sys.stdin:
    j=json.loads(l); r=j["roofline"]; print("   %8.3f ms/step  frac %.3f  launch mean %.3f min %.3f ms" % (j["ms_per_step"], r["frac"], r["mean_launch_ms"], r["min_launch_ms"]))' | tee -a "$OUT/log.txt"
  done
done
