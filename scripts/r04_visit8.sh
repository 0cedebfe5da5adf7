#!/bin/bash
# Round-4 visit 8: wide 16-byte tap loads in the NV12 quad kernel (pre_quads = 3 keeps the per-tap loads), three interleaved rounds.
set -u
TAG=${1:-r04q}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
WLS=${WLS:-nv12_chw_640 nv12_chw_608}
for r in 1 2 3; do
  for q in -1 3; do
    for wl in $WLS; do
      echo -n "round $r pre_quads=$q  " | tee -a "$OUT/wide_taps_ab.txt"
      timeout 300 python bench.py --workload $wl --no-cpu-baseline --also none --dev-option pre_quads=$q 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/wide_taps_ab.txt"
    done
  done
done
