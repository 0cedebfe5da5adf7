#!/bin/bash
# One GPU-box visit: every bench workload + rocprofv3 kernel-trace stats + HBM PMC passes.
# Usage (repo root, via gpurun):  bash scripts/gpu_bench_all.sh [tag] [workloads...]
set -u
TAG=${1:-r01}; shift || true
WLS=${*:-"nv12_chw nv12_chw_640 resize_224 gaussian_4k undistort_warp_4k"}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$(pwd)
for wl in $WLS; do
  echo "== bench $wl" | tee -a "$OUT/bench.log"
  timeout 600 python bench.py --steps 20 --warmup 5 --workload $wl 2>&1 | grep '^{' | tee -a "$OUT/bench.log"
done
cd /tmp
for wl in $WLS; do
  echo "== rocprofv3 kernel-trace $wl"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_$wl" -o kt -- \
      python "$REPO/bench.py" --steps 20 --warmup 5 --workload $wl --no-cpu-baseline > "$REPO/$OUT/prof_$wl.log" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c -d "$REPO/$OUT/pmc_${wl}_$c" -o pmc -- \
        python "$REPO/bench.py" --steps 3 --warmup 1 --workload $wl --no-cpu-baseline > "$REPO/$OUT/pmc_${wl}_$c.log" 2>&1
  done
done
cd "$REPO"
for wl in $WLS; do
  db=$(find "$OUT/prof_$wl" -name '*.db' | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py "$db" > "$OUT/${wl}_kernel_stats.csv" && head -4 "$OUT/${wl}_kernel_stats.csv"
  for c in FETCH_SIZE WRITE_SIZE; do
    db=$(find "$OUT/pmc_${wl}_$c" -name '*.db' | head -1)
    [ -n "$db" ] && python scripts/rocpd_summary.py "$db" > "$OUT/${wl}_pmc_$c.csv" && tail -4 "$OUT/${wl}_pmc_$c.csv"
  done
done
# keep the merged-back payload small
find "$OUT" -name '*.db' -size +20M -delete
du -sh "$OUT"
