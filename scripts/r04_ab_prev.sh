#!/bin/bash
# Same-box A/B of the current library against kornia-rs_amd/lib/libkornia_hip_prev.so (scripts/build_prev_lib.sh), interleaved rounds.
#   bash scripts/r04_ab_prev.sh <tag> <main workload> <also list> [rounds] [extra bench args]
set -u
TAG=$1; WL=$2; ALSO=$3; ROUNDS=${4:-3}; EXTRA=${5:-}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for r in $(seq 1 $ROUNDS); do
  for which in prev new; do
    echo "round $r $which" | tee -a "$OUT/ab.txt"
    if [ $which = prev ]; then export KORNIA_HIP_LIB=$(pwd)/kornia-rs_amd/lib/libkornia_hip_prev.so; else unset KORNIA_HIP_LIB; fi
    timeout 300 python bench.py --workload $WL --no-cpu-baseline --also $ALSO $EXTRA 2>&1 | grep '^{' | python scripts/bench_table.py | cut -c1-125 | tee -a "$OUT/ab.txt"
  done
done
unset KORNIA_HIP_LIB
