#!/bin/bash
# Round-3: same-box A/B of library builds (kornia-rs_amd/lib/libkornia_hip*.so) on a list of workloads.
#   bash scripts/r03_lib_ab.sh <tag> "<wl,...>" "<lib suffixes, '' = the product build>"
set -u
TAG=$1; WLS=$2; LIBS=$3; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
first=${WLS%%,*}; rest=${WLS#*,}; [ "$rest" = "$WLS" ] && rest=none
for round in 1 2 3; do
  for v in $LIBS; do
    [ "$v" = "-" ] && lib=kornia-rs_amd/lib/libkornia_hip.so || lib=kornia-rs_amd/lib/libkornia_hip_$v.so
    echo "== $lib (round $round)" | tee -a "$OUT/ab.txt"
    KORNIA_HIP_LIB=$PWD/$lib timeout 600 python bench.py --workload $first --no-cpu-baseline --also "$rest" --steps 10 --warmup 3 2>&1 | grep '^{' | python scripts/bench_table.py | tee -a "$OUT/ab.txt"
  done
done
