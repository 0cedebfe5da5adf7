#!/bin/bash
# Dev (round 5): time a bench workload with ablated library builds (kornia-rs_amd/lib/libkornia_hip_abl_<name>.so, built out of tree:
# pieces of one kernel removed to see what each costs) against the production library, interleaved, same box.
#   bash scripts/diag/ablation_run_r05.sh <tag> <workload> "<name> <name> ..." [rounds] [also]
set -u
TAG=$1; WL=$2; NAMES=$3; ROUNDS=${4:-2}; ALSO=${5:-none}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for r in $(seq 1 $ROUNDS); do
  for v in prod $NAMES; do
    if [ $v = prod ]; then unset KORNIA_HIP_LIB; else export KORNIA_HIP_LIB=$(pwd)/kornia-rs_amd/lib/libkornia_hip_abl_$v.so; fi
    printf "%-10s " $v | tee -a "$OUT/ablation.txt"
    python bench.py --workload $WL --no-cpu-baseline --also $ALSO 2>&1 | grep '^{' | python scripts/bench_table.py | head -1 | cut -c1-118 | tee -a "$OUT/ablation.txt"
  done
done
unset KORNIA_HIP_LIB
