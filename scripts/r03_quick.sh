#!/bin/bash
# Round-3 quick visit: time a few workloads (no CPU baseline) and optionally collect SQ counters for one of them.
#   bash scripts/r03_quick.sh <tag> "<wl1,wl2,...>" [counter-workload]
set -u
TAG=$1; WLS=$2; CW=${3:-}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
first=${WLS%%,*}; rest=${WLS#*,}; [ "$rest" = "$WLS" ] && rest=none
timeout 600 python bench.py --workload $first --no-cpu-baseline --also "$rest" --steps 10 --warmup 3 2>&1 | grep '^{' | tee "$OUT/bench.log" | python scripts/bench_table.py | tee "$OUT/bench_table.txt"
if [ -n "$CW" ]; then
  bash scripts/diag/pmc_workload.sh $CW $TAG "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE" 2>&1 | tail -20 | tee "$OUT/counters_$CW.txt"
fi
