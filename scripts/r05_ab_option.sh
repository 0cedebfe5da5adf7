#!/bin/bash
# Same-box A/B of a library test option against the production choice, interleaved rounds, same process layout.
#   bash scripts/r05_ab_option.sh <tag> <workload> <option=value> [rounds] [also list]
set -u
TAG=$1; WL=$2; OPT=$3; ROUNDS=${4:-3}; ALSO=${5:-none}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "# $WL: production vs --dev-option $OPT, $ROUNDS interleaved rounds (bench.py --no-cpu-baseline, default steps / warmup)" | tee "$OUT/ab.txt"
for r in $(seq 1 $ROUNDS); do
  for which in option production; do
    echo "round $r $which" | tee -a "$OUT/ab.txt"
    if [ $which = option ]; then X="--dev-option $OPT"; else X=""; fi
    timeout 300 python bench.py --workload $WL --no-cpu-baseline --also $ALSO $X 2>&1 | grep '^{' | python scripts/bench_table.py | cut -c1-125 | tee -a "$OUT/ab.txt"
  done
done
