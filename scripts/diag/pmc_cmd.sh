#!/bin/bash
# Dev: rocprofv3 --pmc passes over an arbitrary command; keeps rocprofv3's own CSV (counter_collection) per pass and prints a
# per-kernel mean table.   bash scripts/diag/pmc_cmd.sh <tag> "<command>" "<counter group 1>" ["<counter group 2>" ...]
# One group per pass (TCC: 4 slots, SQ: 8, GRBM: 2 — MI355X_MICROARCH.md "rocprofv3 PMC slots").  Never combined with tracing.
set -u
TAG=$1; CMD=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
i=0
for g in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $g --output-format csv -d "$OUT/pmc_$i" -o pmc -- $CMD > "$OUT/pmc_$i.log" 2>&1
  f=$(find "$OUT/pmc_$i" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" "$OUT/pmc_${i}_counter_collection.csv"; else echo "pass $i ($g): no counter_collection.csv"; tail -5 "$OUT/pmc_$i.log"; fi
  rm -rf "$OUT/pmc_$i"
done
cd "$REPO"
python scripts/diag/pmc_table.py "$OUT"/pmc_*_counter_collection.csv | tee "$OUT/pmc_table.txt"
