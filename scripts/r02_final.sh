#!/bin/bash
# Round-2 closing visit: the whole device suite (4 xdist workers share the GPU; anything that fails is re-run serially), smoke, then
# scripts/r02_visit3.sh (default + opt-in bench tables, kernel trace, HBM PMC passes).
set -u
TAG=${1:-r02zf}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -n 4 ) > "$OUT/pytest_full.log" 2>&1
tail -5 "$OUT/pytest_full.log"
if ! grep -q " passed" "$OUT/pytest_full.log" || grep -q "failed\|error" "$OUT/pytest_full.log"; then
  echo "== serial re-run of failures" | tee -a "$OUT/pytest_full.log"
  timeout 900 python -m pytest tests -m gpu -q -x --lf 2>&1 | tail -15 | tee -a "$OUT/pytest_full.log"
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a "$OUT/pytest_full.log"
bash scripts/r02_visit3.sh "$TAG"
