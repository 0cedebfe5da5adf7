#!/bin/bash
# One GPU-box visit: parity tests, smoke, headline bench, rocprofv3 kernel-trace summary.
# Usage (from the repo root, via gpurun):  bash scripts/gpu_check.sh [tag]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$(pwd)

echo "== pytest -m gpu" | tee "$OUT/pytest.log"
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -25 | tee -a "$OUT/pytest.log"

echo "== smoke" | tee "$OUT/smoke.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee -a "$OUT/smoke.log"

echo "== bench" | tee "$OUT/bench.log"
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 | tee -a "$OUT/bench.log"
timeout 600 python bench.py --steps 20 --warmup 5 --workload nv12_chw_640 --no-cpu-baseline 2>&1 | tail -2 | tee -a "$OUT/bench.log"

echo "== rocprofv3 kernel trace"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof" -o nv12 -- \
    python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$REPO/$OUT/prof_bench.log" 2>&1
cd "$REPO"
find "$OUT/prof" -name '*kernel_stats*' | head -3
for f in $(find "$OUT/prof" -name '*kernel_stats*.csv' | head -1); do head -8 "$f"; done
# keep the merged-back payload small: drop the raw per-dispatch trace, keep the stats
find "$OUT/prof" -name '*kernel_trace*.csv' -size +8M -delete
ls -la "$OUT" "$OUT/prof" 2>/dev/null | head -30
