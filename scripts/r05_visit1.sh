#!/bin/bash
# Round-5 first visit: the whole device suite on this round's first commit (advisor fixes: device affinity, opt-in zero-copy
# uploads, per-thread test options), smoke, the default bench line and one kernel-trace profile of it.
set -u
TAG=${1:-r05a}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
( time timeout 1500 python -m pytest tests -m gpu -q ) > "$OUT/pytest_full.log" 2>&1
tail -5 "$OUT/pytest_full.log"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a "$OUT/pytest_full.log"
echo "== default bench" | tee "$OUT/bench_table.txt"
( time timeout 900 python bench.py ) > "$OUT/bench_raw.log" 2>&1
grep '^{' "$OUT/bench_raw.log" > "$OUT/bench.log"; python scripts/bench_table.py < "$OUT/bench.log" | tee -a "$OUT/bench_table.txt"
grep "^real" "$OUT/bench_raw.log" | tee -a "$OUT/bench_table.txt"
cp gpurun_out/bench_full.json "$OUT/bench_full.json" 2>/dev/null
echo "== rocprofv3 kernel trace of the default run"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/prof_default" -o kt -- python "$REPO/bench.py" --no-cpu-baseline > "$REPO/$OUT/prof_default.log" 2>&1
cd "$REPO"
f=$(find "$OUT/prof_default" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/default_kernel_stats.csv" && head -12 "$f" | cut -c1-190
rm -rf "$OUT/prof_default"
du -sh "$OUT"
