set -u; export TMPDIR=/tmp; REPO=$(pwd); OUT=gpurun_out/r05z; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/prof_headline" -o kt -- python "$REPO/bench.py" --no-cpu-baseline --also none > "$REPO/$OUT/prof_headline.log" 2>&1
cd "$REPO"
f=$(find "$OUT/prof_headline" -name '*kernel_stats.csv' | head -1); cp "$f" "$OUT/headline_kernel_stats.csv"; cat "$f" | cut -c1-200
grep '^{' "$OUT/prof_headline.log" | python scripts/bench_table.py | head -3
rm -rf "$OUT/prof_headline"
