#!/bin/bash
# Round-4 reference visit: the whole device suite (serial, as the driver runs it), smoke, the default bench line, the opt-in
# workloads, one kernel-trace profile of the default run (rocprofv3's own CSV) and the HBM PMC passes bench.py replays.
set -u
TAG=${1:-r04z}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  ( time timeout 1800 python -m pytest tests -m gpu -q ) > "$OUT/pytest_full.log" 2>&1
  tail -5 "$OUT/pytest_full.log"
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a "$OUT/pytest_full.log"
fi
echo "== default bench" | tee "$OUT/bench_table.txt"
( time timeout 900 python bench.py ) > "$OUT/bench_raw.log" 2>&1
grep '^{' "$OUT/bench_raw.log" > "$OUT/bench.log"; python scripts/bench_table.py < "$OUT/bench.log" | tee -a "$OUT/bench_table.txt"
grep "^real" "$OUT/bench_raw.log" | tee -a "$OUT/bench_table.txt"
cp gpurun_out/bench_full.json "$OUT/bench_full.json" 2>/dev/null
echo "== opt-in workloads" | tee -a "$OUT/bench_table.txt"
timeout 1200 python bench.py --workload fused_rgb_640 --no-cpu-baseline --also nv12_h2d_preprocess_pageable,resize_normalize_f32_224,resize_u8_224,resize_norm_chw_224,pyrdown_u8_4k,pyrup_u8_4k,pyrdown_f32_4k,pyrup_f32_4k,dilate_u8_4k,nv12_chw_640_lanczos,spatial_gradient_1080p,box_blur_fast_1080p,bgr_u8_1080p 2>&1 | grep '^{' | tee -a "$OUT/bench.log" | python scripts/bench_table.py | tee -a "$OUT/bench_table.txt"
cp gpurun_out/bench_full.json "$OUT/bench_full_optin.json" 2>/dev/null
echo "== rocprofv3 kernel trace of the default run"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/prof_default" -o kt -- python "$REPO/bench.py" --no-cpu-baseline > "$REPO/$OUT/prof_default.log" 2>&1
cd "$REPO"
f=$(find "$OUT/prof_default" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/default_kernel_stats.csv" && head -30 "$f" | cut -c1-190
rm -rf "$OUT/prof_default"
echo "== PMC passes (FETCH_SIZE, WRITE_SIZE) of the default run"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d "$REPO/$OUT/pmc_default_$c" -o pmc -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$REPO/$OUT/pmc_default_$c.log" 2>&1
  f=$(find "$REPO/$OUT/pmc_default_$c" -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" "$REPO/$OUT/default_pmc_${c}_counter_collection.csv"
  rm -rf "$REPO/$OUT/pmc_default_$c"
done
cd "$REPO"
python scripts/pmc_to_traffic.py "$OUT/default_pmc_FETCH_SIZE_counter_collection.csv" "$OUT/default_pmc_WRITE_SIZE_counter_collection.csv" "profiles/${TAG}_default_pmc_{FETCH,WRITE}_SIZE_counter_collection.csv: rocprofv3 --pmc passes of the default bench run (scripts/r04_final.sh)" | tee "$OUT/traffic.txt"
cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json"
du -sh "$OUT"; ls -la "$OUT" | head -30
