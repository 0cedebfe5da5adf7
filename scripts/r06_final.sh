#!/bin/bash
# Round-6 reference visit: the whole device suite (serial, as the driver runs it), smoke, the default bench line, the opt-in
# workloads, one kernel-trace profile of the default run (rocprofv3's own CSV) and the HBM PMC passes bench.py replays.
set -u
TAG=${1:-r06z}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
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
timeout 1200 python bench.py --workload fused_rgb_640 --no-cpu-baseline --also resize_normalize_f32_224,resize_u8_224,resize_norm_chw_224,pyrdown_u8_4k,pyrup_u8_4k,pyrdown_f32_4k,pyrup_f32_4k,dilate_u8_4k,nv12_chw_640_lanczos,spatial_gradient_1080p,box_blur_fast_1080p,bgr_u8_1080p,nv12_chw_640_f16,nv12_chw_608_f16,yuyv_chw_640_f16 2>&1 | grep '^{' | tee -a "$OUT/bench.log" | python scripts/bench_table.py | tee -a "$OUT/bench_table.txt"
cp gpurun_out/bench_full.json "$OUT/bench_full_optin.json" 2>/dev/null
echo "== rocprofv3 kernel trace of the default run"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/prof_default" -o kt -- python "$REPO/bench.py" --no-cpu-baseline > "$REPO/$OUT/prof_default.log" 2>&1
cd "$REPO"
f=$(find "$OUT/prof_default" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/default_kernel_stats.csv" && head -30 "$f" | cut -c1-190
rm -rf "$OUT/prof_default"
echo "== rocprofv3 kernel trace of the HEADLINE alone (the default run also launches preprocess_nv12_identity on 64-frame batches in its"
echo "   H2D row, which pulls that kernel's average in default_kernel_stats.csv down: this is the file to check roofline.mean_launch_ms against)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/prof_headline" -o kt -- python "$REPO/bench.py" --no-cpu-baseline --also none > "$REPO/$OUT/prof_headline.log" 2>&1
cd "$REPO"
f=$(find "$OUT/prof_headline" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/headline_kernel_stats.csv" && head -5 "$f" | cut -c1-190
grep '^{' "$OUT/prof_headline.log" | python scripts/bench_table.py | head -1
rm -rf "$OUT/prof_headline"
echo "== PMC passes (FETCH_SIZE, WRITE_SIZE) of the default run"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d "$REPO/$OUT/pmc_default_$c" -o pmc -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$REPO/$OUT/pmc_default_$c.log" 2>&1
  f=$(find "$REPO/$OUT/pmc_default_$c" -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" "$REPO/$OUT/default_pmc_${c}_counter_collection.csv"
  rm -rf "$REPO/$OUT/pmc_default_$c"
done
cd "$REPO"
python scripts/pmc_to_traffic.py "$OUT/default_pmc_FETCH_SIZE_counter_collection.csv" "$OUT/default_pmc_WRITE_SIZE_counter_collection.csv" "profiles/${TAG}_default_pmc_{FETCH,WRITE}_SIZE_counter_collection.csv: rocprofv3 --pmc passes of the default bench run (scripts/r06_final.sh)" | tee "$OUT/traffic.txt"
cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json"
if [ "${COUNTERS:-0}" = "1" ]; then
  echo "== SQ / TA counters of the kernels re-designed in rounds 4 and 6"
  for wl in resize_224 undistort_warp_4k; do
    bash scripts/diag/pmc_cmd.sh $TAG/after_$wl "python $REPO/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --also none" \
      "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
      "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum" 2>&1 | tail -4 | cut -c1-400
    cp "$OUT/after_$wl/pmc_table.txt" "$OUT/after_${wl}_counters.csv" 2>/dev/null
  done
fi
du -sh "$OUT"; ls "$OUT" | head -40
