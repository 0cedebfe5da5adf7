#!/bin/bash
# Round-3 reference visit: the whole device suite (serial, as the driver runs it: four xdist workers with a 256-thread OpenMP team each were SLOWER; failures re-run), smoke, the default bench line
# (headline + also + summary), the opt-in workloads, one kernel-trace profile of the default run and the HBM PMC passes bench.py replays.
set -u
TAG=${1:-r03z}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  ( time timeout 1500 python -m pytest tests -m gpu -q ) > "$OUT/pytest_full.log" 2>&1
  tail -5 "$OUT/pytest_full.log"
  if ! grep -q " passed" "$OUT/pytest_full.log" || grep -q "failed\|error" "$OUT/pytest_full.log"; then
    echo "== serial re-run of failures" | tee -a "$OUT/pytest_full.log"
    timeout 900 python -m pytest tests -m gpu -q -x --lf 2>&1 | tail -15 | tee -a "$OUT/pytest_full.log"
  fi
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a "$OUT/pytest_full.log"
fi
echo "== default bench" | tee "$OUT/bench_table.txt"
( time timeout 900 python bench.py ) > "$OUT/bench_raw.log" 2>&1
grep '^{' "$OUT/bench_raw.log" > "$OUT/bench.log"; python scripts/bench_table.py < "$OUT/bench.log" | tee -a "$OUT/bench_table.txt"
grep "^real" "$OUT/bench_raw.log" | tee -a "$OUT/bench_table.txt"
cp gpurun_out/bench_full.json "$OUT/bench_full.json" 2>/dev/null
echo "== opt-in workloads" | tee -a "$OUT/bench_table.txt"
timeout 1200 python bench.py --workload fused_rgb_640 --no-cpu-baseline --also resize_normalize_f32_224,resize_u8_224,resize_norm_chw_224,pyrdown_u8_4k,pyrup_u8_4k,pyrdown_f32_4k,pyrup_f32_4k,dilate_u8_4k,nv12_chw_640_lanczos,lab_from_rgb_4k,spatial_gradient_1080p,box_blur_fast_1080p,median5_u8_1080p,bilateral_1080p,bgr_u8_1080p 2>&1 | grep '^{' | tee -a "$OUT/bench.log" | python scripts/bench_table.py | tee -a "$OUT/bench_table.txt"
echo "== rocprofv3 kernel trace of the default run"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_default" -o kt -- python "$REPO/bench.py" --no-cpu-baseline > "$REPO/$OUT/prof_default.log" 2>&1
cd "$REPO"
db=$(find "$OUT/prof_default" -name '*.db' | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py "$db" | grep -v "rocclr" > "$OUT/default_kernel_stats.csv" && head -24 "$OUT/default_kernel_stats.csv" | cut -c1-200
echo "== PMC passes (FETCH_SIZE, WRITE_SIZE) of the default run"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d "$REPO/$OUT/pmc_default_$c" -o pmc -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$REPO/$OUT/pmc_default_$c.log" 2>&1
done
cd "$REPO"
for c in FETCH_SIZE WRITE_SIZE; do
  db=$(find "$OUT/pmc_default_$c" -name '*.db' | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py "$db" | grep -v rocclr | sed -n '/counter,mean/,$p' > "$OUT/default_pmc_$c.csv" && cat "$OUT/default_pmc_$c.csv" | cut -c1-60,140-
done
find "$OUT" -name '*.db' -delete
if [ "${COUNTERS:-1}" = "1" ]; then
  echo "== SQ counters of the round-3 rolling RGB8 kernels (VERDICT r02 item 4)"
  for wl in gaussian_u8_4k dilate_u8_4k pyrdown_u8_4k pyrup_u8_4k; do
    bash scripts/diag/pmc_workload.sh $wl $TAG/sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE" 2>&1 | tail -12 | tee -a "$OUT/sq_counters.txt"
  done
fi
du -sh "$OUT"
