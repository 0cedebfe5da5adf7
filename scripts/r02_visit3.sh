#!/bin/bash
# Round-2 visit: numbers for every bench workload (the default line + the opt-in ones through --also), one kernel-trace profile of
# the default run (all `also` kernels in one CSV) and the HBM PMC passes for the workloads whose traffic bench.py replays.
set -u
TAG=${1:-r02r}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
line() { python -c 'import json,sys
for l in sys.stdin:
    if not l.startswith("{"): continue
    j=json.loads(l); r=j.get("roofline",{})
    print("%-56s %8.3f ms/step %10.0f Mpx/s  frac %.3f  launch %.3f ms  cpu %s" % (j["config"]["workload"], j["ms_per_step"], j["value"], r.get("frac") or 0, r.get("mean_launch_ms") or 0, j.get("cpu_baseline",{}).get("value")))
    for a in j.get("also", []):
        r=a["roofline"]; print("  also %-51s %8.3f ms/step %10.0f Mpx/s  frac %.3f  launch %.3f ms  cpu %s" % (a["config"]["workload"], a["ms_per_step"], a["value"], r["frac"], r["mean_launch_ms"], a.get("cpu_baseline",{}).get("value")))'; }
echo "== default bench" | tee "$OUT/bench_table.txt"
timeout 900 python bench.py 2>&1 | grep '^{' | tee "$OUT/bench.log" | line | tee -a "$OUT/bench_table.txt"
echo "== opt-in workloads" | tee -a "$OUT/bench_table.txt"
timeout 1200 python bench.py --no-cpu-baseline --also resize_normalize_f32_224,fused_rgb_640,resize_u8_224,resize_norm_chw_224,pyrdown_u8_4k,pyrup_u8_4k,pyrdown_f32_4k,pyrup_f32_4k,dilate_u8_4k,nv12_chw_640_lanczos,lab_from_rgb_4k,spatial_gradient_1080p,box_blur_fast_1080p,median5_u8_1080p,bilateral_1080p,bgr_u8_1080p 2>&1 | grep '^{' | tee -a "$OUT/bench.log" | line | tee -a "$OUT/bench_table.txt"
echo "== rocprofv3 kernel trace of the default run"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_default" -o kt -- python "$REPO/bench.py" --no-cpu-baseline > "$REPO/$OUT/prof_default.log" 2>&1
cd "$REPO"
db=$(find "$OUT/prof_default" -name '*.db' | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py "$db" | grep -v "rocclr" > "$OUT/default_kernel_stats.csv" && head -16 "$OUT/default_kernel_stats.csv" | cut -c1-200
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
du -sh "$OUT"
