#!/bin/bash
# Round-2 GPU visit: device tests touched since the last visit + the default bench line + kernel trace of the new north-star kernel.
set -u
TAG=${1:-r02h}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$(pwd)
line() { grep '^{' | python -c 'import json,sys
for l in sys.stdin:
    j=json.loads(l); r=j.get("roofline",{})
    print("   %-52s %8.3f ms/step  %10.0f %s  frac %.3f" % (j["config"]["workload"], j["ms_per_step"], j["value"], j["unit"], r.get("frac") or 0))
    for a in j.get("also", []):
        r=a["roofline"]; print("     also %-47s %8.3f ms/step  %10.0f Mpx/s  frac %.3f  launch %.3f ms  cpu %s" % (a["config"]["workload"], a["ms_per_step"], a["value"], r["frac"], r["mean_launch_ms"], a.get("cpu_baseline",{}).get("value")))'; }
echo "== device tests" | tee "$OUT/pytest.log"
timeout 1200 python -m pytest ${TESTS:-tests/test_preprocess_gpu.py tests/test_workspace_cache_gpu.py tests/test_filter_gpu.py tests/test_resize_u8_gpu.py tests/test_filter_extra_gpu.py tests/test_u8_gpu.py tests/test_host_api_gpu.py tests/test_sharding_gpu.py tests/test_unified_gpu.py tests/test_zz_host_extras_gpu.py tests/test_bench_workloads_gpu.py} -m gpu -x -q --timeout 900 -n 4 2>&1 | tail -15 | tee -a "$OUT/pytest.log"
echo "rc=${PIPESTATUS[0]}" | tee -a "$OUT/pytest.log"
echo "== smoke" | tee -a "$OUT/pytest.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a "$OUT/pytest.log"
echo "== default bench" | tee "$OUT/bench.log"
timeout 900 python bench.py 2>&1 | grep '^{' | tee -a "$OUT/bench.log" | line | tee "$OUT/bench_table.txt"
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --also none 2>&1 | grep '^{' | tee -a "$OUT/bench.log" | line | tee -a "$OUT/bench_table.txt"; done
echo "== rocprofv3 kernel trace"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_nv12_chw" -o kt -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --also none > "$REPO/$OUT/prof_nv12_chw.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d "$REPO/$OUT/pmc_nv12_chw_$c" -o pmc -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --also none > "$REPO/$OUT/pmc_nv12_chw_$c.log" 2>&1
done
cd "$REPO"
db=$(find "$OUT/prof_nv12_chw" -name '*.db' | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py "$db" > "$OUT/nv12_chw_kernel_stats.csv" && head -4 "$OUT/nv12_chw_kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  db=$(find "$OUT/pmc_nv12_chw_$c" -name '*.db' | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py "$db" > "$OUT/nv12_chw_pmc_$c.csv" && tail -3 "$OUT/nv12_chw_pmc_$c.csv"
done
find "$OUT" -name '*.db' -delete
du -sh "$OUT"
