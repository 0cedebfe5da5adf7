#!/bin/bash
# Round-2 GPU visit 1 (dev tooling): north-star micro-benchmarks + queued A/Bs, the two-runtime diagnosis, the new device
# tests, the default bench line (with its `also` array) and a kernel-trace profile of the HEAD binary.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/r02_visit1.sh r02a'
set -u
TAG=${1:-r02a}
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

echo "== 1. nv12_r02 micro-benchmark" | tee "$OUT/ubench.txt"
timeout 300 scripts/ubench/bin/nv12_r02 1024 5 2>&1 | tee -a "$OUT/ubench.txt"

echo "== 2. runtimes diagnosis" | tee "$OUT/runtimes.log"
timeout 900 python scripts/diag_runtimes.py 2>&1 | tee -a "$OUT/runtimes.log"

echo "== 3. new device tests" | tee "$OUT/pytest.log"
timeout 900 python -m pytest tests/test_sharding_gpu.py tests/test_unified_gpu.py tests/test_host_api_gpu.py tests/test_cpp_mirror.py tests/test_abi.py \
    "tests/test_bench_workloads_gpu.py::test_colour_map_workloads_1080p" tests/test_color_gpu.py -m gpu -x -q --timeout 600 2>&1 | tail -15 | tee -a "$OUT/pytest.log"

echo "== 4. default bench (headline + also)" | tee "$OUT/bench.log"
timeout 900 python bench.py 2>&1 | grep '^{' | tee -a "$OUT/bench.log" | line | tee -a "$OUT/bench_table.txt"

echo "== 5. north star: default block order vs XCD-per-frame order, interleaved x3" | tee "$OUT/ab.log"
for i in 1 2 3; do
  echo " default" | tee -a "$OUT/ab.log"; timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --also none 2>&1 | line | tee -a "$OUT/ab.log"
  echo " xcd-frames" | tee -a "$OUT/ab.log"; KH_NV12_XCD_FRAMES=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --also none 2>&1 | line | tee -a "$OUT/ab.log"
done
echo "== 6. frame stride / base alignment sweep" | tee -a "$OUT/ab.log"
timeout 600 python scripts/ab_north_star_stride.py --rounds 5 2>&1 | tee -a "$OUT/ab.log"

echo "== 7. in-process sharded mode on this box's device(s)" | tee -a "$OUT/bench.log"
timeout 300 python bench.py --in-process --gpus 1 --steps 20 --warmup 5 2>&1 | grep '^{' | tee -a "$OUT/bench.log" | line
timeout 300 python bench.py --in-process --gpus 2 --batch 512 --steps 20 --warmup 5 2>&1 | grep '^{' | tee -a "$OUT/bench.log" | line

echo "== 8. rocprofv3 kernel trace of the headline workload"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_nv12_chw" -o kt -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --also none > "$REPO/$OUT/prof_nv12_chw.log" 2>&1
cd "$REPO"
db=$(find "$OUT/prof_nv12_chw" -name '*.db' | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py "$db" > "$OUT/nv12_chw_kernel_stats.csv" && head -4 "$OUT/nv12_chw_kernel_stats.csv"
find "$OUT" -name '*.db' -size +20M -delete
du -sh "$OUT"
