#!/bin/bash
# Dev tooling, one GPU-box visit (~12 min): the A/B measurements queued while the GPU budget was spent.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/ab_next_round.sh r02ab'
# Everything lands in gpurun_out/<tag>/ab.log; copy what is worth keeping into profiles/.
set -u
TAG=${1:-r02ab}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
LOG=$OUT/ab.log
line() { grep '^{' | python -c 'import json,sys
for l in sys.stdin:
    j=json.loads(l); r=j.get("roofline",{})
    print("   %-44s %8.3f ms/step  %10.0f %s  frac %.3f" % (j["config"]["workload"], j["ms_per_step"], j["value"], j["unit"], r.get("frac") or 0))'; }

echo "== 1. north star: default block order vs XCD-per-frame order (KH_NV12_XCD_FRAMES), interleaved x3" | tee -a "$LOG"
for i in 1 2 3; do
  echo " default" | tee -a "$LOG"; timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | line | tee -a "$LOG"
  echo " xcd-frames" | tee -a "$LOG"; KH_NV12_XCD_FRAMES=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | line | tee -a "$LOG"
done

echo "== 2. north star: frame stride / base alignment sweep (production kernel)" | tee -a "$LOG"
timeout 600 python scripts/ab_north_star_stride.py --rounds 5 2>&1 | tee -a "$LOG"

echo "== 3. box_blur_fast: LDS-staged rows vs direct (KH_HFILTER_DIRECT)" | tee -a "$LOG"
for i in 1 2; do
  echo " lds" | tee -a "$LOG"; timeout 300 python bench.py --workload box_blur_fast_1080p --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | line | tee -a "$LOG"
  echo " direct" | tee -a "$LOG"; KH_HFILTER_DIRECT=1 timeout 300 python bench.py --workload box_blur_fast_1080p --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | line | tee -a "$LOG"
done

echo "== 3b. median 5x5: dword window loads vs byte loads (KH_MEDIAN_BYTES)" | tee -a "$LOG"
for i in 1 2; do
  echo " wide" | tee -a "$LOG"; timeout 300 python bench.py --workload median5_u8_1080p --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | line | tee -a "$LOG"
  echo " bytes" | tee -a "$LOG"; KH_MEDIAN_BYTES=1 timeout 300 python bench.py --workload median5_u8_1080p --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | line | tee -a "$LOG"
done

echo "== 3c. spatial gradient: four elements per thread vs one (KH_GRAD_SCALAR)" | tee -a "$LOG"
for i in 1 2; do
  echo " x4" | tee -a "$LOG"; timeout 300 python bench.py --workload spatial_gradient_1080p --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | line | tee -a "$LOG"
  echo " scalar" | tee -a "$LOG"; KH_GRAD_SCALAR=1 timeout 300 python bench.py --workload spatial_gradient_1080p --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | line | tee -a "$LOG"
done

echo "== 4. first numbers for the workloads added without GPU time" | tee -a "$LOG"
for wl in spatial_gradient_1080p median5_u8_1080p bilateral_1080p resize_normalize_f32_224 resize_u8_224 resize_norm_chw_224 pyrdown_u8_4k dilate_u8_4k lab_from_rgb_4k; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 2>&1 | grep '^{' | tee -a "$OUT/bench_new.log" | line | tee -a "$LOG"
done
