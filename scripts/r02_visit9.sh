#!/bin/bash
# Round-2 visit: pyrup by source pixel (f32) / source pixel pair (u8): parity + timing against the per-destination-pixel kernels.
set -u
TAG=${1:-r02zi}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== parity" | tee "$OUT/log.txt"
timeout 900 python -m pytest tests/test_pyramid_morph_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee -a "$OUT/log.txt"
run() { wl=$1; shift; echo "== $wl $*" | tee -a "$OUT/log.txt"; env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 2>&1 | grep '^{' | python -c 'import json,sys
for l in sys.stdin:
    j=json.loads(l); r=j["roofline"]; print("   %-50s %8.3f ms/step  frac %.3f  launch %.3f ms" % (j["config"]["workload"], j["ms_per_step"], r["frac"], r["mean_launch_ms"]))' | tee -a "$OUT/log.txt"; }
run pyrup_u8_4k KH_X=0
run pyrup_f32_4k KH_X=0
run pyrdown_f32_4k KH_X=0
