#!/bin/bash
# Round-2 visit: pyrdown_f32 with a destination pixel pair per thread; parity + timing against one pixel per thread.
set -u
TAG=${1:-r02zt}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== parity" | tee "$OUT/log.txt"
timeout 600 python -m pytest tests/test_pyramid_morph_gpu.py -m gpu -q -x -k "pyr" 2>&1 | tail -3 | tee -a "$OUT/log.txt"
run() { wl=$1; shift; echo "== $wl $*" | tee -a "$OUT/log.txt"; env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 2>&1 | grep '^{' | python -c 'import json,sys
for l in sys.stdin:
    j=json.loads(l); r=j["roofline"]; print("   %-50s %8.3f ms/step  frac %.3f  launch %.3f ms" % (j["config"]["workload"], j["ms_per_step"], r["frac"], r["mean_launch_ms"]))' | tee -a "$OUT/log.txt"; }
run pyrdown_f32_4k KH_X=0

