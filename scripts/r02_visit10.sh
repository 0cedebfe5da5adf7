#!/bin/bash
# Round-2 closing check after the filter default flip: filter parity (three kernel paths), default bench line.
set -u
TAG=${1:-r02zl}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_filter_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee "$OUT/log.txt"
timeout 900 python bench.py 2>&1 | grep '^{' | tee "$OUT/bench.log" | python -c 'import json,sys
for l in sys.stdin:
    j=json.loads(l); r=j.get("roofline",{})
    print("%-56s %8.3f ms/step %10.0f Mpx/s  frac %.3f  traffic %s  cpu %s" % (j["config"]["workload"], j["ms_per_step"], j["value"], r.get("frac") or 0, r.get("traffic"), j.get("cpu_baseline",{}).get("value")))
    for a in j.get("also", []):
        r=a["roofline"]; print("  also %-51s %8.3f ms/step %10.0f Mpx/s  frac %.3f  kernel %s" % (a["config"]["workload"], a["ms_per_step"], a["value"], r["frac"], r["kernel"][:40]))' | tee -a "$OUT/log.txt"
