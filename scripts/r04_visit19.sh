#!/bin/bash
# Round-4 visit 19: row-pair vertical pass of the separable u8 resize against the previous library + kernel trace of both.
set -u
TAG=${1:-r04zw}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$(pwd)
timeout 300 python -m pytest tests/test_resize_u8_gpu.py -q -x 2>&1 | tail -1 | tee "$OUT/pytest.log"
bash scripts/r04_ab_prev.sh $TAG resize_u8_224 none 3
cd /tmp
for which in prev new; do
  if [ $which = prev ]; then export KORNIA_HIP_LIB=$REPO/kornia-rs_amd/lib/libkornia_hip_prev.so; else unset KORNIA_HIP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/kt_$which" -o kt -- python "$REPO/bench.py" --workload resize_u8_224 --no-cpu-baseline --also none > /dev/null 2>&1
  f=$(find "$REPO/$OUT/kt_$which" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$REPO/$OUT/resize_u8_${which}_kernel_stats.csv" && grep "sep_" "$f" | cut -c1-160
  rm -rf "$REPO/$OUT/kt_$which"
done
