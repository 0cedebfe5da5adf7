#!/bin/bash
# Same-box A/B of the library before (scripts/ab_prev = round 5's tree, untracked) and after the pointer-list change: three interleaved rounds.
OUT=gpurun_out/${AB_OUT:-r06c_ab_ptrlist.txt}
mkdir -p gpurun_out; : > $OUT
W="--workload undistort_warp_4k --no-cpu-baseline --also gaussian_4k,box_blur_4k,sobel_4k,resize_224,resize_bicubic_540,warp_affine_f32_1080p,nv12_chw,nv12_chw_640,nv12_chw_608,yuyv_chw_640"
for r in 1 2 3; do
  echo "== round $r: NEW (pointer-list kernels)" >> $OUT
  python bench.py $W 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); [print('   %-40s %9.4f ms' % (r[0], r[2])) for r in j['summary']]" >> $OUT
  echo "== round $r: PREV (round 5 library)" >> $OUT
  (cd scripts/ab_prev && python bench.py $W 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); [print('   %-40s %9.4f ms' % (r[0], r[2])) for r in j['summary']]") >> $OUT
done
cat $OUT
