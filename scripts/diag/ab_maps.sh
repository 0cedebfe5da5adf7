for i in 1 2 3; do for v in plain wt; do for w in gray_u8_1080p gray_f32_1080p hsv_f32_1080p bgr_u8_1080p; do
  KORNIA_HIP_LIB=$PWD//tmp/kh_ab/libkornia_hip_$v.so python bench.py --workload $w --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', j['config']['workload'], j['ms_per_step'], j['roofline']['frac'])"
done; done; done
