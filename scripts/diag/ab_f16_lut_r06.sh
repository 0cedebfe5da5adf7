python -m pytest tests/test_preprocess_gpu.py tests/test_bench_workloads_gpu.py tests/test_dev_options_gpu.py -q -x -n 4 2>&1 | tail -2
for i in 1 2 3; do
  echo "== arithmetic kernel (pre_f16_lut=0)"; python scripts/diag/f16_identity_r06.py pre_f16_lut=0 2>&1 | grep "f16=True"
  echo "== table kernel"; python scripts/diag/f16_identity_r06.py 2>&1 | grep "f16=True"
done
