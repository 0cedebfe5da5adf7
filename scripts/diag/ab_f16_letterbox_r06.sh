python -m pytest tests/test_preprocess_gpu.py tests/test_bench_workloads_gpu.py -q -x -n 4 2>&1 | tail -3
python scripts/diag/f16_letterbox_r06.py 2>&1 | tail -8
