python -m pytest tests/test_resize_u8_gpu.py tests/test_fuzz_gpu.py tests/test_dev_options_gpu.py -q -x -n 4 2>&1 | tail -3
python scripts/diag/resize_u8_modes_r06.py 2>&1 | grep "nearest\|bilinear"
