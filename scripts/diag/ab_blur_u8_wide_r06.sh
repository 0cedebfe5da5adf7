python -m pytest tests/test_u8_gpu.py -q -x -n 4 2>&1 | tail -3
python scripts/diag/blur_sizes_r06.py 2>&1 | grep "u8"
