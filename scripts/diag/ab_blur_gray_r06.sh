python -m pytest tests/test_u8_gpu.py -q -x -n 4 2>&1 | tail -2
echo "== interleaved kernel (u8_blur_rgb=0)"; python scripts/diag/blur_u8_gray_sizes_r06.py u8_blur_rgb=0 2>&1 | grep -E " c1 +(3|5|7|9|11|13|15)x"
echo "== gray rolling kernel"; python scripts/diag/blur_u8_gray_sizes_r06.py 2>&1 | grep -E " c1 +(3|5|7|9|11|13|15)x"
