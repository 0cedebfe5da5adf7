python -m pytest tests/test_resize_u8_gpu.py tests/test_pyramid_morph_gpu.py -q -x -n 4 2>&1 | tail -2
echo "== per-pixel kernels (resize_u8_px=2)"; python scripts/diag/resize_up2_r06.py resize_u8_px=2 2>&1 | grep "4K"
echo "== rolling kernels"; python scripts/diag/resize_up2_r06.py 2>&1 | grep "4K"
