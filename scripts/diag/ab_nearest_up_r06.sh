python -m pytest tests/test_resize_u8_gpu.py -q -x -n 4 2>&1 | tail -2
echo "== quad kernel (resize_u8_px=2)"; python scripts/diag/resize_nearest_up_r06.py resize_u8_px=2 2>&1 | grep nearest
echo "== column selectors once per lane"; python scripts/diag/resize_nearest_up_r06.py 2>&1 | grep nearest
