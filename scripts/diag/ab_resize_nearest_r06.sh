python -m pytest tests/test_resize_u8_gpu.py -q -x -n 4 2>&1 | tail -2
echo "== f64 column index (resize_u8_px=3)"; python scripts/diag/resize_up2_r06.py resize_u8_px=3 2>&1 | grep "nearest"; python scripts/diag/warp_channels_r06.py resize_u8_px=3 2>&1 | grep -E "resize_fast u8 .*nearest"
echo "== integer column index"; python scripts/diag/resize_up2_r06.py 2>&1 | grep "nearest"; python scripts/diag/warp_channels_r06.py 2>&1 | grep -E "resize_fast u8 .*nearest"
