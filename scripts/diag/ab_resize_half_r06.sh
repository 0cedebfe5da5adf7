python -m pytest tests/test_resize_u8_gpu.py -q -x -n 4 2>&1 | tail -2
echo "== generic Q14 quads (resize_u8_px=2)"; python scripts/diag/warp_channels_r06.py resize_u8_px=2 2>&1 | grep -E "resize_fast u8 4K -> 1080p bilinear"
echo "== exact-2x box"; python scripts/diag/warp_channels_r06.py 2>&1 | grep -E "resize_fast u8 4K -> 1080p (bilinear|nearest)"
