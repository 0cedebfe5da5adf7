python -m pytest tests/test_geom_gpu.py -q -x -n 4 2>&1 | tail -2
echo "== one pixel per lane (resize_rows=0)"; python scripts/diag/resize_f32_up2_r06.py resize_rows=0 2>&1 | grep " c1"; python scripts/diag/warp_channels_r06.py resize_rows=0 2>&1 | grep -E "resize f32 .* c1"
echo "== four pixels per lane"; python scripts/diag/resize_f32_up2_r06.py 2>&1 | grep " c1"; python scripts/diag/warp_channels_r06.py 2>&1 | grep -E "resize f32 .* c1"
