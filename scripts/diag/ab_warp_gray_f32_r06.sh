python -m pytest tests/test_geom_gpu.py -q -x -n 4 2>&1 | tail -2
python scripts/diag/warp_channels_r06.py 2>&1 | grep -E "f32 .*c1"
