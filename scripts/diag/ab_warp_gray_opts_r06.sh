python -m pytest tests/test_u8_gpu.py -q -x -n 4 2>&1 | tail -2
for o in "" "warp_u8_rows=16"; do echo "== $o"; python scripts/diag/warp_channels_r06.py $o 2>&1 | grep -E "(warp_affine|warp_perspective|remap) u8"; done
