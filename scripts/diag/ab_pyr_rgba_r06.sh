python -m pytest tests/test_pyramid_morph_gpu.py -q -x -n 4 2>&1 | tail -2
echo "== tile kernels (pyr_roll=0)"; U8_ONLY=1 python scripts/diag/pyr_channels_r06.py pyr_roll=0 2>&1 | grep -E "^pyr.* u8 "
echo "== default"; U8_ONLY=1 python scripts/diag/pyr_channels_r06.py 2>&1 | grep -E "^pyr.* u8 "
