python -m pytest tests/test_pyramid_morph_gpu.py -q -x -n 4 2>&1 | tail -3
echo "== tile kernels (pyr_roll=0)"; python scripts/diag/pyr_channels_r06.py pyr_roll=0 2>&1 | grep -E "^pyr"
echo "== default"; python scripts/diag/pyr_channels_r06.py 2>&1 | grep -E "^pyr"
