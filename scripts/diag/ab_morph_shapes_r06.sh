python -m pytest tests/test_pyramid_morph_gpu.py -q -x -n 4 2>&1 | tail -3
echo "== tile kernel (morph_roll=2)"; python scripts/diag/morph_sizes_r06.py morph_roll=2 2>&1 | grep -E " (3x3|5x5|7x7) " | grep -E "cross|ellipse"
echo "== rolling kernels"; python scripts/diag/morph_sizes_r06.py 2>&1 | grep -E " (3x3|5x5|7x7) "
