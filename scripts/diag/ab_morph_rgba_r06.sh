python -m pytest tests/test_pyramid_morph_gpu.py -q -x -n 4 2>&1 | tail -2
echo "== tile kernel (morph_roll=2)"; python scripts/diag/morph_sizes_r06.py morph_roll=2 2>&1 | grep -E "dilate c4"
echo "== rolling kernel"; python scripts/diag/morph_sizes_r06.py 2>&1 | grep -E "dilate c4"
