python -m pytest tests/test_pyramid_morph_gpu.py tests/test_workspace_cache_gpu.py -q -x -n 4 2>&1 | tail -3
python scripts/diag/morph_sizes_r06.py 2>&1 | grep " box"
