python -m pytest tests/test_preprocess_gpu.py -q -x -n 4 2>&1 | tail -3
python scripts/diag/_f16_tmp.py 2>&1 | tail -2
