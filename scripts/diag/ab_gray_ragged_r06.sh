python -m pytest tests/test_pyramid_morph_gpu.py -q -x -n 4 2>&1 | tail -2
python scripts/diag/misaligned_rows_r06.py 2>&1 | grep -E "^# 1|dilate|pyr.* u8 .*c1"
