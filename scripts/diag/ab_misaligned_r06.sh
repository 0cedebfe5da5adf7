python -m pytest tests/test_pyramid_morph_gpu.py tests/test_resize_u8_gpu.py -q -x -n 4 2>&1 | tail -2
echo "== streaming stores always (row_stores=0)"; python scripts/diag/misaligned_rows_r06.py row_stores=0 2>&1 | grep -E "^# 1|: "
echo "== write-back stores always (row_stores=1)"; python scripts/diag/misaligned_rows_r06.py row_stores=1 2>&1 | grep -E "^# 1|: "
echo "== the rule"; python scripts/diag/misaligned_rows_r06.py 2>&1 | grep -E "^# 1|: "
