python -m pytest tests/test_filter_gpu.py tests/test_dev_options_gpu.py -q -x -n 4 2>&1 | tail -3
echo "== streaming stores always (row_stores=0)"; python scripts/diag/filter_widths_r06.py row_stores=0 | grep gaussian
echo "== write-back stores always (row_stores=1)"; python scripts/diag/filter_widths_r06.py row_stores=1 | grep gaussian
echo "== the rule (write-back where row bytes % 128 != 0)"; python scripts/diag/filter_widths_r06.py | grep gaussian
