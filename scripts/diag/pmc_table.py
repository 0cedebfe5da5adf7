#!/usr/bin/env python3
"""Per-kernel mean of every counter in rocprofv3 `--output-format csv` counter_collection files (one row per dispatch x counter).
    python scripts/diag/pmc_table.py pass1_counter_collection.csv [pass2...]  ->  kernel x counter table (mean per dispatch)"""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))
            if "rocclr" in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
print("kernel," + ",".join(counters) + ",dispatches")
for k in sorted(acc):
    n = max(len(v) for v in acc[k].values())
    print(k[:60] + "," + ",".join(f"{sum(acc[k][c]) / len(acc[k][c]):.0f}" if c in acc[k] else "" for c in counters) + f",{n}")
