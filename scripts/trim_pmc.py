#!/usr/bin/env python3
"""Trim a rocprofv3 counter_collection.csv to the <= 16 largest-grid dispatches of every kernel (what scripts/pmc_to_traffic.py averages),
so that the file fits profiles/:   python scripts/trim_pmc.py in.csv out.csv"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1], newline="")))
by = defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append(r)
keep = []
for k, v in by.items():
    if "rocclr" in k:
        continue
    g = max(int(r.get("Grid_Size") or 0) for r in v)
    keep += [r for r in v if int(r.get("Grid_Size") or 0) == g][:16]
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(keep)
print(len(rows), "->", len(keep), "rows")
