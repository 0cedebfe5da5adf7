#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (the ROCm 7.2 default output) as CSV text:
per-kernel calls / total / mean / min / max duration (us) + launch geometry and registers.
    python scripts/rocpd_summary.py gpurun_out/r01/prof/x_results.db > profiles/r01_x_kernel_stats.csv
"""
import sqlite3
import sys


def main(path: str) -> None:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, "
        "max(duration)/1e3, max(grid_x), max(grid_y), max(workgroup_x), max(vgpr_count), "
        "max(sgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1.0
    print("kernel,calls,total_us,mean_us,min_us,max_us,pct,grid_x,grid_y,block_x,vgpr,sgpr,lds_bytes")
    for r in rows:
        name = r[0].replace(",", ";")
        print(f"\"{name}\",{r[1]},{r[2]:.1f},{r[3]:.2f},{r[4]:.2f},{r[5]:.2f},{100*r[2]/total:.2f},"
              f"{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{r[11]}")
    try:
        pmc = cur.execute("select * from counters_collection limit 1").description
    except sqlite3.Error:
        pmc = None
    if pmc:
        cols = [d[0] for d in pmc]
        if "counter_name" in cols and "value" in cols and "kernel_name" in cols:
            print("\nkernel,counter,mean_value_per_dispatch,dispatches")
            for k, c, v, n in cur.execute(
                "select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                "group by kernel_name, counter_name order by kernel_name, counter_name"):
                print(f"\"{k.replace(',', ';')}\",{c},{v:.1f},{n}")


if __name__ == "__main__":
    main(sys.argv[1])
