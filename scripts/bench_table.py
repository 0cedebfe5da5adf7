#!/usr/bin/env python3
"""Pretty-print bench.py JSON lines (stdin) as a table: the headline, then every row of the line's `summary` table."""
import json
import sys

for l in sys.stdin:
    if not l.startswith("{"):
        continue
    j = json.loads(l)
    r = j.get("roofline", {})
    c = j.get("cpu_baseline") or {}
    print("%-56s %8.3f ms/step %10.0f Mpx/s  frac %.3f  launch %.3f ms  cpu %s x%s   [line %d chars]" % (
        j["config"]["workload"], j["ms_per_step"], j["value"], r.get("frac") or 0, r.get("mean_launch_ms") or 0, c.get("value"), c.get("cores"), len(l)))
    cols = j.get("summary_columns") or []
    for row in (j.get("summary") or [])[1:]:
        d = dict(zip(cols, row))
        print("  also %-51s %8.3f ms/step %10.0f Mpx/s  frac %.3f  traffic_frac %s  cpu %s x%s" % (
            d.get("workload"), d.get("ms_per_step") or 0, d.get("Mpx_s") or 0, d.get("roofline_frac") or 0, d.get("traffic_frac"), d.get("cpu_Mpx_s"), d.get("cpu_cores")))
    d = j.get("device", {})
    if "flat_fill_ms" in d:
        print("  ceilings: flat_fill %.3f ms, three_plane_store_only %.3f ms; kernel = %.3f of flat fill, %.3f of the three-plane stores" % (
            d["flat_fill_ms"], d["three_plane_store_only_ms"], d.get("frac_of_flat_fill", 0), d.get("frac_of_three_plane_store", 0)))
        if "read_stream_ms" in d:
            print("  read stream of the same bytes %.3f ms = %.0f GB/s; kernel R + W rate = %.3f of that measured read rate" % (
                d["read_stream_ms"], d.get("read_stream_GBps", 0), d.get("kernel_rate_over_read_stream_rate", 0)))
    elif "store_ceilings_error" in d:
        print("  ceilings: ERROR", d["store_ceilings_error"])
    ex = {k: v for k, v in r.items() if k.startswith(("h2d_", "kernel_only", "end_to_end", "hidden_by", "serial_sum"))}
    if ex:
        print("  breakdown:", ex)
