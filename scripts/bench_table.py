#!/usr/bin/env python3
"""Pretty-print bench.py JSON lines (stdin) as a table: headline + every `also` entry (round-3 compact format)."""
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
    for a in j.get("also", []):
        r, c = a["roofline"], a.get("cpu_baseline") or {}
        print("  also %-51s %8.3f ms/step %10.0f Mpx/s  frac %.3f  traffic_frac %s  cpu %s x%s" % (
            a.get("workload") or a["config"]["workload"], a["ms_per_step"], a["value"], r["frac"], r.get("traffic_frac"), c.get("value"), c.get("cores")))
    d = j.get("device", {})
    if "flat_fill_ms" in d:
        print("  ceilings: flat_fill %.3f ms, three_plane_store_only %.3f ms; kernel = %.3f of flat fill, %.3f of the three-plane stores" % (
            d["flat_fill_ms"], d["three_plane_store_only_ms"], d.get("frac_of_flat_fill", 0), d.get("frac_of_three_plane_store", 0)))
    elif "store_ceilings_error" in d:
        print("  ceilings: ERROR", d["store_ceilings_error"])
