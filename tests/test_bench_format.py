"""The ONE JSON line of bench.py must fit the driver's 8 KB stdout tail with all BASELINE configs in it (round-2 VERDICT: the
line was ~13 KB, so C2 / C4 fell off the front of the record): the headline's full record, then ONE table row per further
workload as the last key (round 4: 22 workloads no longer fit as objects AND rows)."""
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _bench():
    spec = importlib.util.spec_from_file_location("bench_fmt", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fake(name):
    return {"key": name, "value": 482690.2, "unit": "Mpixels/s", "steps": 10, "warmup": 2, "ms_per_step": 19.5512, "dtype": "f32",
            "config": {"workload": name, "op": "x" * 120, "src": "3840x2160x3 f32", "dst": "same", "batch_per_gpu": 256,
                       "parallelism": "batch-sharded, no collective"},
            "roofline": {"bound": "hbm", "achieved": 6081.3, "peak": 8000.0, "unit": "GB/s", "frac": 0.7602, "traffic": 98012345678,
                         "kernel": "remap_kernel<3,bilinear>+warp_perspective_kernel<3,bilinear>", "alg_bytes_per_launch": 118908518400,
                         "mean_launch_ms": 19.5512, "min_launch_ms": 19.4, "traffic_frac": None, "traffic_over_alg": 1.035, "floor_bytes": 2812345678,
                         "floor_GBps": 6012.2, "floor_frac": 0.7515},
            "n_gpus": 1,
            "cpu_baseline": {"value": 219.33, "unit": "Mpixels/s", "cores": 128, "kind": "port", "sample": "y" * 200, "threads": 128, "team_threads": 16,
                             "cgroup_cpus": 16.0, "physical_cores": 128, "affinity_cpus": 256}}


def test_line_fits_the_drivers_tail_and_ends_with_the_summary():
    b = _bench()
    names = [f"{n}_b256" for n in ["nv12_1080p_to_chw_f32"] + b.ALSO_DEFAULT + ["extra_a", "extra_b"]]  # room for two more workloads
    recs = [_fake(n) for n in names]
    head = recs[0]
    line = {"metric": "Mpixels/s, fused 1080p NV12->normalized CHW f32 (achieved HBM GB/s in roofline)", "value": head["value"],
            "unit": "Mpixels/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 4.399, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (LCG bytes, reference pattern_u8; frame k shifted by 31k)",
            "config": head["config"], "roofline": head["roofline"], "cpu_baseline": head["cpu_baseline"]}
    line["device"] = {"name": "AMD Instinct MI355X", "cus": 256, "hbm_bytes": 309220868096, "host_cpus": 128, "hip_runtime": "z" * 80,
                      "cpu_team": {"threads": 16, "physical_cores": 128, "affinity_cpus": 256, "cgroup_cpus": 16.0, "logical_cpus": 256},
                      "flat_fill_ms": 3.55, "three_plane_store_only_ms": 4.04, "store_bytes": 25480396800, "frac_of_flat_fill": 0.82,
                      "frac_of_three_plane_store": 0.93, "note": "n" * 230}
    line["traffic_source"] = "t" * 260
    line["summary_columns"] = b.SUMMARY_COLUMNS
    line["summary"] = [b.summary_row(r) for r in recs]
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 7600, len(text)
    assert list(line)[-1] == "summary"
    tail = text[-3500:]
    start = tail.find('"summary":')
    assert start >= 0, "the summary table must fit the last 3.5 KB of the line"
    rows = json.loads(tail[start + len('"summary":'):-1])
    assert [r[0] for r in rows] == names and all(len(r) == len(b.SUMMARY_COLUMNS) for r in rows)
    assert rows[0][-1] == 128  # cores travel with every row
