#!/usr/bin/env python3
"""Rebuild profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of the default bench run (scripts/r03_final.sh):
HBM bytes per step = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the step's kernels — x2 is the gfx950 FETCH_SIZE correction
MI355X_MICROARCH.md prescribes, WRITE_SIZE 1:1.

    python scripts/pmc_to_traffic.py profiles/r03z_default_pmc_FETCH_SIZE.csv profiles/r03z_default_pmc_WRITE_SIZE.csv "<source note>"
"""
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
# workload (bench.py name incl. batch) -> substrings of the kernels one step launches.  gray_u8_1080p is left out: its kernel is
# shared with the 258x195 plumbing workload in the same run, so the per-dispatch mean mixes two sizes.
KERNELS = {
    "nv12_1080p_to_chw_f32_b1024": ["preprocess_nv12_identity"],
    "nv12_1080p_to_chw_f32_letterbox640_b1024": ["preprocess_generic"],
    "resize_bilinear_1080p_to_224_f32_b256": ["resize_kernel<"],
    "gaussian_blur_7x7_4k_f32_b256": ["sep_roll4_kernel<7", "sep_roll_kernel<7"],
    "undistort_remap_then_warp_perspective_4k_f32_b256": ["remap_kernel<", "warp_perspective_kernel<"],
    "gray_from_rgb_f32_1080p_b1024": ["GrayFromRgbF32"],
    "hsv_from_rgb_f32_1080p_b512": ["HsvFromRgbF32"],
    "warp_affine_u8_4k_b256": ["gather_u8_staged_kernel<3; 0>"],
    "warp_perspective_u8_4k_b256": ["gather_u8_staged_kernel<3; 1>"],
    "remap_u8_undistort_4k_b256": ["gather_u8_staged_kernel<3; 2>"],
    "gaussian_blur_u8_7x7_4k_b256": ["blur_u8_rgb_kernel<7>", "blur_u8_roll_kernel<7; 3"],
}


def read(path):
    out = {}
    with open(path) as f:
        for row in csv.reader(f):
            if len(row) >= 4 and row[1] in ("FETCH_SIZE", "WRITE_SIZE"):
                out[row[0]] = float(row[2])
    return out


def main():
    fetch, write, note = read(sys.argv[1]), read(sys.argv[2]), sys.argv[3]
    res = {"_source": note,
           "_note": "HBM bytes per step = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the step's kernels; x2 is the gfx950 FETCH_SIZE correction "
                    "MI355X_MICROARCH.md prescribes, WRITE_SIZE 1:1. Workloads not listed here were not in that run: bench.py reports traffic null for them."}
    for wl, pats in KERNELS.items():
        total, hit = 0.0, False
        for k in fetch:
            if any(p in k for p in pats):
                total += (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0
                hit = True
        if hit:
            res[wl] = int(round(total))
    (ROOT / "profiles" / "pmc_traffic.json").write_text(json.dumps(res, indent=1) + "\n")
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:58s} {v / 1e9:9.3f} GB")


if __name__ == "__main__":
    main()
