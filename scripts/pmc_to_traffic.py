#!/usr/bin/env python3
"""Rebuild profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of the default bench run (scripts/r04_final.sh):
HBM bytes per step = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the step's kernels, means per dispatch.

x2: on gfx950 every HBM read is a 128-byte request that FETCH_SIZE tallies at 64 B (MI355X_MICROARCH.md "HBM").  Round 4 CALIBRATED
the factor on known byte counts per access shape (scripts/ubench/fetch_calib_r04.hip, profiles/r04a_fetch_calibration.txt): wide
16 B / lane, 4 B / lane, 12-byte pixels, one dword per 128 / 64 / 32 B, the C2 resize's 24-byte tap pairs, unaligned 6- and 8-byte
u8 taps, LDS-staged reads — in every one TCC_EA0_RDREQ_128B x 128 B equals the 128-byte lines touched and FETCH_SIZE is exactly half
of it (32-B and 64-B read requests: none).  WRITE_SIZE equals the bytes written by full-line 16 B / lane stores exactly.

    python scripts/pmc_to_traffic.py <FETCH counter_collection.csv> <WRITE counter_collection.csv> "<source note>"
accepts rocprofv3's own `--output-format csv` counter_collection files (one row per dispatch x counter)."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
# workload (bench.py name incl. batch) -> substrings of the kernels one step launches
KERNELS = {
    "gray_from_rgb_u8_1080p_b1024": ["GrayFromRgbU8"],
    "nv12_1080p_to_chw_f32_b1024": ["preprocess_nv12_identity("],
    "nv12_1080p_to_chw_f32_letterbox640_b1024": ["preprocess_generic_quads<3, 100"],
    "nv12_1080p_to_chw_f32_letterbox608_b1024": ["preprocess_generic_quads<3, 1,", "preprocess_generic<3, 1, float"],
    "yuyv_1080p_to_chw_f32_letterbox640_b1024": ["preprocess_generic_quads<4, 100"],
    "resize_bilinear_1080p_to_224_f32_b256": ["resize_rows_bilinear_kernel<3, 1, 256, false"],
    "resize_bilinear_1080p_to_224_f32_api_list_b256": ["resize_rows_bilinear_kernel<3, 1, 256, true"],
    "resize_bicubic_1080p_to_540p_f32_b256": ["resize_bicubic_half_kernel<3, true, false", "resize_kernel<3, 2, false"],
    "gaussian_blur_7x7_4k_f32_b256": ["sep_roll4_kernel<7"],
    "box_blur_5x5_4k_f32_b128": ["sep_roll4_kernel<5"],
    "sobel_3x3_4k_f32_b128": ["sep_roll4_kernel<3, 3, true", "sep_roll_kernel<3, true"],
    "undistort_remap_then_warp_perspective_4k_f32_b256": ["remap_kernel<3, 1, false", "warp_perspective_px_kernel<3, 1, 2, false"],
    "undistort_remap_then_warp_perspective_4k_f32_api_list_b256": ["remap_kernel<3, 1, true", "warp_perspective_px_kernel<3, 1, 2, true"],
    "warp_affine_f32_1080p_b256": ["warp_affine_kernel<3, 1, false", "warp_affine_px_kernel<3, 1, 2, false"],
    "nv12_1080p_to_chw_f32_frame_list_b1024": ["preprocess_nv12_identity_list"],
    "nv12_1080p_to_chw_f16_b1024": ["preprocess_nv12_identity_f16<true>(", "preprocess_nv12_identity_f16("],
    "gaussian_blur_7x7_4k_f32_api_list_b256": [],   # (shares sep_roll4_kernel<7 with the equally spaced row: not separable by name)
    "normalize_mean_std_1080p_f32_b512": ["normalize_mean_std_quads3_kernel", "normalize_mean_std_kernel<3"],
    "gray_from_rgb_f32_1080p_b1024": ["GrayFromRgbF32"],
    "hsv_from_rgb_f32_1080p_b512": ["HsvFromRgbF32"],
    "ycc_from_rgb_u8_1080p_b1024": ["YccFromRgbU8"],
    "ycc_from_rgb_f32_1080p_b512": ["YccFromRgbF32"],
    "warp_affine_u8_4k_b256": ["gather_u8_staged_kernel<3, 0,"],
    "warp_perspective_u8_4k_b256": ["gather_u8_staged_kernel<3, 1,"],
    "remap_u8_undistort_4k_b256": ["gather_u8_staged_kernel<3, 2,"],
    "gaussian_blur_u8_7x7_4k_b256": ["blur_u8_rgb_kernel<7, false, 3>", "blur_u8_roll_kernel<7, 3"],
}


# launches of each matched kernel per step, where a step is more than one (pointer-list rows: 256 frames / 128 images per launch)
LAUNCHES = {"nv12_1080p_to_chw_f32_frame_list_b1024": 4, "resize_bilinear_1080p_to_224_f32_api_list_b256": 2,
            "undistort_remap_then_warp_perspective_4k_f32_api_list_b256": 2}


def read(path, counter):
    """kernel -> mean counter value per dispatch, over the dispatches with that kernel's LARGEST grid: a kernel that the run also
    launches on a smaller problem (the identity kernel inside the 64-frame H2D workload, gray_u8 on the 258x195 plumbing image,
    setup launches) must not dilute the per-step figure of the full-size workload."""
    acc = defaultdict(list)
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append((int(row.get("Grid_Size") or 0), float(row["Counter_Value"])))
    out = {}
    for k, v in acc.items():
        g = max(x[0] for x in v)
        vals = [x[1] for x in v if x[0] == g]
        out[k] = sum(vals) / len(vals)
    return out


def main():
    fetch, write, note = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    res = {"_source": note,
           "_note": "HBM bytes per step = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the step's kernels (mean per dispatch); x2 is the gfx950 "
                    "FETCH_SIZE correction of MI355X_MICROARCH.md. Workloads not listed here were not in that run: bench.py reports traffic null for them.",
           "_calibration": "profiles/r04a_fetch_calibration.txt: ten known-byte read shapes (16 B, 4 B and 12 B per lane dense; one dword per 128 / 64 / 32 B; "
                           "the C2 resize's 24-B tap pairs; unaligned 6- / 8-B u8 taps; LDS-staged) — in every one FETCH_SIZE x 2 equals the 128-B lines touched "
                           "(all HBM reads are 128-B requests on gfx950, counted at 64 B); WRITE_SIZE equals the bytes of full-line 16 B / lane stores exactly. "
                           "Factor 2.000 for every shape: it is the line granularity, not the access width, that the counter sees."}
    unmatched = []
    for wl, pats in KERNELS.items():
        total, hit = 0.0, False
        for k in fetch:
            if any(p in k for p in pats):
                total += (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0 * LAUNCHES.get(wl, 1)
                hit = True
        if hit:
            res[wl] = int(round(total))
        else:
            unmatched.append(wl)
    (ROOT / "profiles" / "pmc_traffic.json").write_text(json.dumps(res, indent=1) + "\n")
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:58s} {v / 1e9:9.3f} GB")
    if unmatched:
        print("no kernel matched for:", unmatched)
        print("kernels seen:", sorted(fetch)[:60])


if __name__ == "__main__":
    main()
