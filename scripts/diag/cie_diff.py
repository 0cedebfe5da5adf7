#!/usr/bin/env python3
"""Dev: per-conversion, per-channel max |device - restatement| of the CIE conversions on the inputs of tests/test_cie.py (+ time)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "kornia-rs_amd"), str(ROOT / "tests")]
import oracle_ffi as O
from kornia_rs import _ffi, hip
from gpu_util import dev, out_buf

st = hip.Stream.new(0)
rng = np.random.default_rng(11)
base = rng.random((1080, 1920, 3)).astype(np.float32)
base[0, :4] = [[0, 0, 0], [1, 1, 1], [0.04045, 0.0031308, 0.008856], [-0.25, 1.5, 0.5]]
for name in O.CIE:
    src = base
    if name in ("rgb_from_lab", "rgb_from_luv", "rgb_from_xyz", "rgb_from_linear_rgb"):
        src = O.cie({"rgb_from_lab": "lab_from_rgb", "rgb_from_luv": "luv_from_rgb", "rgb_from_xyz": "xyz_from_rgb",
                     "rgb_from_linear_rgb": "linear_rgb_from_rgb"}[name], src)
    d_src, d_dst = dev(st, src), out_buf(st, src.nbytes)
    _ffi.check(_ffi.lib.kh_cie_convert_f32(st.cuda_stream_ptr, d_src.ptr, d_dst.ptr, src.size // 3, O.CIE[name]))
    got = d_dst.to_numpy(np.float32, src.shape)
    want = O.cie(name, src)
    d = np.abs(got.astype(np.float64) - want.astype(np.float64)).reshape(-1, 3)
    ulp = (np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))).reshape(-1, 3)
    rel = d / np.maximum(np.abs(want.reshape(-1, 3)), 1e-30)
    print(f"{name:22s} max abs {d.max(axis=0)}  max ulp {ulp.max(axis=0)}  differing {100.0 * (ulp > 0).mean():.2f} %  max |want| {np.abs(want).reshape(-1,3).max(axis=0)}")
