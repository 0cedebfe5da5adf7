#!/usr/bin/env python3
"""Round 6: warp_affine / warp_perspective f32 and u8 over transforms and interpolation modes, 64 x 1080p x 3 (sweep for slow fallbacks)."""
import ctypes as C
import math
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kornia-rs_amd")); sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
from kornia_rs import _ffi, hip
from kornia_rs.hip import DeviceBuffer
import bench
lib, check = _ffi.lib, _ffi.check
hip.set_device(0); st = hip.Stream.new(0)
N, W, H, Cc = 64, 1920, 1080, 3
n = W * H * Cc
srcf = DeviceBuffer(N * n * 4, st, zeroed=True); dstf = DeviceBuffer(N * n * 4, st, zeroed=False)
srcu = DeviceBuffer.from_numpy(bench.lcg_bytes(N * n), st); dstu = DeviceBuffer(N * n, st, zeroed=False)
def rot(deg, s=1.0):
    out = (C.c_float * 6)(); lib.kh_get_rotation_matrix2d(W / 2, H / 2, deg, s, out); return list(out)
AFF = {"identity": [1, 0, 0, 0, 1, 0], "shift": [1, 0, 7.25, 0, 1, -3.5], "rot5": rot(5), "rot45": rot(45), "rot90": rot(90), "zoom2": rot(0, 2.0), "shrink4": rot(0, 0.25), "hflip": [-1, 0, W - 1, 0, 1, 0]}
MODES = {"nearest": 0, "bilinear": 1, "bicubic": 2}
def timeit(fn):
    rc = fn()
    if rc != 0:
        return None
    st.synchronize(); ts = []
    for r in range(3):
        e0, e1 = hip.Event(), hip.Event(); e0.record(st)
        for _ in range(2):
            fn()
        e1.record(st); st.synchronize(); ts.append(e0.elapsed_ms(e1) / 2)
    return float(np.median(ts))
for name, m in AFF.items():
    mm = (C.c_float * 6)(*m)
    row = [f"affine {name:9s}"]
    for mode, code in MODES.items():
        t = timeit(lambda: lib.kh_warp_affine_f32(st.cuda_stream_ptr, srcf.ptr, dstf.ptr, W, H, W, H, Cc, mm, code, N, n, n))
        row.append(f"f32 {mode} {t:7.3f}" if t else f"f32 {mode}   error")
    t = timeit(lambda: lib.kh_warp_affine_u8(st.cuda_stream_ptr, srcu.ptr, dstu.ptr, W, H, W, H, Cc, mm, N, n, n))
    row.append(f"u8 {t:7.3f}" if t else "u8 error " + _ffi.last_error()[:40])
    print(" | ".join(row))
HOM = {"identity": [1, 0, 0, 0, 1, 0, 0, 0, 1], "mild": [1.03, 0.05, -14.0, -0.02, 0.97, 44.0, 2.0 / (H * W), 1.5 / (W * H), 1.0], "strong": [0.7, -0.2, 300.0, 0.25, 0.8, -100.0, 0.0004, -0.0002, 1.0],
       "keystone": [1.0, 0.3, 0.0, 0.0, 1.2, 0.0, 0.0, 0.0005, 1.0]}
for name, m in HOM.items():
    mm = (C.c_float * 9)(*m)
    row = [f"persp  {name:9s}"]
    for mode, code in MODES.items():
        t = timeit(lambda: lib.kh_warp_perspective_f32(st.cuda_stream_ptr, srcf.ptr, dstf.ptr, W, H, W, H, Cc, mm, code, N, n, n))
        row.append(f"f32 {mode} {t:7.3f}" if t else f"f32 {mode}   error")
    t = timeit(lambda: lib.kh_warp_perspective_u8(st.cuda_stream_ptr, srcu.ptr, dstu.ptr, W, H, W, H, Cc, mm, N, n, n))
    row.append(f"u8 {t:7.3f}" if t else "u8 error " + _ffi.last_error()[:40])
    print(" | ".join(row))
print(f"(copy floors: f32 {2 * N * n * 4 / 6.4e9:.3f} ms, u8 {2 * N * n / 6.4e9:.3f} ms)")
