#!/usr/bin/env python3
"""Round 6: single-image device time of the camera-format conversions (the reference's API is one image per call), 1080p and 4K."""
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
for (W, H) in ((1920, 1080), (3840, 2160)):
    rgb = DeviceBuffer.from_numpy(bench.lcg_bytes(W * H * 3), st)
    yuv = DeviceBuffer.from_numpy(bench.lcg_bytes(W * H * 2), st)
    out = DeviceBuffer(W * H * 4, st, zeroed=False)
    s = st.cuda_stream_ptr
    ops = {"rgb_from_nv12": lambda: lib.kh_rgb_from_planar420_u8(s, yuv.ptr, out.ptr, W, H, 0),
           "rgb_from_i420": lambda: lib.kh_rgb_from_planar420_u8(s, yuv.ptr, out.ptr, W, H, 2),
           "rgb_from_yuyv": lambda: lib.kh_rgb_from_packed422_u8(s, yuv.ptr, out.ptr, W, H, 0),
           "nv12_from_rgb": lambda: lib.kh_nv12_from_rgb_u8(s, rgb.ptr, out.ptr, W, H),
           "yuyv_from_rgb": lambda: lib.kh_yuyv_from_rgb_u8(s, rgb.ptr, out.ptr, W, H),
           "gray_from_rgb_u8": lambda: lib.kh_gray_from_rgb_u8(s, rgb.ptr, out.ptr, W * H),
           "bgr_from_rgb_u8": lambda: lib.kh_rgb_swizzle_u8(s, rgb.ptr, out.ptr, W * H, 0) if hasattr(lib, "kh_rgb_swizzle_u8") else 0}
    for name, fn in ops.items():
        if fn() != 0:
            print(name, "error", _ffi.last_error()); continue
        st.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = hip.Event(), hip.Event(); e0.record(st)
            for _ in range(50):
                fn()
            e1.record(st); st.synchronize()
            ts.append(e0.elapsed_ms(e1) / 50 * 1000)
        print(f"{W}x{H} {name:18s} {np.median(ts):7.1f} us per call (back to back)")
