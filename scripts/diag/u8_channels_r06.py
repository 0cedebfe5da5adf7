#!/usr/bin/env python3
"""Round 6: the u8 twins on 1- and 4-channel images next to RGB (32 x 4K): gaussian 5x5 / 7x7, box dilate 5x5, pyrdown, resize nearest / bilinear 2x."""
import ctypes as C
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
N, W, H = 32, 3840, 2160
def timeit(fn):
    rc = fn()
    if rc != 0:
        return float("nan")
    st.synchronize(); ts = []
    for r in range(3):
        e0, e1 = hip.Event(), hip.Event(); e0.record(st)
        for _ in range(2):
            fn()
        e1.record(st); st.synchronize(); ts.append(e0.elapsed_ms(e1) / 2)
    return float(np.median(ts))
for ch in (1, 3, 4):
    n = W * H * ch
    src = DeviceBuffer.from_numpy(bench.lcg_bytes(N * n), st); dst = DeviceBuffer(N * n, st, zeroed=False)
    s = st.cuda_stream_ptr
    mask = (C.c_uint8 * 25)(*([1] * 25)); cval = (C.c_uint8 * 4)(0, 0, 0, 0)
    ops = {"gaussian 5x5": (lambda: lib.kh_gaussian_blur_u8(s, src.ptr, dst.ptr, W, H, ch, 5, 5, 1.1, 1.1, N, n, n), 2 * n * N),
           "gaussian 7x7": (lambda: lib.kh_gaussian_blur_u8(s, src.ptr, dst.ptr, W, H, ch, 7, 7, 1.5, 1.5, N, n, n), 2 * n * N),
           "dilate 5x5 box": (lambda: lib.kh_morphology_u8(s, src.ptr, dst.ptr, W, H, ch, 0, mask, 5, 5, 0, cval, N, n, n), 2 * n * N),
           "pyrdown": (lambda: lib.kh_pyrdown_u8(s, src.ptr, dst.ptr, W, H, ch, N, n, n // 4), n * N * 5 // 4),
           "resize nearest 2x down": (lambda: lib.kh_resize_fast_u8(s, src.ptr, dst.ptr, W, H, W // 2, H // 2, ch, 0, 1, N, n, n // 4), n * N * 5 // 4),
           "resize bilinear 1.5x down": (lambda: lib.kh_resize_fast_u8(s, src.ptr, dst.ptr, W, H, 2560, 1440, ch, 1, 1, N, n, 2560 * 1440 * ch), N * (n + 2560 * 1440 * ch))}
    for name, (fn, nbytes) in ops.items():
        t = timeit(fn)
        print(f"c{ch} {name:26s}: {t:7.3f} ms   frac {nbytes / t / 1e6 / 8000:.3f}")
    del src, dst
