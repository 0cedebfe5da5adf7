#!/usr/bin/env python3
"""Round 6: gaussian blur f32x3 4K x 32 with UNEQUAL tap counts (one-dimensional blurs, (3, 7), ...): the masked rolling kernel vs the tile kernel."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kornia-rs_amd")); sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
from kornia_rs import _ffi, hip
from kornia_rs.hip import DeviceBuffer
lib, check = _ffi.lib, _ffi.check
hip.set_device(0); st = hip.Stream.new(0)
N, W, H, C = 32, 3840, 2160, 3
n = W * H * C
src = DeviceBuffer(N * n * 4, st, zeroed=True); dst = DeviceBuffer(N * n * 4, st, zeroed=False)
for (kx, ky) in ((7, 7), (3, 7), (9, 5), (1, 7), (15, 1), (13, 3)):
    for opt in (-1, 1):
        check(lib.kh_debug_set_option(b"filter_force_tile", opt))
        fn = lambda: lib.kh_gaussian_blur_f32(st.cuda_stream_ptr, src.ptr, dst.ptr, W, H, C, kx, ky, 1.5, 1.5, N, n, n)
        check(fn()); st.synchronize()
        ts = []
        for r in range(3):
            e0, e1 = hip.Event(), hip.Event(); e0.record(st)
            for _ in range(2):
                fn()
            e1.record(st); st.synchronize()
            ts.append(e0.elapsed_ms(e1) / 2)
        print(f"gaussian ({kx:2d}, {ky:2d}) force_tile={opt:2d}: {np.median(ts):8.3f} ms")
