#!/usr/bin/env python3
"""Round 6: gaussian blur over kernel sizes, f32x3 and u8x3 4K, 32 images (sweep for slow fallbacks)."""
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
N, W, H, C = 32, 3840, 2160, 3
n = W * H * C
srcf = DeviceBuffer(N * n * 4, st, zeroed=True); dstf = DeviceBuffer(N * n * 4, st, zeroed=False)
srcu = DeviceBuffer.from_numpy(bench.lcg_bytes(N * n), st); dstu = DeviceBuffer(N * n, st, zeroed=False)
for K in (3, 5, 7, 9, 11, 13, 15, 17, 21, 31):
    sig = 0.3 * ((K - 1) * 0.5 - 1) + 0.8
    for name, fn, nbytes in (("f32", lambda: lib.kh_gaussian_blur_f32(st.cuda_stream_ptr, srcf.ptr, dstf.ptr, W, H, C, K, K, sig, sig, N, n, n), 8 * n * N),
                             ("u8 ", lambda: lib.kh_gaussian_blur_u8(st.cuda_stream_ptr, srcu.ptr, dstu.ptr, W, H, C, K, K, sig, sig, N, n, n), 2 * n * N)):
        rc = fn()
        if rc != 0:
            print(f"K={K} {name}: error {rc} {_ffi.last_error()[:80]}"); continue
        st.synchronize()
        ts = []
        for r in range(3):
            e0, e1 = hip.Event(), hip.Event(); e0.record(st)
            for _ in range(2):
                fn()
            e1.record(st); st.synchronize()
            ts.append(e0.elapsed_ms(e1) / 2)
        t = float(np.median(ts))
        print(f"gaussian {K:2d}x{K:<2d} {name}: {t:8.3f} ms   frac {nbytes / t / 1e6 / 8000:.3f}")
