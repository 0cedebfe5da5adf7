#!/usr/bin/env python3
"""Round 6: gaussian 7x7 on 256 4K RGB f32 images with the images spaced n + pad floats apart (is the bench row's gap to its pointer-list twin an address-interleaving effect?)."""
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
s = st.cuda_stream_ptr
N, W, H, C = 256, 3840, 2160, 3
n = W * H * C
PADMAX = 2 * 1024 * 1024
src = DeviceBuffer(N * (n + PADMAX) * 4, st, zeroed=False); dst = DeviceBuffer(N * (n + PADMAX) * 4, st, zeroed=False)
# random f32 in [1, 2)
seed = DeviceBuffer.from_numpy((bench.lcg_bytes(n * 4).view(np.uint32) >> 9 | 0x3f800000).view(np.uint8), st)
for rnd in range(2):
    for pad in (0, 32, 1024, 4096, 65536, 262144, 524288 + 1024, 1048576, 2 * 1024 * 1024 - (n % (512 * 1024))):
        stride = n + pad
        for k in range(N):
            check(lib.kh_memcpy_d2d_async(src.ptr + k * stride * 4, seed.ptr, n * 4, s))
        fn = lambda: check(lib.kh_gaussian_blur_f32(s, src.ptr, dst.ptr, W, H, C, 7, 7, 1.5, 1.5, N, stride, stride))
        fn(); st.synchronize()
        e0, e1 = hip.Event(), hip.Event(); e0.record(st)
        for _ in range(3):
            fn()
        e1.record(st); st.synchronize()
        t = e0.elapsed_ms(e1) / 3
        print(f"round {rnd}: image stride n + {pad:8d} floats ({stride * 4 % (2 << 20):8d} B mod 2 MiB): {t:7.3f} ms  frac {8 * n * N / t / 1e6 / 8000:.3f}")
