#!/usr/bin/env python3
"""Round 6: exact 2x upscales of u8 images (1080p -> 4K, 16 per call) by channel count and mode."""
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
for arg in sys.argv[1:]:
    name, val = arg.split("=")
    check(lib.kh_debug_set_option(name.encode(), int(val)))
    print(f"# dev option {name} = {val}")
N, W, H = 16, 1920, 1080
src = DeviceBuffer.from_numpy(bench.lcg_bytes(N * W * H * 4), st); dst = DeviceBuffer(N * 4 * W * H * 4, st, zeroed=False)
for ch in (1, 3, 4):
    n = W * H * ch
    for mode, code in (("nearest", 0), ("bilinear", 1)):
        fn = lambda: check(lib.kh_resize_fast_u8(s, src.ptr, dst.ptr, W, H, 2 * W, 2 * H, ch, code, 1, N, n, 4 * n))
        fn(); st.synchronize(); ts = []
        for r in range(3):
            e0, e1 = hip.Event(), hip.Event(); e0.record(st)
            for _ in range(2):
                fn()
            e1.record(st); st.synchronize(); ts.append(e0.elapsed_ms(e1) / 2)
        t = float(np.median(ts))
        print(f"1080p -> 4K {mode:8s} c{ch}: {t:7.3f} ms  frac {5 * n * N / t / 1e6 / 8000:.3f}")
