#!/usr/bin/env python3
"""Round 6: nearest upscales of one-channel u8 images (masks) by factor, 16 images; name=value dev options."""
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
N = 16
src = DeviceBuffer.from_numpy(bench.lcg_bytes(N * 1920 * 1080 * 3), st); dst = DeviceBuffer(N * 3840 * 2160 * 3, st, zeroed=False)
for (sw, sh, dw, dh) in ((1920, 1080, 3840, 2160), (1280, 720, 3840, 2160), (960, 540, 3840, 2160), (480, 270, 3840, 2160), (1920, 1080, 2560, 1440), (1000, 750, 3000, 2000)):
  for ch in (1, 3):
    n, m = sw * sh * ch, dw * dh * ch
    for api in ("fast", "opencv"):
        fn = (lambda: check(lib.kh_resize_fast_u8(s, src.ptr, dst.ptr, sw, sh, dw, dh, ch, 0, 1, N, n, m))) if api == "fast" else (lambda: check(lib.kh_resize_opencv_u8(s, src.ptr, dst.ptr, sw, sh, dw, dh, ch, 0, N, n, m)))
        fn(); st.synchronize(); ts = []
        for r in range(3):
            e0, e1 = hip.Event(), hip.Event(); e0.record(st)
            for _ in range(2):
                fn()
            e1.record(st); st.synchronize(); ts.append(e0.elapsed_ms(e1) / 2)
        t = float(np.median(ts))
        print(f"nearest {api:6s} c{ch} {sw}x{sh} -> {dw}x{dh}: {t:7.3f} ms  frac {(n + m) * N / t / 1e6 / 8000:.3f}")
