#!/usr/bin/env python3
"""Round 6: f32 resizes 1080p -> 4K (exact 2x up) and 1080p -> 1440p (1.33x up) by mode and channel count, 8 images."""
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
s = st.cuda_stream_ptr
for arg in sys.argv[1:]:
    name, val = arg.split("=")
    check(lib.kh_debug_set_option(name.encode(), int(val)))
N = 8
src = DeviceBuffer(N * 1920 * 1080 * 4 * 4, st, zeroed=True); dst = DeviceBuffer(N * 3840 * 2160 * 4 * 4, st, zeroed=False)
for (dw, dh) in ((3840, 2160), (2560, 1440)):
    for ch in (1, 3, 4):
        n, m = 1920 * 1080 * ch, dw * dh * ch
        for mode, code in (("nearest", 0), ("bilinear", 1), ("bicubic", 2)):
            fn = lambda: check(lib.kh_resize_f32(s, src.ptr, dst.ptr, 1920, 1080, dw, dh, ch, code, N, n, m))
            fn(); st.synchronize(); ts = []
            for r in range(3):
                e0, e1 = hip.Event(), hip.Event(); e0.record(st)
                for _ in range(2):
                    fn()
                e1.record(st); st.synchronize(); ts.append(e0.elapsed_ms(e1) / 2)
            t = float(np.median(ts))
            print(f"f32 1080p -> {dw}x{dh} {mode:8s} c{ch}: {t:7.3f} ms  frac {(n + m) * 4 * N / t / 1e6 / 8000:.3f}")
