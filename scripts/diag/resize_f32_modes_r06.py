#!/usr/bin/env python3
"""Round 6: kh_resize_f32 over interpolation modes and common geometries, 128 x 1080p f32x3 (sweep for anomalies)."""
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
N, W, H, C = 128, 1920, 1080, 3
n = W * H * C
one = np.random.default_rng(1).random(n, dtype=np.float32)
src = DeviceBuffer(N * n * 4, st, zeroed=False)
for k in range(N):
    hip.h2d(src.ptr + k * n * 4, np.roll(one, 31 * k), st) if k < 4 else check(lib.kh_memcpy_d2d_async(src.ptr + k * n * 4, src.ptr + (k % 4) * n * 4, n * 4, st.cuda_stream_ptr))
MODES = {"nearest": 0, "bilinear": 1, "bicubic": 2, "lanczos": 3}
for (dw, dh) in ((224, 224), (640, 360), (960, 540), (1280, 720), (1920, 1080), (2560, 1440)):
    dst = DeviceBuffer(N * dw * dh * C * 4, st, zeroed=False)
    for mode, code in MODES.items():
        def run():
            check(lib.kh_resize_f32(st.cuda_stream_ptr, src.ptr, dst.ptr, W, H, dw, dh, C, code, N, n, dw * dh * C))
        ts = []
        for r in range(4):
            run(); st.synchronize()
            e0, e1 = hip.Event(), hip.Event(); e0.record(st)
            for _ in range(3):
                run()
            e1.record(st); st.synchronize()
            if r:
                ts.append(e0.elapsed_ms(e1) / 3)
        t = float(np.median(ts))
        print(f"1080p -> {dw}x{dh} {mode:8s}: {t:7.3f} ms   src+dst at 6.4 TB/s: {N * 4 * (n + dw * dh * C) / 6.4e9:.3f} ms   dst alone: {N * 4 * dw * dh * C / 6.4e9:.3f}")
    del dst
