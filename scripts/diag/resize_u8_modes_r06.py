#!/usr/bin/env python3
"""Round 6: kh_resize_fast_u8 (the reference's resize_fast_u8_aa cascade) over interpolation modes and common geometries, 256 x 1080p RGB8."""
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
N, W, H, C = 256, 1920, 1080, 3
n = W * H * C
base = bench.lcg_bytes(n + 31 * N)
dbase = DeviceBuffer.from_numpy(base, st)
src = DeviceBuffer(n * N, st, zeroed=False)
for k in range(N):
    check(lib.kh_memcpy_d2d_async(src.ptr + k * n, dbase.ptr + 31 * k, n, st.cuda_stream_ptr))
MODES = {"nearest": 0, "bilinear": 1, "bicubic": 2, "lanczos": 3}
for (dw, dh) in ((224, 224), (960, 540), (640, 360), (1280, 720), (512, 288), (2560, 1440)):
    dst = DeviceBuffer(N * dw * dh * C, st, zeroed=False)
    for mode, code in MODES.items():
        for aa in ((1,) if mode in ("nearest", "bilinear") else (1, 0)):
            def run():
                check(lib.kh_resize_fast_u8(st.cuda_stream_ptr, src.ptr, dst.ptr, W, H, dw, dh, C, code, aa, N, n, dw * dh * C))
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
            print(f"1080p -> {dw}x{dh} {mode:8s} aa={aa}: {t:7.3f} ms   src+dst at 6.4 TB/s: {N * (n + dw * dh * C) / 6.4e9:.3f} ms")
    del dst
