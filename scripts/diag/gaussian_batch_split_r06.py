#!/usr/bin/env python3
"""Round 6: gaussian 7x7 on 256 4K RGB f32 images (the bench row) as one launch of 256 images against 2 x 128, 4 x 64, 8 x 32 ... on the same buffers."""
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
N, W, H, C = 256, 3840, 2160, 3
n = W * H * C
src = DeviceBuffer(N * n * 4, st, zeroed=True); dst = DeviceBuffer(N * n * 4, st, zeroed=False)
def run(parts):
    per = N // parts
    for k in range(parts):
        check(lib.kh_gaussian_blur_f32(s, src.ptr + k * per * n * 4, dst.ptr + k * per * n * 4, W, H, C, 7, 7, 1.5, 1.5, per, n, n))
for rnd in range(3):
    for parts in (1, 2, 4, 8, 16, 32):
        run(parts); st.synchronize()
        e0, e1 = hip.Event(), hip.Event(); e0.record(st)
        for _ in range(3):
            run(parts)
        e1.record(st); st.synchronize()
        t = e0.elapsed_ms(e1) / 3
        print(f"round {rnd}: {parts:2d} launch(es) of {N // parts:3d} images: {t:7.3f} ms  frac {8 * n * N / t / 1e6 / 8000:.3f}")
