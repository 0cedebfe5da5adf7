#!/usr/bin/env python3
"""Round 6: warp_perspective f32x3 bilinear 4K (the slower half of BASELINE configs[4]) under test option warp_f32_px, interleaved in one
process; every variant's output is compared with the production kernel's bit for bit."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kornia-rs_amd"))
import torch  # noqa: F401
from kornia_rs import _ffi, hip
from kornia_rs.hip import DeviceBuffer

lib, check = _ffi.lib, _ffi.check
N, W, H, Cc = 128, 3840, 2160, 3
opts = [int(v) for v in sys.argv[1:]] or [-1, 2, 4]
hip.set_device(0)
st = hip.Stream.new(0)
n = W * H * Cc
rng = np.random.default_rng(1)
one = rng.random(n, dtype=np.float32)
src = DeviceBuffer(N * n * 4, st, zeroed=False)
for k in range(N):
    hip.h2d(src.ptr + k * n * 4, np.roll(one, 31 * k), st) if k < 4 else check(lib.kh_memcpy_d2d_async(src.ptr + k * n * 4, src.ptr + (k % 4) * n * 4, n * 4, st.cuda_stream_ptr))
dst, ref = DeviceBuffer(N * n * 4, st, zeroed=False), DeviceBuffer(4 * n * 4, st, zeroed=False)
w, h = float(W), float(H)
hm = (C.c_float * 9)(1.03, 0.05, -3.0 * w / 129.0, -0.02, 0.97, 4.0 * h / 97.0, 2.0 / (h * w), 1.5 / (w * h), 1.0)


def run(o, d=dst, nimg=N):
    check(lib.kh_debug_set_option(b"warp_f32_px", o))
    check(lib.kh_warp_perspective_f32(st.cuda_stream_ptr, src.ptr, d.ptr, W, H, W, H, Cc, hm, 1, nimg, n, n))


run(-1, ref, 4)
want = ref.to_numpy(np.uint32, (4, n))
times = {o: [] for o in opts}
for o in opts:
    check(lib.kh_memset_async(dst.ptr, 0xCD, 4 * n * 4, st.cuda_stream_ptr))
    run(o)
    got = dst.to_numpy(np.uint32, (4, n))
    print(f"option {o}: {'bit-equal' if np.array_equal(got, want) else 'MISMATCH %d' % int((got != want).sum())}")
for rnd in range(7):
    for o in opts:
        run(o); st.synchronize()
        e0, e1 = hip.Event(), hip.Event()
        e0.record(st)
        for _ in range(3):
            run(o)
        e1.record(st); st.synchronize()
        if rnd:
            times[o].append(e0.elapsed_ms(e1) / 3)
print(f"# warp_perspective f32x3 bilinear, {N} 4K images, 6 interleaved rounds x 3 launches")
for o in opts:
    v = times[o]
    print(f"warp_f32_px={o:3d}  median {np.median(v):.4f}  min {min(v):.4f} ms")
