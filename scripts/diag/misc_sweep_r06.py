#!/usr/bin/env python3
"""Round 6: sweep of the remaining operator families at 4K / 1080p batches (fraction of 8 TB/s on in + out bytes) — looking for slow fallbacks."""
import ctypes as C
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
def timeit(fn):
    rc = fn()
    if rc != 0:
        return float("nan")
    st.synchronize(); ts = []
    for r in range(3):
        e0, e1 = hip.Event(), hip.Event(); e0.record(st)
        for _ in range(2):
            fn()
        e1.record(st); st.synchronize(); ts.append(e0.elapsed_ms(e1) / 2)
    return float(np.median(ts))
N, W, H = 32, 3840, 2160
npx = W * H * N
bufu = DeviceBuffer.from_numpy(bench.lcg_bytes(npx * 4), st)
buff = DeviceBuffer(npx * 4 * 4, st, zeroed=True)
outb = DeviceBuffer(npx * 4 * 4, st, zeroed=False)
f3 = (C.c_float * 3)(0.4, 0.5, 0.6); f3b = (C.c_float * 3)(0.2, 0.3, 0.25)
rows = []
def add(name, fn, nbytes):
    t = timeit(fn)
    print(f"{name:42s}: {t:8.3f} ms   frac {nbytes / t / 1e6 / 8000:.3f}" if t == t else f"{name:42s}: error {_ffi.last_error()[:60]}")
for ch in (1, 3, 4):
    add(f"pyrdown_u8 c{ch}", lambda: lib.kh_pyrdown_u8(s, bufu.ptr, outb.ptr, W, H, ch, N, W * H * ch, W * H * ch // 4), npx * ch * 5 // 4)
    add(f"pyrup_u8 c{ch} (1080p src)", lambda: lib.kh_pyrup_u8(s, bufu.ptr, outb.ptr, W // 2, H // 2, ch, N, W * H * ch // 4, W * H * ch), npx * ch * 5 // 4)
    add(f"pyrdown_f32 c{ch}", lambda: lib.kh_pyrdown_f32(s, buff.ptr, outb.ptr, W, H, ch, N // 4, W * H * ch, W * H * ch // 4), npx // 4 * ch * 5)
    add(f"pyrup_f32 c{ch} (1080p src)", lambda: lib.kh_pyrup_f32(s, buff.ptr, outb.ptr, W // 2, H // 2, ch, N // 4, W * H * ch // 4, W * H * ch), npx // 4 * ch * 5)
    add(f"normalize_mean_std c{ch}", lambda: lib.kh_normalize_mean_std_f32(s, buff.ptr, outb.ptr, npx // 4, ch, f3 if ch <= 3 else (C.c_float * 4)(.1, .2, .3, .4), f3b if ch <= 3 else (C.c_float * 4)(.1, .2, .3, .4)), npx // 4 * ch * 8)
add("normalize_rgb_u8 -> f32", lambda: lib.kh_normalize_rgb_u8_f32(s, bufu.ptr, outb.ptr, npx // 4, f3, f3b), npx // 4 * 15)
add("gray_from_rgb_u8", lambda: lib.kh_gray_from_rgb_u8(s, bufu.ptr, outb.ptr, npx), npx * 4)
add("rgb_from_gray_u8", lambda: lib.kh_rgb_from_gray_u8(s, bufu.ptr, outb.ptr, npx), npx * 4)
add("rgb_from_gray_f32", lambda: lib.kh_rgb_from_gray_f32(s, buff.ptr, outb.ptr, npx // 4), npx // 4 * 16)
add("hsv_from_rgb_f32", lambda: lib.kh_hsv_from_rgb_f32(s, buff.ptr, outb.ptr, npx // 4), npx // 4 * 24)
add("rgb_from_hsv_f32", lambda: lib.kh_rgb_from_hsv_f32(s, buff.ptr, outb.ptr, npx // 4), npx // 4 * 24)
add("rgb_from_hls_f32", lambda: lib.kh_rgb_from_hls_f32(s, buff.ptr, outb.ptr, npx // 4), npx // 4 * 24)
add("sepia_from_rgb_u8", lambda: lib.kh_sepia_from_rgb_u8(s, bufu.ptr, outb.ptr, npx), npx * 6)
add("sepia_from_rgb_f32", lambda: lib.kh_sepia_from_rgb_f32(s, buff.ptr, outb.ptr, npx // 4), npx // 4 * 24)
add("rgb_from_ycc_u8", lambda: lib.kh_rgb_from_ycc_u8(s, bufu.ptr, outb.ptr, npx, 0), npx * 6)
add("rgb_from_ycc_f32", lambda: lib.kh_rgb_from_ycc_f32(s, buff.ptr, outb.ptr, npx // 4, 0), npx // 4 * 24)
add("rgb_from_rgba_u8", lambda: lib.kh_rgb_from_rgba_u8(s, bufu.ptr, outb.ptr, npx, 0) if False else 1, 1)
