#!/usr/bin/env python3
"""Round 6: the f32 and u8 filters on 1- / 3- / 4-channel 4K images and on odd widths (sweep for slow fallbacks); optional name=value dev option."""
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
if len(sys.argv) > 1:
    name, val = sys.argv[1].split("=")
    check(lib.kh_debug_set_option(name.encode(), int(val)))
    print(f"# dev option {name} = {val}")
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
def add(name, fn, nbytes):
    t = timeit(fn)
    print(f"{name:44s}: {t:8.3f} ms   frac {nbytes / t / 1e6 / 8000:.3f}" if t == t else f"{name:44s}: error {_ffi.last_error()[:60]}")
N = 16
bufu = DeviceBuffer.from_numpy(bench.lcg_bytes(N * 3840 * 2160 * 4), st)
buff = DeviceBuffer(N * 3840 * 2160 * 4 * 4, st, zeroed=True)
outb = DeviceBuffer(N * 3840 * 2160 * 4 * 4, st, zeroed=False)
out2 = DeviceBuffer(N * 3840 * 2160 * 4 * 4, st, zeroed=False)
for (W, H) in ((3840, 2160), (3839, 2160)):
    for ch in (1, 3, 4):
        n = W * H * ch
        tag = f"c{ch} {W}"
        for K in (3, 5, 9):
            add(f"gaussian f32 {K}x{K} {tag}", lambda: lib.kh_gaussian_blur_f32(s, buff.ptr, outb.ptr, W, H, ch, K, K, 1.5, 1.5, N, n, n), 8 * n * N)
        add(f"box f32 5x5 {tag}", lambda: lib.kh_box_blur_f32(s, buff.ptr, outb.ptr, W, H, ch, 5, 5, N, n, n), 8 * n * N)
        add(f"sobel f32 3 {tag}", lambda: lib.kh_gradient_magnitude_f32(s, buff.ptr, outb.ptr, W, H, ch, 0, 3, N, n, n), 8 * n * N)
        add(f"spatial_gradient f32 {tag}", lambda: lib.kh_spatial_gradient_f32(s, buff.ptr, outb.ptr, out2.ptr, W, H, ch, 0, N, n, n), 12 * n * N)
        for K in (3, 5, 9):
            add(f"gaussian u8 {K}x{K} {tag}", lambda: lib.kh_gaussian_blur_u8(s, bufu.ptr, outb.ptr, W, H, ch, K, K, 1.5, 1.5, N, n, n), 2 * n * N)
        add(f"box u8 5x5 {tag}", lambda: lib.kh_box_blur_u8(s, bufu.ptr, outb.ptr, W, H, ch, 5, 5, N, n, n), 2 * n * N)
