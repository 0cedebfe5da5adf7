#!/usr/bin/env python3
"""Round 6: the row-structured operator families on images whose rows are NOT whole 128-byte lines (1000-pixel rows and friends).
Run once with the product library and once with KORNIA_HIP_LIB pointing at a build whose streaming policy is off (kAuxStream = 0)
to see which kernels pay for partially written lines (scripts/diag/ab_misaligned_r06.sh)."""
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
if len(sys.argv) > 1:   # name=value dev option (row_stores=0 streaming always, 1 write-back always)
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
def add(name, fn):
    t = timeit(fn)
    print(f"{name:46s}: {t:8.3f} ms" if t == t else f"{name:46s}: error {_ffi.last_error()[:60]}")
N = 64
W, H = 1000, 750
bufu = DeviceBuffer.from_numpy(bench.lcg_bytes(N * 2048 * 1100 * 4), st)
buff = DeviceBuffer(N * 2048 * 1100 * 4 * 4, st, zeroed=True)
outb = DeviceBuffer(N * 2048 * 1100 * 4 * 4, st, zeroed=False)
out2 = DeviceBuffer(N * 2048 * 1100 * 4 * 4, st, zeroed=False)
cval = (C.c_uint8 * 4)(0, 0, 0, 0)
aff = (C.c_float * 6)(0.996, -0.087, 30.0, 0.087, 0.996, -20.0)
hom = (C.c_float * 9)(1.03, 0.05, -14.0, -0.02, 0.97, 44.0, 2e-6, 1.5e-6, 1.0)
for (W, H) in ((1000, 750), (1008, 750), (1024, 750)):
    print(f"# {W} x {H}, {N} images")
    for ch in (3, 1):
        n = W * H * ch
        t = f"c{ch} {W}"
        add(f"gaussian u8 5x5 {t}", lambda: lib.kh_gaussian_blur_u8(s, bufu.ptr, outb.ptr, W, H, ch, 5, 5, 1.0, 1.0, N, n, n))
        add(f"gaussian u8 3x3 {t}", lambda: lib.kh_gaussian_blur_u8(s, bufu.ptr, outb.ptr, W, H, ch, 3, 3, 0.8, 0.8, N, n, n))
        mask = (C.c_uint8 * 25)(*([1] * 25))
        add(f"dilate u8 5x5 {t}", lambda: lib.kh_morphology_u8(s, bufu.ptr, outb.ptr, W, H, ch, 0, mask, 5, 5, 0, cval, N, n, n))
        add(f"pyrdown u8 {t}", lambda: lib.kh_pyrdown_u8(s, bufu.ptr, outb.ptr, W, H, ch, N, n, (W // 2) * (H // 2) * ch))
        add(f"pyrup u8 (to {W}) {t}", lambda: lib.kh_pyrup_u8(s, bufu.ptr, outb.ptr, W // 2, H // 2, ch, N, (W // 2) * (H // 2) * ch, n))
        add(f"pyrdown f32 {t}", lambda: lib.kh_pyrdown_f32(s, buff.ptr, outb.ptr, W, H, ch, N, n, (W // 2) * (H // 2) * ch))
        add(f"pyrup f32 (to {W}) {t}", lambda: lib.kh_pyrup_f32(s, buff.ptr, outb.ptr, W // 2, H // 2, ch, N, (W // 2) * (H // 2) * ch, n))
        add(f"spatial_gradient f32 {t}", lambda: lib.kh_spatial_gradient_f32(s, buff.ptr, outb.ptr, out2.ptr, W, H, ch, 0, N, n, n))
    ch = 3; n = W * H * ch; t = f"c3 {W}"
    for mode, code in (("nearest", 0), ("bilinear", 1), ("bicubic", 2)):
        add(f"resize_fast_u8 2x{W} -> {W} {mode} {t}", lambda: lib.kh_resize_fast_u8(s, bufu.ptr, outb.ptr, 2 * W, 2 * H, W, H, ch, code, 1, N // 4, 4 * n, n))
        add(f"resize f32 1920x1080 -> {W} {mode} {t}", lambda: lib.kh_resize_f32(s, buff.ptr, outb.ptr, 1920, 1080, W, H, ch, code, N // 4, 1920 * 1080 * 3, n))
    add(f"resize f32 1920x1080 -> {W} lanczos {t}", lambda: lib.kh_resize_f32(s, buff.ptr, outb.ptr, 1920, 1080, W, H, ch, 3, N // 4, 1920 * 1080 * 3, n))
    add(f"warp_affine u8 {t}", lambda: lib.kh_warp_affine_u8(s, bufu.ptr, outb.ptr, W, H, W, H, ch, aff, N, n, n))
    add(f"warp_perspective u8 {t}", lambda: lib.kh_warp_perspective_u8(s, bufu.ptr, outb.ptr, W, H, W, H, ch, hom, N, n, n))
    for mode, code in (("nearest", 0), ("bilinear", 1), ("bicubic", 2)):
        add(f"warp_affine f32 {mode} {t}", lambda: lib.kh_warp_affine_f32(s, buff.ptr, outb.ptr, W, H, W, H, ch, aff, code, N // 4, n, n))
    add(f"warp_perspective f32 bilinear {t}", lambda: lib.kh_warp_perspective_f32(s, buff.ptr, outb.ptr, W, H, W, H, ch, hom, 1, N // 4, n, n))
    add(f"box_blur f32 5x5 {t}", lambda: lib.kh_box_blur_f32(s, buff.ptr, outb.ptr, W, H, ch, 5, 5, N // 4, n, n))
