#!/usr/bin/env python3
"""Round 6: the warps / remap / resizes on 1- and 4-channel images next to RGB (16 x 4K u8, 8 x 4K f32): sweep for slow fallbacks."""
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
for arg in sys.argv[1:]:
    name, val = arg.split("=")
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
W, H = 3840, 2160
NU, NF = 16, 8
bufu = DeviceBuffer.from_numpy(bench.lcg_bytes(NU * W * H * 4), st)
buff = DeviceBuffer(NF * W * H * 4 * 4, st, zeroed=True)
outb = DeviceBuffer(NF * W * H * 4 * 4, st, zeroed=False)
aff = (C.c_float * 6)(0.996, -0.087, 30.0, 0.087, 0.996, -20.0)
hom = (C.c_float * 9)(1.03, 0.05, -14.0, -0.02, 0.97, 44.0, 2e-7, 1.5e-7, 1.0)
mx = np.tile(np.arange(W, dtype=np.float32) * 0.98 + 7.3, (H, 1)); my = np.tile((np.arange(H, dtype=np.float32) * 0.97 + 3.6)[:, None], (1, W))
dmx = DeviceBuffer.from_numpy(mx.view(np.uint8).reshape(-1), st); dmy = DeviceBuffer.from_numpy(my.view(np.uint8).reshape(-1), st)
for ch in (1, 3, 4):
    n = W * H * ch
    add(f"warp_affine u8 c{ch}", lambda: lib.kh_warp_affine_u8(s, bufu.ptr, outb.ptr, W, H, W, H, ch, aff, NU, n, n), 2 * n * NU)
    add(f"warp_perspective u8 c{ch}", lambda: lib.kh_warp_perspective_u8(s, bufu.ptr, outb.ptr, W, H, W, H, ch, hom, NU, n, n), 2 * n * NU)
    add(f"remap u8 bilinear c{ch}", lambda: lib.kh_remap_u8(s, bufu.ptr, dmx.ptr, dmy.ptr, outb.ptr, W, H, W, H, ch, 1, NU, n, n), 2 * n * NU)
    for mode, code in (("nearest", 0), ("bilinear", 1), ("bicubic", 2)):
        add(f"warp_affine f32 {mode} c{ch}", lambda: lib.kh_warp_affine_f32(s, buff.ptr, outb.ptr, W, H, W, H, ch, aff, code, NF, n, n), 8 * n * NF)
    add(f"warp_perspective f32 bilinear c{ch}", lambda: lib.kh_warp_perspective_f32(s, buff.ptr, outb.ptr, W, H, W, H, ch, hom, 1, NF, n, n), 8 * n * NF)
    add(f"remap f32 bilinear c{ch}", lambda: lib.kh_remap_f32(s, buff.ptr, dmx.ptr, dmy.ptr, outb.ptr, W, H, W, H, ch, 1, NF, n, n), 8 * n * NF)
    for mode, code in (("nearest", 0), ("bilinear", 1), ("bicubic", 2), ("lanczos", 3)):
        dn = (W // 2) * (H // 2) * ch
        add(f"resize f32 4K -> 1080p {mode} c{ch}", lambda: lib.kh_resize_f32(s, buff.ptr, outb.ptr, W, H, W // 2, H // 2, ch, code, NF, n, dn), 4 * (n + dn) * NF)
        add(f"resize_fast u8 4K -> 1080p {mode} c{ch}", lambda: lib.kh_resize_fast_u8(s, bufu.ptr, outb.ptr, W, H, W // 2, H // 2, ch, code, 1, NU, n, dn), (n + dn) * NU)
