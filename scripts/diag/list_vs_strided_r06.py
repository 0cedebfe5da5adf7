#!/usr/bin/env python3
"""Round 6 diagnostic: where does the C5 pointer-list row lose against the equally spaced one — in the LIST kernels, or in the
separately allocated images (pool allocations instead of one large buffer)?  Times, on one box:
  strided            kh_remap_f32 + kh_warp_perspective_f32 on three big buffers
  list / contiguous  the *_list entries fed pointers INTO those same big buffers
  list / separate    the *_list entries on separately allocated images (what bench.py's api_list row does)"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kornia-rs_amd"))
import torch  # noqa: F401  (one HIP runtime)
from kornia_rs import _ffi, hip
from kornia_rs.hip import DeviceBuffer

lib, check = _ffi.lib, _ffi.check
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W, H, Cc = 3840, 2160, 3
n = W * H * Cc
hip.set_device(0)
import os
for opt in filter(None, os.environ.get("KH_DIAG_OPTS", "").split(",")):   # e.g. KH_DIAG_OPTS=warp_f32_px=1
    name, _, value = opt.partition("=")
    check(lib.kh_debug_set_option(name.encode(), int(value)))
    print("# test option", opt)
st = hip.Stream.new(0)
s = st.cuda_stream_ptr
big = [DeviceBuffer(N * n * 4, st, zeroed=(k == 0)) for k in range(3)]
mx, my = DeviceBuffer(W * H * 4, st, zeroed=False), DeviceBuffer(W * H * 4, st, zeroed=False)
INTR = (577.48583984375 * 3.0, 652.8748779296875 * 3.0, 577.48583984375 * 2.7, 386.1428833007812 * 2.7)
DIST = (1.7547749280929563, 0.0097926277667284, -0.027250492945313457, 2.1092164516448975, 0.462927520275116,
        -0.08215277642011642, -0.00005535508171073161, 0.00003768636770639569)
check(lib.kh_correction_map_polynomial_f32(s, mx.ptr, my.ptr, W, H, (C.c_double * 4)(*INTR), (C.c_double * 8)(*DIST)))
w, h = float(W), float(H)
hm = (C.c_float * 9)(1.03, 0.05, -3.0 * w / 129.0, -0.02, 0.97, 4.0 * h / 97.0, 2.0 / (h * w), 1.5 / (w * h), 1.0)
sep = [[DeviceBuffer(n * 4, st, zeroed=(k == 0)) for _ in range(N)] for k in range(3)]
spacers = [DeviceBuffer(((5 * i) % 3 + 1) << 21, st, zeroed=False) for i in range(8)]
st.synchronize()


def arr(ptrs):
    return _ffi.pointer_array(ptrs)


cont = [arr([b.ptr + k * n * 4 for k in range(N)]) for b in big]
sepp = [arr([b.ptr for b in bufs]) for bufs in sep]


def strided():
    check(lib.kh_remap_f32(s, big[0].ptr, mx.ptr, my.ptr, big[1].ptr, W, H, W, H, Cc, 1, N, n, n))
    check(lib.kh_warp_perspective_f32(s, big[1].ptr, big[2].ptr, W, H, W, H, Cc, hm, 1, N, n, n))


def lists(p):
    def f():
        check(lib.kh_remap_f32_list(s, p[0], mx.ptr, my.ptr, p[1], N, W, H, W, H, Cc, 1))
        check(lib.kh_warp_perspective_f32_list(s, p[1], p[2], N, W, H, W, H, Cc, hm, 1))
    return f


def only(fn_name, *args):
    return lambda: check(getattr(lib, fn_name)(s, *args))


runs = {"strided (2 launches)": strided, "list / contiguous memory": lists(cont), "list / separate allocations": lists(sepp),
        "  remap strided": only("kh_remap_f32", big[0].ptr, mx.ptr, my.ptr, big[1].ptr, W, H, W, H, Cc, 1, N, n, n),
        "  remap list / contiguous": only("kh_remap_f32_list", cont[0], mx.ptr, my.ptr, cont[1], N, W, H, W, H, Cc, 1),
        "  remap list / separate": only("kh_remap_f32_list", sepp[0], mx.ptr, my.ptr, sepp[1], N, W, H, W, H, Cc, 1),
        "  warp_perspective strided": only("kh_warp_perspective_f32", big[1].ptr, big[2].ptr, W, H, W, H, Cc, hm, 1, N, n, n),
        "  warp_perspective list / contiguous": only("kh_warp_perspective_f32_list", cont[1], cont[2], N, W, H, W, H, Cc, hm, 1),
        "  warp_perspective list / separate": only("kh_warp_perspective_f32_list", sepp[1], sepp[2], N, W, H, W, H, Cc, hm, 1)}
times = {k: [] for k in runs}
for rnd in range(6):
    for name, fn in runs.items():
        fn(); st.synchronize()
        e0, e1 = hip.Event(), hip.Event()
        e0.record(st)
        for _ in range(3):
            fn()
        e1.record(st); st.synchronize()
        if rnd:
            times[name].append(e0.elapsed_ms(e1) / 3)
print(f"# C5 shape, N = {N} 4K f32x3 images, 5 interleaved rounds x 3 steps; ms per step (median, min)")
for name, v in times.items():
    print(f"{name:42s} {np.median(v):8.3f} {min(v):8.3f}")
