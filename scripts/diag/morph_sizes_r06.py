#!/usr/bin/env python3
"""Round 6: u8 dilate over structuring-element shapes / sizes / borders / channels, 32 x 4K (sweep for slow fallbacks)."""
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
if len(sys.argv) > 1:   # e.g. morph_roll=2: the tile kernel instead of the rolling kernels (A/B)
    name, val = sys.argv[1].split("=")
    check(lib.kh_debug_set_option(name.encode(), int(val)))
    print(f"# dev option {name} = {val}")
N, W, H = 32, 3840, 2160
for ch in (3, 1, 4):
    n = W * H * ch
    src = DeviceBuffer.from_numpy(bench.lcg_bytes(N * n), st); dst = DeviceBuffer(N * n, st, zeroed=False)
    cval = (C.c_uint8 * 4)(0, 0, 0, 0)
    for shape, sname in ((0, "box"), (1, "cross"), (2, "ellipse")):
        for k in (3, 5, 7, 9, 15, 21, 31):
            for border in ((0, 1) if (shape == 0 and k == 5) else (0,)):
                mask = (C.c_uint8 * (k * k))()
                check(lib.kh_morph_kernel(shape, k, k, mask))
                fn = lambda: lib.kh_morphology_u8(st.cuda_stream_ptr, src.ptr, dst.ptr, W, H, ch, 0, mask, k, k, border, cval, N, n, n)
                rc = fn()
                if rc != 0:
                    print(f"c{ch} {sname} {k}: error {_ffi.last_error()[:70]}"); continue
                st.synchronize()
                ts = []
                for r in range(3):
                    e0, e1 = hip.Event(), hip.Event(); e0.record(st)
                    for _ in range(2):
                        fn()
                    e1.record(st); st.synchronize()
                    ts.append(e0.elapsed_ms(e1) / 2)
                t = float(np.median(ts))
                print(f"dilate c{ch} {sname:7s} {k:2d}x{k:<2d} border {border}: {t:8.3f} ms  frac {2 * n * N / t / 1e6 / 8000:.3f}")
    del src, dst
