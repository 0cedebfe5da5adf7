#!/usr/bin/env python3
"""Round 6: u8 gaussian / box over kernel sizes on 1- and 4-channel 4K images (32 / 16 per call): sweep for cliffs."""
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
W, H = 3840, 2160
for ch, N in ((1, 32), (4, 16), (3, 16)):
    n = W * H * ch
    src = DeviceBuffer.from_numpy(bench.lcg_bytes(N * n), st); dst = DeviceBuffer(N * n, st, zeroed=False)
    for K in (3, 5, 7, 9, 11, 13, 15, 17, 21, 31):
        sig = 0.3 * ((K - 1) * 0.5 - 1) + 0.8
        for name, fn in (("gaussian", lambda: lib.kh_gaussian_blur_u8(s, src.ptr, dst.ptr, W, H, ch, K, K, sig, sig, N, n, n)),
                         ("box     ", lambda: lib.kh_box_blur_u8(s, src.ptr, dst.ptr, W, H, ch, K, K, N, n, n))):
            rc = fn()
            if rc != 0:
                print(f"{name} u8 c{ch} {K}: error {_ffi.last_error()[:70]}"); continue
            st.synchronize(); ts = []
            for r in range(3):
                e0, e1 = hip.Event(), hip.Event(); e0.record(st)
                for _ in range(2):
                    fn()
                e1.record(st); st.synchronize(); ts.append(e0.elapsed_ms(e1) / 2)
            t = float(np.median(ts))
            print(f"{name} u8 c{ch} {K:2d}x{K:<2d} x{N}: {t:8.3f} ms  frac {2 * n * N / t / 1e6 / 8000:.3f}")
    del src, dst
