#!/usr/bin/env python3
"""Round 6: f32 gaussian 5x5 over row lengths whose bytes are / are not multiples of 16 and of 128 (the streaming stores' line alignment); optional name=value dev option."""
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
N = 16
buff = DeviceBuffer(N * 3840 * 2160 * 4 * 4, st, zeroed=True)
outb = DeviceBuffer(N * 3840 * 2160 * 4 * 4, st, zeroed=False)
for ch, widths in ((1, (3840, 3839, 3836, 3808, 1920, 1000, 1001, 640)), (3, (3840, 3839, 3836, 1920, 1000, 1001, 1004, 640, 224))):
    for W in widths:
        H = 2160 if W > 2000 else 1080
        NB = N if W > 2000 else 4 * N
        n = W * H * ch
        for K in (5, 11):
            t = timeit(lambda: lib.kh_gaussian_blur_f32(s, buff.ptr, outb.ptr, W, H, ch, K, K, 1.5, 1.5, NB, n, n))
            print(f"gaussian f32 {K:2d} c{ch} {W:4d}x{H} x{NB}: {t:8.3f} ms  frac {8 * n * NB / t / 1e6 / 8000:.3f}  (row bytes % 128 = {W * ch * 4 % 128})")
