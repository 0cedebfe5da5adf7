#!/usr/bin/env python3
"""Round 6: BASELINE configs[1] (bilinear 1920x1080 -> 224x224 f32x3, N = 256) under the launcher's test options, interleaved in ONE
process: gather kernel (resize_rows=0) against the row-streamed kernel at several part widths / block sizes (resize_rows=N)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kornia-rs_amd"))
import torch  # noqa: F401
from kornia_rs import _ffi, hip
from kornia_rs.hip import DeviceBuffer

lib, check = _ffi.lib, _ffi.check
N, SW, SH, DW, DH, C = 256, 1920, 1080, 224, 224, 3
opts = [int(v) for v in sys.argv[1:]] or [0, -1, 32, 64, 96, 224, 2032, 1032]
hip.set_device(0)
st = hip.Stream.new(0)
src = DeviceBuffer(N * SW * SH * C * 4, st, zeroed=True)
dsts = [DeviceBuffer(N * DW * DH * C * 4, st, zeroed=False) for _ in range(4)]
times = {o: [] for o in opts}
turn = 0
for rnd in range(9):
    for o in opts:
        check(lib.kh_debug_set_option(b"resize_rows", o))
        def step():
            global turn
            turn += 1
            check(lib.kh_resize_f32(st.cuda_stream_ptr, src.ptr, dsts[turn % 4].ptr, SW, SH, DW, DH, C, 1, N, SW * SH * C, DW * DH * C))
        for _ in range(5):
            step()
        st.synchronize()
        e0, e1 = hip.Event(), hip.Event()
        e0.record(st)
        for _ in range(20):
            step()
        e1.record(st); st.synchronize()
        if rnd:
            times[o].append(e0.elapsed_ms(e1) / 20)
floor = 2796552192
print("# configs[1], 8 interleaved rounds x 20 launches; option: 0 = gather kernel, -1 = launcher's choice, N = N columns per part (+1000: 64-thread, +2000: 128-thread blocks)")
for o in opts:
    v = times[o]
    print(f"resize_rows={o:5d}  median {np.median(v):.4f}  min {min(v):.4f} ms   floor-bytes rate {floor / np.median(v) / 1e6:.0f} GB/s = {floor / np.median(v) / 1e6 / 8000:.3f} of peak")
