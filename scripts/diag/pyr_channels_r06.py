#!/usr/bin/env python3
"""Round 6: pyrdown / pyrup, u8 and f32, on 1- / 3- / 4-channel 4K images (32 per call); optional argument name=value sets a dev option (A/B)."""
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
if len(sys.argv) > 1:
    name, val = sys.argv[1].split("=")
    check(lib.kh_debug_set_option(name.encode(), int(val)))
    print(f"# dev option {name} = {val}")
N, W, H = 32, 3840, 2160
import os
for dt, es in ((("u8", 1),) if os.environ.get("U8_ONLY") else (("u8", 1), ("f32", 4))):
    for ch in (1, 3, 4):
        for up in (False, True):
            n = W * H * ch
            dw, dh = (2 * W, 2 * H) if up else (W // 2, H // 2)
            m = dw * dh * ch
            nb = N if not (up and es == 4) else 8
            raw = bench.lcg_bytes(nb * n * es)
            if es == 4:
                raw = (raw.view(np.uint32) >> 9 | 0x3f800000).view(np.float32).view(np.uint8)
            src = DeviceBuffer.from_numpy(raw, st); dst = DeviceBuffer(nb * m * es, st, zeroed=False)
            fn = getattr(lib, f"kh_{'pyrup' if up else 'pyrdown'}_{dt}")
            call = lambda: check(fn(st.cuda_stream_ptr, src.ptr, dst.ptr, W, H, ch, nb, n, m))
            call(); st.synchronize()
            ts = []
            for r in range(3):
                e0, e1 = hip.Event(), hip.Event(); e0.record(st)
                for _ in range(2):
                    call()
                e1.record(st); st.synchronize()
                ts.append(e0.elapsed_ms(e1) / 2)
            t = float(np.median(ts))
            print(f"{'pyrup  ' if up else 'pyrdown'} {dt:3s} c{ch} x{nb:2d}: {t:8.3f} ms  frac {(n + m) * es * nb / t / 1e6 / 8000:.3f}")
            del src, dst
