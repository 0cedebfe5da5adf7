#!/usr/bin/env python3
"""Round 6: the north star (NV12 1080p x 1024 -> CHW) into f32 and into f16 planes through Preprocessor.run_raw_batch, timed with HIP events."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kornia-rs_amd")); sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
from kornia_rs import Preprocessor, Tensor, hip
from kornia_rs.hip import DeviceBuffer, lib, check
import bench
hip.set_device(0); st = hip.Stream.new(0)
for arg in sys.argv[1:]:   # name=value dev options
    name, val = arg.split("=")
    check(lib.kh_debug_set_option(name.encode(), int(val)))
    print(f"# dev option {name} = {val}")
N, W, H = 1024, 1920, 1080
fb = W * H * 3 // 2
base = bench.lcg_bytes(fb + 31 * N)
dbase = DeviceBuffer.from_numpy(base, st)
src = DeviceBuffer(fb * N, st, zeroed=False)
for k in range(N):
    check(lib.kh_memcpy_d2d_async(src.ptr + k * fb, dbase.ptr + 31 * k, fb, st.cuda_stream_ptr))
for f16 in (False, True):
    dst = Tensor.uninit((N, 3, H, W), "float16" if f16 else "float32", st)
    pre = Preprocessor(mode="stretch", format="nv12", sampling="bilinear", f16=f16, mean=bench.IMAGENET_MEAN, std=bench.IMAGENET_STD, stream=st)
    ts = []
    for r in range(6):
        pre.run_raw_batch(src, W, H, dst, frame_stride=fb); st.synchronize()
        e0, e1 = hip.Event(), hip.Event(); e0.record(st)
        for _ in range(5):
            pre.run_raw_batch(src, W, H, dst, frame_stride=fb)
        e1.record(st); st.synchronize()
        if r:
            ts.append(e0.elapsed_ms(e1) / 5)
    nbytes = N * (fb + W * H * 3 * (2 if f16 else 4))
    print(f"f16={f16}: {np.median(ts):.3f} ms  {nbytes / np.median(ts) / 1e6:.0f} GB/s  frac {nbytes / np.median(ts) / 1e6 / 8000:.3f}")
    del dst
