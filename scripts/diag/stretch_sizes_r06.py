#!/usr/bin/env python3
"""Round 6: NV12 / YUYV 1080p x 1024 stretched / letterboxed to common network input sizes (Preprocessor.run_raw_batch), f32 and f16."""
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
N, W, H = 1024, 1920, 1080
fmt = sys.argv[1] if len(sys.argv) > 1 else "nv12"
fb = W * H * 3 // 2 if fmt == "nv12" else W * H * 2
base = bench.lcg_bytes(fb + 31 * N)
dbase = DeviceBuffer.from_numpy(base, st)
src = DeviceBuffer(fb * N, st, zeroed=False)
for k in range(N):
    check(lib.kh_memcpy_d2d_async(src.ptr + k * fb, dbase.ptr + 31 * k, fb, st.cuda_stream_ptr))
for mode, (ow, oh) in (("stretch", (224, 224)), ("stretch", (384, 384)), ("stretch", (512, 512)), ("stretch", (640, 640)), ("letterbox", (320, 320)), ("letterbox", (416, 416)),
                       ("letterbox", (1280, 1280)), ("stretch", (960, 540)), ("stretch", (1280, 720))):
    for f16 in (False, True):
        dst = Tensor.uninit((N, 3, oh, ow), "float16" if f16 else "float32", st)
        pre = Preprocessor(mode=mode, format=fmt, sampling="bilinear", f16=f16, mean=bench.IMAGENET_MEAN, std=bench.IMAGENET_STD, stream=st)
        ts = []
        for r in range(5):
            pre.run_raw_batch(src, W, H, dst, frame_stride=fb); st.synchronize()
            e0, e1 = hip.Event(), hip.Event(); e0.record(st)
            for _ in range(5):
                pre.run_raw_batch(src, W, H, dst, frame_stride=fb)
            e1.record(st); st.synchronize()
            if r:
                ts.append(e0.elapsed_ms(e1) / 5)
        out_b = N * ow * oh * 3 * (2 if f16 else 4)
        print(f"{fmt} {mode:9s} -> {ow}x{oh} f16={f16}: {np.median(ts):.3f} ms   (output alone at 6.4 TB/s: {out_b / 6.4e9:.3f} ms)")
        del dst
