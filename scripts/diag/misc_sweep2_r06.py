#!/usr/bin/env python3
"""Round 6: second sweep — CIE conversions, the fused RGB8 -> CHW resize over modes / sizes, Preprocessor sampling modes (256 x 1080p)."""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kornia-rs_amd")); sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
from kornia_rs import _ffi, hip, Preprocessor, Tensor
from kornia_rs.hip import DeviceBuffer
import bench
lib, check = _ffi.lib, _ffi.check
hip.set_device(0); st = hip.Stream.new(0)
s = st.cuda_stream_ptr
def timeit(fn):
    rc = fn()
    if rc not in (0, None):
        return float("nan")
    st.synchronize(); ts = []
    for r in range(3):
        e0, e1 = hip.Event(), hip.Event(); e0.record(st)
        for _ in range(2):
            fn()
        e1.record(st); st.synchronize(); ts.append(e0.elapsed_ms(e1) / 2)
    return float(np.median(ts))
N, W, H = 256, 1920, 1080
npx = W * H * N
buff = DeviceBuffer(npx // 4 * 12, st, zeroed=True); outf = DeviceBuffer(npx // 4 * 12, st, zeroed=False)
for conv, name in enumerate(["linear_from_rgb", "rgb_from_linear", "xyz_from_rgb", "rgb_from_xyz", "lab_from_rgb", "rgb_from_lab", "luv_from_rgb", "rgb_from_luv"]):
    t = timeit(lambda: lib.kh_cie_convert_f32(s, buff.ptr, outf.ptr, npx // 4, conv))
    print(f"cie {name:18s}: {t:7.3f} ms  frac {npx // 4 * 24 / t / 1e6 / 8000:.3f}")
del buff, outf
srcu = DeviceBuffer.from_numpy(bench.lcg_bytes(npx * 3), st)
scale = (C.c_float * 3)(0.017, 0.0175, 0.0174); bias = (C.c_float * 3)(-2.1, -2.0, -1.8)
for (dw, dh) in ((224, 224), (640, 360), (960, 540), (1280, 720)):
    out = DeviceBuffer(N * 3 * dw * dh * 4, st, zeroed=False)
    for mode, code in (("nearest", 0), ("bilinear", 1), ("bicubic", 2), ("lanczos", 3)):
        t = timeit(lambda: lib.kh_resize_normalize_to_chw_u8_f32(s, srcu.ptr, out.ptr, W, H, dw, dh, scale, bias, code, 1, N, W * H * 3, 3 * dw * dh))
        print(f"rgb8 -> chw f32 {dw}x{dh} {mode:8s}: {t:7.3f} ms   (src + dst at 6.4 TB/s: {N * (W * H * 3 + dw * dh * 12) / 6.4e9:.3f} ms)")
    del out
del srcu
fb = W * H * 3 // 2
src = DeviceBuffer.from_numpy(bench.lcg_bytes(fb * N), st)
for sampling in ("nearest", "bilinear", "lanczos"):
    for mode, (ow, oh) in (("letterbox", (640, 640)), ("stretch", (224, 224))):
        dst = Tensor.uninit((N, 3, oh, ow), "float32", st)
        pre = Preprocessor(mode=mode, format="nv12", sampling=sampling, mean=bench.IMAGENET_MEAN, std=bench.IMAGENET_STD, stream=st)
        t = timeit(lambda: pre.run_raw_batch(src, W, H, dst, frame_stride=fb))
        print(f"preprocess nv12 {mode} -> {ow}x{oh} {sampling:8s}: {t:7.3f} ms")
        del dst
