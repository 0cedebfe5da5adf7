"""Dev (r04zs): how much of the 1080p -> 640 x 640 letterbox kernel's time is the padding?  Times, with events, the letterbox launch,
the same active area alone (stretch to 640 x 360) and a pure constant fill of the padded rows' bytes."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kornia-rs_amd")); sys.path.insert(0, str(ROOT))
import importlib.util
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from kornia_rs import hip, Preprocessor, Tensor
from kornia_rs.hip import DeviceBuffer

hip.set_device(0); st = hip.Stream.new(0)
N, W, H = 1024, 1920, 1080
fb = W * H * 3 // 2
base = bench.lcg_bytes(fb + 31 * N)
src = DeviceBuffer(N * fb, st, zeroed=False)
dbase = DeviceBuffer.from_numpy(base, st)
from kornia_rs.hip import lib, check
for k in range(N):
    check(lib.kh_memcpy_d2d_async(src.ptr + k * fb, dbase.ptr + 31 * k, fb, st.cuda_stream_ptr))
st.synchronize()

def timed(fn, reps=10):
    for _ in range(3): fn()
    st.synchronize()
    e0, e1 = hip.Event(timing=True), hip.Event(timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); st.synchronize()
    return e0.elapsed_ms(e1) / reps

kw = dict(format="nv12", mean=bench.IMAGENET_MEAN, std=bench.IMAGENET_STD, stream=st)
for mode, (ow, oh) in [("letterbox", (640, 640)), ("stretch", (640, 360)), ("letterbox", (608, 608)), ("stretch", (608, 342))]:
    pre = Preprocessor(mode=mode, **kw)
    dst = Tensor.uninit((N, 3, oh, ow), "float32", st)
    ms = timed(lambda: pre.run_raw_batch(src, W, H, dst, frame_stride=fb))
    print(f"{mode:9s} 1080p -> {ow}x{oh}: {ms:.3f} ms   ({N * 3 * oh * ow * 4 / 1e9:.2f} GB written)")
    del dst
