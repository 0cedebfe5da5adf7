import sys, ctypes as C
sys.path.insert(0, "kornia-rs_amd"); sys.path.insert(0, "tests")
import numpy as np
import oracle_ffi as O
from kornia_rs import _ffi, hip
from gpu_util import dev, out_buf, fptr
hip.set_device(0)
s = hip.Stream.new(0)
def img(w, h, c, seed=0):
    return np.roll(O.pattern_f32(w * h * c + seed), -seed)[: w * h * c].reshape(h, w, c).copy()
src = img(129, 97, 3)
def once(dw, dh, poison=True):
    d_src, d_dst = dev(s, src), out_buf(s, dh * dw * 12, poison)
    rc = _ffi.lib.kh_warp_affine_f32(s.cuda_stream_ptr, d_src.ptr, d_dst.ptr, 129, 97, dw, dh, 3, fptr([1, 0, 0, 0, 1, 0]), 0, 1, 0, 0)
    assert rc == 0
    got = d_dst.to_numpy(np.float32, (dh, dw, 3))
    want = O.warp_affine(src, [1, 0, 0, 0, 1, 0], dw, dh, "nearest")
    back = d_src.to_numpy(np.float32, src.shape)
    return int((got.view(np.uint32) != want.view(np.uint32)).sum()), int((back != src).sum()), hex(d_src.ptr), hex(d_dst.ptr)
for label in ("default threshold", "max threshold"):
    if label.startswith("max"):
        _ffi.check(_ffi.lib.kh_mempool_set_release_threshold(0, 2**64 - 1))
    print(label)
    for k in range(6):
        print("  ", once(129, 97), once(80, 120), once(80, 120, poison=False))
