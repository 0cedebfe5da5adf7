import sys
sys.path.insert(0, "kornia-rs_amd"); sys.path.insert(0, "tests")
import numpy as np
import oracle_ffi as O
from kornia_rs import _ffi, hip
from gpu_util import dev, out_buf
hip.set_device(0)
s = hip.Stream.new(0)
def img(w, h, c, seed=0):
    return np.roll(O.pattern_f32(w * h * c + seed), -seed)[: w * h * c].reshape(h, w, c).copy()
def run(src, batch):
    h, w, c = src.shape[-3:]
    d_src, d_dst = dev(s, src), out_buf(s, src.nbytes)
    rc = _ffi.lib.kh_gaussian_blur_f32(s.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, c, 7, 7, 1.5, 1.5, batch, h*w*c, h*w*c)
    print("rc", rc, _ffi.last_error() if rc else "")
    return d_dst.to_numpy(np.float32, src.shape)
small = np.stack([img(300, 130, 3, seed=k) for k in range(2)])
for tag, arr, b in [("b1", small[0], 1), ("b2", small, 2), ("b1 again", small[1], 1)]:
    got = run(arr, b)
    want = np.stack([O.gaussian_blur(x, (7, 7), (1.5, 1.5)) for x in (arr if arr.ndim == 4 else [arr])]).reshape(got.shape)
    print(tag, "nan", np.isnan(got).sum(), "zeros", (got == 0).sum(), "mismatch", (got.view(np.uint32) != want.view(np.uint32)).sum(), "of", got.size)
big = img(3840, 2160, 3)
got = run(big, 1)
print("4k mismatch", (got.view(np.uint32) != O.gaussian_blur(big, (7,7),(1.5,1.5)).view(np.uint32)).sum())
got = run(small, 2)
want = np.stack([O.gaussian_blur(x, (7, 7), (1.5, 1.5)) for x in small])
print("after 4k b2: nan", np.isnan(got).sum(), "zeros", (got == 0).sum(), "mismatch", (got.view(np.uint32) != want.view(np.uint32)).sum())
