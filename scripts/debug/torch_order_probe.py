"""Dev probe: does importing / initialising torch before large alloc-free cycles corrupt later results?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "kornia-rs_amd"))
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
from kornia_rs import _ffi, hip
from kornia_rs.hip import DeviceBuffer
import oracle_ffi as O
stream = hip.Stream.new(0)
if mode in ("import", "init", "tensor"):
    import torch
    if mode in ("init", "tensor"):
        torch.cuda.init(); torch.cuda.synchronize()
    if mode == "tensor":
        t = torch.ones(1 << 20, device="cuda"); torch.cuda.synchronize(); del t
r = np.arange(256, dtype=np.uint8)
s = np.stack(np.meshgrid(r, r, r, indexing="ij"), axis=-1).reshape(-1)
want = {n: O.color_map(n, s, c) for n, c in (("gray_from_rgb_u8", 1), ("sepia_from_rgb_u8", 3))}
def run(name, cout):
    dsrc = DeviceBuffer(s.nbytes + 64, stream); ddst = DeviceBuffer(s.size // 3 * cout + 64, stream)
    dsrc.copy_from_host(s, 0)
    _ffi.check(getattr(_ffi.lib, "kh_" + name)(stream.cuda_stream_ptr, dsrc.ptr, ddst.ptr, s.size // 3))
    return ddst.to_numpy(np.uint8, (s.size // 3 * cout,), 0)
for it in range(6):
    for name, cout in (("gray_from_rgb_u8", 1), ("sepia_from_rgb_u8", 3)):
        got = run(name, cout)
        bad = np.nonzero(got != want[name])[0]
        print(mode, it, name, "mismatches", bad.size, ("first %d last %d got %s want %s" % (bad[0], bad[-1], got[bad[:4]], want[name][bad[:4]])) if bad.size else "")
