"""Helpers for GPU tests: move numpy arrays through the C ABI."""
import ctypes as C

import numpy as np


def dev(gpu_stream, a):
    from kornia_rs.hip import DeviceBuffer
    a = np.ascontiguousarray(a)
    return DeviceBuffer.from_numpy(a.reshape(-1), gpu_stream) if a.size else DeviceBuffer(16, gpu_stream)


def out_buf(gpu_stream, nbytes, poison=True):
    """Uninitialised-looking destination: filled with 0xFF so unwritten pixels are caught."""
    from kornia_rs import _ffi
    from kornia_rs.hip import DeviceBuffer
    b = DeviceBuffer(nbytes + 16, gpu_stream, zeroed=False)
    if poison:
        _ffi.check(_ffi.lib.kh_memset_async(b.ptr, 0xFF, nbytes + 16, gpu_stream.cuda_stream_ptr))
    return b


def fptr(values):
    arr = (C.c_float * len(values))(*[float(v) for v in values])
    return arr


def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_same_bits(got, want, what=""):
    g, w = bits(np.ascontiguousarray(got)), bits(np.ascontiguousarray(want))
    assert g.shape == w.shape, (g.shape, w.shape)
    bad = np.nonzero(g != w)
    if bad[0].size:
        idx = tuple(b[0] for b in bad)
        raise AssertionError(f"{what}: {bad[0].size} of {g.size} elements differ; first at {idx}: "
                             f"got {got[idx]!r} want {want[idx]!r}")
