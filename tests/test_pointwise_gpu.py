"""GPU parity: normalize_mean_std / normalize_min_max / find_min_max / normalize_rgb_u8 / crop / flip
(P/normalize.rs, P/crop.rs, P/flip.rs).  Checked against numpy restatements of the reference's
scalar expressions (f32, same operation order) — bit-exact."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits, dev, fptr, out_buf

pytestmark = pytest.mark.gpu
f32 = np.float32


def lib_s(gpu_stream):
    from kornia_rs import _ffi
    return _ffi, _ffi.lib, gpu_stream.cuda_stream_ptr


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("n", [258 * 195, 260 * 196, 1920 * 1080, 1028, 5, 4, 0])   # 3 n % 4 == 0: the four-floats-per-lane kernel (round 5)
def test_normalize_mean_std(gpu_stream, c, n):
    _ffi, lib, s = lib_s(gpu_stream)
    src = O.pattern_f32(n * c)
    mean = np.array([0.485, 0.456, 0.406, 0.5][:c], f32)
    std = np.array([0.229, 0.224, 0.225, 0.25][:c], f32)
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, src.nbytes)
    _ffi.check(lib.kh_normalize_mean_std_f32(s, d_src.ptr, d_dst.ptr, n, c, fptr(mean), fptr(std)))
    want = ((src.reshape(-1, c) - mean) / std).astype(f32).reshape(-1)
    assert_same_bits(d_dst.to_numpy(f32, (n * c,)), want, "normalize_mean_std")


def test_normalize_mean_std_unaligned_buffers_take_the_per_pixel_kernel(gpu_stream):
    """A source or destination that is not 16-byte aligned cannot use the four-floats-per-lane kernel: same values either way."""
    _ffi, lib, s = lib_s(gpu_stream)
    n, c = 260 * 196, 3
    src = O.pattern_f32(n * c + 8)
    mean, std = np.array([0.485, 0.456, 0.406], f32), np.array([0.229, 0.224, 0.225], f32)
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, src.nbytes)
    for so, do in ((4, 0), (0, 4), (4, 4), (8, 12)):
        _ffi.check(lib.kh_normalize_mean_std_f32(s, d_src.ptr + so, d_dst.ptr + do, n, c, fptr(mean), fptr(std)))
        want = ((src[so // 4: so // 4 + n * c].reshape(-1, c) - mean) / std).astype(f32).reshape(-1)
        assert_same_bits(d_dst.to_numpy(f32, (n * c + 8,))[do // 4: do // 4 + n * c], want, f"offsets {so} / {do}")


def test_normalize_mean_std_reference_example(gpu_stream):  # normalize.rs doc example / tests
    _ffi, lib, s = lib_s(gpu_stream)
    src = np.array([0, 1, 0, 1, 2, 3, 0, 1, 0, 1, 2, 3], f32)
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, src.nbytes)
    _ffi.check(lib.kh_normalize_mean_std_f32(s, d_src.ptr, d_dst.ptr, 4, 3, fptr([0.5, 1.0, 0.5]), fptr([1.0, 1.0, 1.0])))
    assert d_dst.to_numpy(f32, (12,)).tolist() == [-0.5, 0.0, -0.5, 0.5, 1.0, 2.5, -0.5, 0.0, -0.5, 0.5, 1.0, 2.5]


def test_normalize_rgb_u8(gpu_stream):
    """Pixel counts that are / are not multiples of four (four pixels per lane with 16-byte stores since round 6 / one pixel per lane),
    a source 1 byte and a destination 4 bytes off alignment (the one-pixel kernel), a count that leaves a partial last block."""
    _ffi, lib, s = lib_s(gpu_stream)
    scale = (f32(1.0) / (np.array([0.229, 0.224, 0.225], f32) * f32(255.0))).astype(f32)
    offset = (-np.array([0.485, 0.456, 0.406], f32) / np.array([0.229, 0.224, 0.225], f32)).astype(f32)
    for n in (258 * 195, 258 * 195 + 2, 4, 1, 1024 * 4 + 4, 1920 * 1080):
        src = O.pattern_u8(3 * n + 1)
        want = (src[:3 * n].reshape(-1, 3).astype(f32) * scale + offset).astype(f32).reshape(-1)
        d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, 12 * n + 16)
        _ffi.check(lib.kh_normalize_rgb_u8_f32(s, d_src.ptr, d_dst.ptr, n, fptr(scale), fptr(offset)))
        assert_same_bits(d_dst.to_numpy(f32, (3 * n + 4,))[:3 * n], want, f"normalize_rgb_u8 n={n}")
        want1 = (src[1:3 * n + 1].reshape(-1, 3).astype(f32) * scale + offset).astype(f32).reshape(-1)
        _ffi.check(lib.kh_normalize_rgb_u8_f32(s, d_src.ptr + 1, d_dst.ptr + 4, n, fptr(scale), fptr(offset)))
        assert_same_bits(d_dst.to_numpy(f32, (3 * n + 4,))[1:3 * n + 1], want1, f"normalize_rgb_u8 n={n}, unaligned")


def test_min_max_and_normalize_min_max(gpu_stream):
    _ffi, lib, s = lib_s(gpu_stream)
    rng = np.random.default_rng(5)
    src = (rng.standard_normal(3 * 640 * 480) * 7).astype(f32)
    src[1234] = np.nan  # NaNs lose every comparison in the reference loop
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, src.nbytes)
    d_mm, d_scr = out_buf(gpu_stream, 8), out_buf(gpu_stream, 8)
    _ffi.check(lib.kh_find_min_max_f32(s, d_src.ptr, src.size, d_mm.ptr, d_scr.ptr))
    mm = d_mm.to_numpy(f32, (2,))
    assert mm[0] == np.nanmin(src) and mm[1] == np.nanmax(src)
    _ffi.check(lib.kh_normalize_min_max_f32(s, d_src.ptr, d_dst.ptr, src.size, 0.0, 1.0, d_mm.ptr, d_scr.ptr))
    lo, hi = f32(0.0), f32(1.0)
    want = ((src - mm[0]) * (hi - lo) / (mm[1] - mm[0]) + lo).astype(f32)
    got = d_dst.to_numpy(f32, src.shape)
    ok = ~np.isnan(src)
    assert_same_bits(got[ok], want[ok], "normalize_min_max")
    # reference doc example: [0,1,0,1,2,3,...] -> min 0, max 3
    ex = np.array([0, 1, 0, 1, 2, 3, 0, 1, 0, 1, 2, 3], f32)
    d_ex = dev(gpu_stream, ex)  # named: a temporary would be released (stream-ordered free) BEFORE the launch is enqueued
    _ffi.check(lib.kh_find_min_max_f32(s, d_ex.ptr, ex.size, d_mm.ptr, d_scr.ptr))
    assert d_mm.to_numpy(f32, (2,)).tolist() == [0.0, 3.0]
    assert lib.kh_find_min_max_f32(s, d_src.ptr, 0, d_mm.ptr, d_scr.ptr) == _ffi.KH_ERR_INVALID_ARG  # ImageDataNotInitialized


@pytest.mark.parametrize("dtype,c", [(np.uint8, 1), (np.uint8, 3), (np.uint8, 4), (np.float32, 1), (np.float32, 3)])
def test_crop_and_flip(gpu_stream, dtype, c):
    _ffi, lib, s = lib_s(gpu_stream)
    w, h = 67, 43
    src = (O.pattern_u8(w * h * c).astype(dtype)).reshape(h, w, c)
    pb = c * np.dtype(dtype).itemsize
    d_src = dev(gpu_stream, src)
    for (x, y, cw, ch) in [(1, 1, 20, 10), (0, 0, w, h), (47, 33, 20, 10), (5, 7, 1, 1)]:
        d_dst = out_buf(gpu_stream, cw * ch * pb)
        _ffi.check(lib.kh_crop(s, d_src.ptr, d_dst.ptr, w, h, cw, ch, x, y, pb))
        assert np.array_equal(d_dst.to_numpy(dtype, (ch, cw, c)), src[y:y + ch, x:x + cw])
    assert lib.kh_crop(s, d_src.ptr, d_src.ptr, w, h, 20, 10, 48, 0, pb) == _ffi.KH_ERR_INVALID_ARG  # PixelIndexOutOfBounds
    for horizontal, want in [(1, src[:, ::-1]), (0, src[::-1])]:
        d_dst = out_buf(gpu_stream, src.nbytes)
        _ffi.check(lib.kh_flip(s, d_src.ptr, d_dst.ptr, w, h, pb, horizontal))
        assert np.array_equal(d_dst.to_numpy(dtype, src.shape), want)


def test_crop_reference_example(gpu_stream):  # crop.rs doc example
    _ffi, lib, s = lib_s(gpu_stream)
    src = np.arange(16, dtype=np.uint8)
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, 4)
    _ffi.check(lib.kh_crop(s, d_src.ptr, d_dst.ptr, 4, 4, 2, 2, 1, 1, 1))
    assert d_dst.to_numpy(np.uint8, (4,)).tolist() == [5, 6, 9, 10]
