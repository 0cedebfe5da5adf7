"""GPU parity for the colour family (C ABI -> HIP kernels) against the CPU oracle.
u8: bit-exact.  f32: bit-identical (same expression trees, no contraction) — asserted on the bit
pattern, which is stricter than the 1e-6 the north star asks for."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

SIZES = [258 * 195, 1, 3, 4, 5, 64 * 1024 + 3, 0]


def run(gpu_stream, name, src, cout, *extra, offset=0):
    """Call kh_<name>(stream, src, dst, npixels, *extra) on device copies; `offset` shifts both
    device pointers by that many bytes to exercise the unaligned path."""
    from kornia_rs import _ffi
    from kornia_rs.hip import DeviceBuffer
    src = np.ascontiguousarray(src).reshape(-1)
    cin = {"gray_from_rgb": 3, "rgb_from_gray": 1, "apply_colormap": 1, "rgb_from_rgba": 4}.get(
        name.rsplit("_", 1)[0], 3)
    n = src.size // cin
    item = src.dtype.itemsize
    dsrc = DeviceBuffer(src.nbytes + 64, gpu_stream)
    ddst = DeviceBuffer(n * cout * item + 64, gpu_stream)
    if src.nbytes:
        dsrc.copy_from_host(src, offset)
    rc = getattr(_ffi.lib, "kh_" + name)(gpu_stream.cuda_stream_ptr, dsrc.ptr + offset, ddst.ptr + offset, n, *extra)
    _ffi.check(rc)
    return ddst.to_numpy(src.dtype, (n * cout,), offset)


def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def rgb_u8(n, seed=0):
    return np.roll(O.pattern_u8(3 * n + seed), -seed)[: 3 * n].copy()


def rgb_f32_255(n):  # HSV/HLS domain: 0..255 floats, incl. exact greys and extremes
    return O.pattern_u8(3 * n).astype(np.float32)


@pytest.mark.parametrize("n", SIZES)
def test_gray_u8_f32(gpu_stream, n):
    s = rgb_u8(n)
    assert np.array_equal(run(gpu_stream, "gray_from_rgb_u8", s, 1), O.color_map("gray_from_rgb_u8", s, 1))
    f = O.pattern_f32(3 * n)
    assert np.array_equal(bits(run(gpu_stream, "gray_from_rgb_f32", f, 1)), bits(O.color_map("gray_from_rgb_f32", f, 1)))
    g = O.pattern_u8(n)
    assert np.array_equal(run(gpu_stream, "rgb_from_gray_u8", g, 3), np.repeat(g, 3))
    gf = O.pattern_f32(n)
    assert np.array_equal(run(gpu_stream, "rgb_from_gray_f32", gf, 3), np.repeat(gf, 3))


def test_gray_known_answer(gpu_stream):  # color/gray/mod.rs:395-412
    out = run(gpu_stream, "gray_from_rgb_u8", np.array([0, 128, 255, 128, 0, 128], np.uint8), 1)
    assert out.tolist() == [104, 53]


@pytest.mark.parametrize("offset", [1, 2, 3])
def test_unaligned_buffers_take_byte_path(gpu_stream, offset):
    s = rgb_u8(1001)
    assert np.array_equal(run(gpu_stream, "gray_from_rgb_u8", s, 1, offset=offset), O.color_map("gray_from_rgb_u8", s, 1))
    assert np.array_equal(run(gpu_stream, "bgr_from_rgb_u8", s, 3, offset=offset), O.color_map("bgr_from_rgb_u8", s, 3))


@pytest.mark.parametrize("n", [258 * 195, 7, 0])
@pytest.mark.parametrize("order", [0, 1])
def test_ycc_family_a(gpu_stream, n, order):
    s = rgb_u8(n)
    for name in ("ycc_from_rgb_u8", "rgb_from_ycc_u8"):
        assert np.array_equal(run(gpu_stream, name, s, 3, order), O.color_map(name, s, 3, order)), name
    f = O.pattern_f32(3 * n)
    for name in ("ycc_from_rgb_f32", "rgb_from_ycc_f32"):
        assert np.array_equal(bits(run(gpu_stream, name, f, 3, order)), bits(O.color_map(name, f, 3, order))), name


def test_ycc_exhaustive_u8(gpu_stream):
    """All 2^24 RGB triples through the Q14 forward+inverse kernels, both chroma orders."""
    r = np.arange(256, dtype=np.uint8)
    s = np.stack(np.meshgrid(r, r, r, indexing="ij"), axis=-1).reshape(-1)
    for order in (0, 1):
        for name in ("ycc_from_rgb_u8", "rgb_from_ycc_u8"):
            assert np.array_equal(run(gpu_stream, name, s, 3, order), O.color_map(name, s, 3, order)), (name, order)
    assert np.array_equal(run(gpu_stream, "gray_from_rgb_u8", s, 1), O.color_map("gray_from_rgb_u8", s, 1))
    got, want = run(gpu_stream, "sepia_from_rgb_u8", s, 3), O.color_map("sepia_from_rgb_u8", s, 3)
    bad = np.nonzero(got != want)[0]
    if bad.size:  # where and what: holes of stale bytes point at the copy path, not the kernel
        again = run(gpu_stream, "sepia_from_rgb_u8", s, 3)
        raise AssertionError(f"sepia: {bad.size} mismatches in [{bad[0]}, {bad[-1]}], got {got[bad[:6]]} want {want[bad[:6]]}; "
                             f"a second device run differs from the oracle in {int((again != want).sum())} places")


@pytest.mark.parametrize("n", [258 * 195, 5, 0])
def test_hsv_hls(gpu_stream, n):
    f = rgb_f32_255(n)
    for name in ("hsv_from_rgb_f32", "hls_from_rgb_f32"):
        got, want = run(gpu_stream, name, f, 3), O.color_map(name, f, 3)
        assert np.array_equal(bits(got), bits(want)), name
    # inverse directions on valid HSV/HLS triples produced by the forward oracle
    for fwd, inv in (("hsv_from_rgb_f32", "rgb_from_hsv_f32"), ("hls_from_rgb_f32", "rgb_from_hls_f32")):
        h = O.color_map(fwd, f, 3)
        got, want = run(gpu_stream, inv, h, 3), O.color_map(inv, h, 3)
        assert np.array_equal(bits(got), bits(want)), inv
        if n:
            assert np.abs(want - f).max() < 1e-2  # the pair really inverts


@pytest.mark.parametrize("n", [258 * 195, 6, 0])
def test_swizzles_sepia_colormap(gpu_stream, n):
    from kornia_rs.hip import DeviceBuffer
    s, f = rgb_u8(n), O.pattern_f32(3 * n)
    assert np.array_equal(run(gpu_stream, "bgr_from_rgb_u8", s, 3), O.color_map("bgr_from_rgb_u8", s, 3))
    assert np.array_equal(run(gpu_stream, "bgr_from_rgb_f32", f, 3), O.color_map("bgr_from_rgb_f32", f, 3))
    for swap in (0, 1):
        assert np.array_equal(run(gpu_stream, "rgba_from_rgb_u8", s, 4, swap), O.color_map("rgba_from_rgb_u8", s, 4, swap))
        assert np.array_equal(run(gpu_stream, "rgba_from_rgb_f32", f, 4, swap), O.color_map("rgba_from_rgb_f32", f, 4, swap))
        rgba = O.pattern_u8(4 * n)
        assert np.array_equal(run(gpu_stream, "rgb_from_rgba_u8", rgba, 3, swap, None),
                              O.color_map("rgb_from_rgba_u8", rgba, 3, swap, None))
        bg = (C.c_uint8 * 3)(100, 50, 200)
        assert np.array_equal(run(gpu_stream, "rgb_from_rgba_u8", rgba, 3, swap, C.cast(bg, C.c_void_p)),
                              O.color_map("rgb_from_rgba_u8", rgba, 3, swap, C.cast(bg, C.c_void_p)))
    got, want = run(gpu_stream, "sepia_from_rgb_u8", s, 3), O.color_map("sepia_from_rgb_u8", s, 3)
    bad = np.nonzero(got != want)[0]
    if bad.size:  # where and what: holes of stale bytes point at the copy path, not the kernel
        again = run(gpu_stream, "sepia_from_rgb_u8", s, 3)
        raise AssertionError(f"sepia: {bad.size} mismatches in [{bad[0]}, {bad[-1]}], got {got[bad[:6]]} want {want[bad[:6]]}; "
                             f"a second device run differs from the oracle in {int((again != want).sum())} places")
    assert np.array_equal(bits(run(gpu_stream, "sepia_from_rgb_f32", f, 3)), bits(O.color_map("sepia_from_rgb_f32", f, 3)))
    lut = np.roll(O.pattern_u8(768 + 5), -5)[:768].copy()
    dlut = DeviceBuffer.from_numpy(lut, gpu_stream)
    g = O.pattern_u8(n)
    assert np.array_equal(run(gpu_stream, "apply_colormap_u8", g, 3, dlut.ptr), O.color_map("apply_colormap_u8", g, 3, lut))


def test_rgb_from_rgba_known(gpu_stream):  # color/rgb/mod.rs tests: drop alpha, verified with opencv
    src = np.array([0, 1, 2, 255, 3, 4, 5, 255, 6, 7, 8, 255, 9, 10, 11, 255, 12, 13, 14, 255, 15, 16, 17, 255], np.uint8)
    assert run(gpu_stream, "rgb_from_rgba_u8", src, 3, 0, None).tolist() == list(range(18))
    assert run(gpu_stream, "rgb_from_rgba_u8", src, 3, 1, None).tolist() == [2, 1, 0, 5, 4, 3, 8, 7, 6, 11, 10, 9, 14, 13, 12, 17, 16, 15]


def video(gpu_stream, name, src, out_bytes, w, h, *extra):
    from kornia_rs import _ffi
    from kornia_rs.hip import DeviceBuffer
    src = np.ascontiguousarray(src, np.uint8).reshape(-1)
    dsrc = DeviceBuffer.from_numpy(src, gpu_stream) if src.size else DeviceBuffer(16, gpu_stream)
    ddst = DeviceBuffer(out_bytes + 16, gpu_stream)
    _ffi.check(getattr(_ffi.lib, "kh_" + name)(gpu_stream.cuda_stream_ptr, dsrc.ptr, ddst.ptr, w, h, *extra))
    return ddst.to_numpy(np.uint8, (out_bytes,))


@pytest.mark.parametrize("w,h", [(64, 6), (70, 4), (1920, 1080), (2, 2), (6, 2)])
@pytest.mark.parametrize("layout", [0, 1, 2, 3])
def test_planar420_decode(gpu_stream, w, h, layout):
    raw = np.roll(O.pattern_u8(w * h * 3 // 2 + 9), -9)[: w * h * 3 // 2].copy()
    got = video(gpu_stream, "rgb_from_planar420_u8", raw, w * h * 3, w, h, layout)
    assert np.array_equal(got, O.rgb_from_nv12(raw, w, h, layout).reshape(-1))


@pytest.mark.parametrize("w,h", [(70, 3), (1920, 1080), (2, 1)])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_packed422_decode(gpu_stream, w, h, layout):
    raw = O.pattern_u8(w * h * 2)
    got = video(gpu_stream, "rgb_from_packed422_u8", raw, w * h * 3, w, h, layout)
    assert np.array_equal(got, O.rgb_from_yuyv(raw, w, h, layout).reshape(-1))


def test_known_video_answers(gpu_stream):
    # packed422_known_gray (color/yuv/kernels.rs:2068) and the NV12 4x4 reference case (:2078)
    assert video(gpu_stream, "rgb_from_packed422_u8", np.array([16, 128, 16, 128], np.uint8), 6, 2, 1, 0).tolist() == [0] * 6
    w = h = 4
    y = np.array([(v * 9 + 16) & 0xFF for v in range(16)], np.uint8)
    uv = np.array([(v * 5 + 100) & 0xFF for v in range(8)], np.uint8)
    raw = np.concatenate([y, uv])
    assert np.array_equal(video(gpu_stream, "rgb_from_planar420_u8", raw, 48, w, h, 0), O.rgb_from_nv12(raw, w, h).reshape(-1))


@pytest.mark.parametrize("w,h", [(70, 4), (1920, 1080), (2, 2)])
def test_video_encoders(gpu_stream, w, h):
    rgb = O.pattern_u8(w * h * 3).reshape(h, w, 3)
    assert np.array_equal(video(gpu_stream, "nv12_from_rgb_u8", rgb, w * h * 3 // 2, w, h), O.nv12_from_rgb(rgb))
    assert np.array_equal(video(gpu_stream, "yuyv_from_rgb_u8", rgb, w * h * 2, w, h), O.yuyv_from_rgb(rgb))


def test_video_validation(gpu_stream):
    from kornia_rs import _ffi
    lib = _ffi.lib
    assert lib.kh_rgb_from_planar420_u8(None, C.c_void_p(64), C.c_void_p(64), 7, 4, 0) == _ffi.KH_ERR_INVALID_ARG
    assert lib.kh_rgb_from_planar420_u8(None, C.c_void_p(64), C.c_void_p(64), 8, 5, 0) == _ffi.KH_ERR_INVALID_ARG
    assert lib.kh_rgb_from_planar420_u8(None, C.c_void_p(64), C.c_void_p(64), 8, 4, 9) == _ffi.KH_ERR_INVALID_ARG
    assert lib.kh_rgb_from_packed422_u8(None, C.c_void_p(64), C.c_void_p(64), 7, 4, 0) == _ffi.KH_ERR_INVALID_ARG
    assert lib.kh_ycc_from_rgb_u8(None, C.c_void_p(64), C.c_void_p(64), 4, 2) == _ffi.KH_ERR_INVALID_ARG
    assert lib.kh_gray_from_rgb_u8(None, None, None, 4) == _ffi.KH_ERR_INVALID_ARG
    assert lib.kh_gray_from_rgb_u8(None, None, None, 0) == _ffi.KH_OK
