"""The library's test options (kh_debug_set_option, include/kornia_hip.h): every alternate code path a launcher can be forced
onto gives the oracle's bytes on inputs that would otherwise take the production kernel.  These paths are the fallbacks other
geometries / alignments / channel counts use anyway; forcing them on RGB images of convenient sizes keeps them covered.
The library never reads the environment (round 3: 26 getenv knobs, several on launch paths)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits, dev, out_buf
from test_pyramid_morph_gpu import make, morph_gpu, pyr_gpu
from test_u8_gpu import blur_gpu, pat

pytestmark = pytest.mark.gpu


def test_unknown_option_is_a_typed_error(gpu_stream):
    from kornia_rs import _ffi
    assert _ffi.lib.kh_debug_set_option(b"no_such_option", 1) == _ffi.KH_ERR_INVALID_ARG
    assert "no_such_option" in _ffi.last_error()
    assert _ffi.lib.kh_debug_set_option(None, 1) == _ffi.KH_ERR_INVALID_ARG
    for name in (b"pre_ieee_div", b"pre_grid", b"pre_quads", b"filter_force_tile", b"filter_four_columns", b"grad_scalar", b"hfilter_direct",
                 b"resize_u8_gather", b"pyr_direct", b"pyr_roll", b"morph_direct", b"morph_roll", b"u8_blur_rgb", b"u8_blur_swar", b"warp_u8_direct", b"warp_u8_spans", b"warp_u8_rows", b"resize_rows", b"warp_f32_px", b"resize_u8_px", b"row_stores", b"pre_f16_lut"):
        assert _ffi.lib.kh_debug_set_option(name, -1) == _ffi.KH_OK, name


def test_the_environment_is_not_consulted(gpu_stream, monkeypatch):
    """A round-3 knob in the environment changes nothing: the four-tap fallback is reachable only through the option."""
    from kornia_rs import _ffi
    from test_preprocess_gpu import _pre, IMAGENET
    monkeypatch.setenv("KH_PRE_GRID", "0")
    pre = _pre(gpu_stream, mode="letterbox", format="nv12", sampling="bilinear", **IMAGENET)
    p = pre._params(1920, 1080, 1920, 1, _ffi.KH_FMT_NV12, 640, 640, 1, 0, False, False)
    assert _ffi.lib.kh_preprocess_variant(C.byref(p)) == b"generic_bilinear_on_grid"
    _ffi.check(_ffi.lib.kh_debug_set_option(b"pre_grid", 0))
    try:
        assert _ffi.lib.kh_preprocess_variant(C.byref(p)) == b"generic"
    finally:
        _ffi.lib.kh_debug_set_option(b"pre_grid", -1)


@pytest.mark.parametrize("option,value", [("pyr_direct", 1), ("pyr_roll", 0)])
def test_pyramid_fallbacks(gpu_stream, dev_option, option, value):
    dev_option(option, value)
    for (w, h) in [(129, 97), (520, 140), (64, 48)]:
        src = make(w, h, 3, np.uint8, seed=w)
        assert_same_bits(pyr_gpu(gpu_stream, src, False)[0], O.pyrdown(src), f"{option} pyrdown_u8 {w}x{h}")
        assert_same_bits(pyr_gpu(gpu_stream, src, True)[0], O.pyrup(src), f"{option} pyrup_u8 {w}x{h}")
    srcf = make(131, 67, 3, np.float32, seed=5)
    assert_same_bits(pyr_gpu(gpu_stream, srcf, True)[0], O.pyrup(srcf), f"{option} pyrup_f32")
    assert_same_bits(pyr_gpu(gpu_stream, srcf, False)[0], O.pyrdown(srcf), f"{option} pyrdown_f32")


@pytest.mark.parametrize("option,value", [("morph_direct", 1), ("morph_roll", 0)])
def test_morphology_fallbacks(gpu_stream, dev_option, option, value):
    dev_option(option, value)
    src = make(300, 121, 3, np.uint8, seed=9)
    for op in ("dilate", "erode"):
        for k in (3, 5, 7):
            mask = O.morph_kernel("box", k, k)
            got = morph_gpu(gpu_stream, src, op, mask, "replicate", [0, 0, 0])[0]
            assert_same_bits(got, O.morphology_u8(src, op, mask, "replicate", [0, 0, 0]), f"{option} {op} box{k}")


@pytest.mark.parametrize("option,value", [("u8_blur_rgb", 0), ("u8_blur_swar", 0)])
def test_u8_blur_fallbacks(gpu_stream, dev_option, option, value):
    dev_option(option, value)
    if option == "u8_blur_swar":
        dev_option("u8_blur_rgb", 0)   # the SWAR / plain-integer choice is made inside the interleaved kernel's launcher
    for c in (1, 3, 4):
        src = pat(261, 97, c, seed=c)
        for ksize, sigma in [((7, 7), (1.5, 1.5)), ((3, 3), (1.0, 1.0)), ((5, 9), (1.0, 2.0))]:
            got = blur_gpu(gpu_stream, "gaussian", src, ksize, sigma)[0]
            assert_same_bits(got, O.gaussian_blur_u8(src, ksize, sigma)[0], f"{option} c{c} {ksize}")


def test_gradient_and_hfilter_fallbacks(gpu_stream, dev_option):
    from kornia_rs import _ffi
    dev_option("grad_scalar", 1)
    dev_option("hfilter_direct", 1)
    h, w, c = 33, 100, 3
    img = O.pattern_f32(h * w * c).reshape(h, w, c)
    n = h * w * c
    d_src, d_gx, d_gy = dev(gpu_stream, img), out_buf(gpu_stream, 4 * n), out_buf(gpu_stream, 4 * n)
    _ffi.check(_ffi.lib.kh_spatial_gradient_f32(gpu_stream.cuda_stream_ptr, d_src.ptr, d_gx.ptr, d_gy.ptr, w, h, c, O.GRADIENT_KINDS["sobel"], 1, n, n))
    wx, wy = O.spatial_gradient(img, "sobel")
    assert_same_bits(d_gx.to_numpy(np.float32, (h, w, c)), wx, "grad_scalar dx")
    assert_same_bits(d_gy.to_numpy(np.float32, (h, w, c)), wy, "grad_scalar dy")
    d_dst, d_tmp = out_buf(gpu_stream, 4 * n), out_buf(gpu_stream, 4 * n)
    _ffi.check(_ffi.lib.kh_box_blur_fast_f32(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, d_tmp.ptr, w, h, c, 2.0, 2.0, 1, n, n))
    assert_same_bits(d_dst.to_numpy(np.float32, (h, w, c)), O.box_blur_fast(img, (2.0, 2.0)), "hfilter_direct box_blur_fast")
