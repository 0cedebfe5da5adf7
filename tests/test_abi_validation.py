"""Argument validation of every compute entry happens BEFORE any HIP call: on this GPU-less host each bad call must
come back with its typed status (and a message naming the entry), never KH_ERR_HIP — the error contract of the
reference's adapters (typed ImageError before launch; P/cuda/dispatch.rs:166-211, P/warp/perspective.rs:41-60,
P/filter/ops.rs:120-125, P/resize/cuda.rs:436-470).  Device pointers are fake non-null values: a validation
failure never dereferences or launches."""
import ctypes as C

import pytest

from kornia_rs import _ffi

L = _ffi.lib
S = None
P, Q, R3 = 0x10000, 0x2000000, 0x3000000  # fake device addresses (distinct: src != dst)
M6 = (C.c_float * 6)(1, 0, 0, 0, 1, 0)
SING = (C.c_float * 9)(1, 2, 3, 2, 4, 6, 3, 6, 9)
K3 = (C.c_float * 3)(0.25, 0.5, 0.25)
MASK = (C.c_uint8 * 9)(*[1] * 9)
D4, D8 = (C.c_double * 4)(), (C.c_double * 8)()
PA = lambda *ptrs: _ffi.pointer_array(list(ptrs))  # noqa: E731  a host void*[n] of fake device addresses
INVALID, UNSUPPORTED, TOO_LARGE, SINGULAR = _ffi.KH_ERR_INVALID_ARG, _ffi.KH_ERR_UNSUPPORTED, _ffi.KH_ERR_TOO_LARGE, _ffi.KH_ERR_SINGULAR

CASES = [
    # ---- f32 geometry
    ("resize: zero-sized dst", lambda: L.kh_resize_f32(S, P, Q, 8, 8, 0, 8, 3, 1, 1, 0, 0), INVALID, "zero-sized"),
    ("resize: 2 channels", lambda: L.kh_resize_f32(S, P, Q, 8, 8, 4, 4, 2, 1, 1, 0, 0), UNSUPPORTED, "2 channels"),
    ("resize: unknown mode", lambda: L.kh_resize_f32(S, P, Q, 8, 8, 4, 4, 3, 7, 1, 0, 0), UNSUPPORTED, "mode 7"),
    ("resize: null src", lambda: L.kh_resize_f32(S, None, Q, 8, 8, 4, 4, 3, 1, 1, 0, 0), INVALID, "null"),
    ("resize: negative stride", lambda: L.kh_resize_f32(S, P, Q, 8, 8, 4, 4, 3, 1, 2, -1, 0), INVALID, "stride"),
    ("resize: batch > 65535", lambda: L.kh_resize_f32(S, P, Q, 8, 8, 4, 4, 3, 1, 70000, 0, 0), TOO_LARGE, "batch"),
    ("resize: > 2^31 elements", lambda: L.kh_resize_f32(S, P, Q, 8, 8, 50000, 50000, 3, 1, 1, 0, 0), TOO_LARGE, "32-bit"),
    ("resize_mapped: unknown mapping", lambda: L.kh_resize_mapped_f32(S, P, Q, 8, 8, 4, 4, 3, 1, 5, 1, 0, 0), INVALID, "mapping 5"),
    ("resize_normalize_f32: zero std", lambda: L.kh_resize_bilinear_normalize_f32(S, P, Q, 8, 8, 4, 4, K3, (C.c_float * 3)(1, 0, 1), 0, 1, 0, 0), INVALID, "non-zero"),
    ("resize_normalize_f32: null mean", lambda: L.kh_resize_bilinear_normalize_f32(S, P, Q, 8, 8, 4, 4, None, K3, 0, 1, 0, 0), INVALID, "mean"),
    ("resize_normalize_f32: unknown mapping", lambda: L.kh_resize_bilinear_normalize_f32(S, P, Q, 8, 8, 4, 4, K3, K3, 2, 1, 0, 0), INVALID, "mapping 2"),
    ("warp_affine: null matrix", lambda: L.kh_warp_affine_f32(S, P, Q, 8, 8, 8, 8, 3, None, 1, 1, 0, 0), INVALID, "matrix"),
    ("warp_perspective: singular", lambda: L.kh_warp_perspective_f32(S, P, Q, 8, 8, 8, 8, 3, SING, 1, 1, 0, 0), SINGULAR, "determinant"),
    ("remap: null map", lambda: L.kh_remap_f32(S, P, None, P, Q, 8, 8, 8, 8, 3, 1, 1, 0, 0), INVALID, "map"),
    ("correction map: zero-sized", lambda: L.kh_correction_map_polynomial_f32(S, P, Q, 0, 4, D4, D8), INVALID, "zero-sized"),
    # ---- f32 filters
    ("gaussian: even kernel", lambda: L.kh_gaussian_blur_f32(S, P, Q, 8, 8, 3, 4, 3, 1.0, 1.0, 1, 0, 0), INVALID, "sigma"),
    ("gaussian: nothing to resolve", lambda: L.kh_gaussian_blur_f32(S, P, Q, 8, 8, 3, 0, 0, 0.0, 0.0, 1, 0, 0), INVALID, "sigma"),
    ("gaussian: > 63 taps", lambda: L.kh_gaussian_blur_f32(S, P, Q, 8, 8, 3, 65, 3, 1.0, 1.0, 1, 0, 0), UNSUPPORTED, "63"),
    ("gaussian: in place", lambda: L.kh_gaussian_blur_f32(S, P, P, 8, 8, 3, 3, 3, 1.0, 1.0, 1, 0, 0), INVALID, "in-place"),
    ("box: zero kernel", lambda: L.kh_box_blur_f32(S, P, Q, 8, 8, 3, 0, 3, 1, 0, 0), INVALID, "kernel length"),
    ("box: > 63 taps", lambda: L.kh_box_blur_f32(S, P, Q, 8, 8, 3, 65, 3, 1, 0, 0), UNSUPPORTED, "63"),
    ("separable: null kernel", lambda: L.kh_separable_filter_f32(S, P, Q, 8, 8, 3, None, 3, K3, 3, 1, 0, 0), INVALID, "kernel length"),
    ("separable: empty kernel", lambda: L.kh_separable_filter_f32(S, P, Q, 8, 8, 3, K3, 3, K3, 0, 1, 0, 0), INVALID, "kernel length"),
    ("gradient: unknown kind", lambda: L.kh_gradient_magnitude_f32(S, P, Q, 8, 8, 3, 9, 3, 1, 0, 0), INVALID, "kind 9"),
    ("gradient: sobel size 4", lambda: L.kh_gradient_magnitude_f32(S, P, Q, 8, 8, 3, 0, 4, 1, 0, 0), INVALID, "kernel length"),
    # ---- pointer-list batches (round 6): host arrays of device pointers, checked before any launch
    ("resize_list: null lists", lambda: L.kh_resize_f32_list(S, None, None, 2, 8, 8, 4, 4, 3, 1, 0), INVALID, "null pointer list"),
    ("resize_list: null entry", lambda: L.kh_resize_f32_list(S, PA(P, 0), PA(Q, R3), 2, 8, 8, 4, 4, 3, 1, 0), INVALID, "list index 1"),
    ("resize_list: src == dst", lambda: L.kh_resize_f32_list(S, PA(P, Q), PA(R3, Q), 2, 8, 8, 4, 4, 3, 1, 0), INVALID, "alias"),
    ("resize_list: > 65535 images", lambda: L.kh_resize_f32_list(S, PA(P), PA(Q), 70000, 8, 8, 4, 4, 3, 1, 0), TOO_LARGE, "65535"),
    ("resize_list: 2 channels", lambda: L.kh_resize_f32_list(S, PA(P), PA(Q), 1, 8, 8, 4, 4, 2, 1, 0), UNSUPPORTED, "2 channels"),
    ("resize_list: unknown mapping", lambda: L.kh_resize_f32_list(S, PA(P), PA(Q), 1, 8, 8, 4, 4, 3, 1, 5), INVALID, "mapping 5"),
    ("resize_normalize_list: zero std", lambda: L.kh_resize_bilinear_normalize_f32_list(S, PA(P), PA(Q), 1, 8, 8, 4, 4, K3, (C.c_float * 3)(1, 0, 1), 0), INVALID, "non-zero"),
    ("warp_affine_list: null matrix", lambda: L.kh_warp_affine_f32_list(S, PA(P), PA(Q), 1, 8, 8, 8, 8, 3, None, 1), INVALID, "matrix"),
    ("warp_perspective_list: singular", lambda: L.kh_warp_perspective_f32_list(S, PA(P), PA(Q), 1, 8, 8, 8, 8, 3, SING, 1), SINGULAR, "determinant"),
    ("remap_list: null map", lambda: L.kh_remap_f32_list(S, PA(P), None, P, PA(Q), 1, 8, 8, 8, 8, 3, 1), INVALID, "map"),
    ("gaussian_list: even kernel", lambda: L.kh_gaussian_blur_f32_list(S, PA(P), PA(Q), 1, 8, 8, 3, 4, 3, 1.0, 1.0), INVALID, "sigma"),
    ("gaussian_list: in place", lambda: L.kh_gaussian_blur_f32_list(S, PA(P), PA(P), 1, 8, 8, 3, 3, 3, 1.0, 1.0), INVALID, "alias"),
    ("box_list: > 63 taps", lambda: L.kh_box_blur_f32_list(S, PA(P), PA(Q), 1, 8, 8, 3, 65, 3), UNSUPPORTED, "63"),
    ("separable_list: empty kernel", lambda: L.kh_separable_filter_f32_list(S, PA(P), PA(Q), 1, 8, 8, 3, K3, 3, K3, 0), INVALID, "kernel length"),
    ("gradient_list: sobel size 4", lambda: L.kh_gradient_magnitude_f32_list(S, PA(P), PA(Q), 1, 8, 8, 3, 0, 4), INVALID, "kernel length"),
    # ---- the rest of the filter module
    ("spatial_gradient: unknown kind", lambda: L.kh_spatial_gradient_f32(S, P, Q, R3, 8, 8, 3, 4, 1, 0, 0), INVALID, "kind 4"),
    ("spatial_gradient: dx aliases src", lambda: L.kh_spatial_gradient_f32(S, P, P, Q, 8, 8, 3, 0, 1, 0, 0), INVALID, "distinct"),
    ("spatial_gradient: zero-sized", lambda: L.kh_spatial_gradient_f32(S, P, Q, R3, 0, 8, 3, 0, 1, 0, 0), INVALID, "zero-sized"),
    ("fast_horizontal_filter: half >= cols", lambda: L.kh_fast_horizontal_filter_f32(S, P, Q, 8, 8, 3, 8, 1, 0, 0), INVALID, "does not fit"),
    ("box_blur_fast: sigma too wide", lambda: L.kh_box_blur_fast_f32(S, P, Q, R3, 8, 8, 3, 6.0, 0.5, 1, 0, 0), INVALID, "do not fit"),
    ("box_blur_fast: null scratch", lambda: L.kh_box_blur_fast_f32(S, P, Q, None, 8, 8, 3, 0.5, 0.5, 1, 0, 0), INVALID, "null"),
    ("box_blur_fast_kernels: null out", lambda: L.kh_box_blur_fast_kernels_1d(1.0, 3, None), INVALID, "bad argument"),
    ("median: ksize 4", lambda: L.kh_median_blur_u8(S, P, Q, 8, 8, 3, 4, 1, 0, 0), INVALID, "kernel length 4"),
    ("median: ksize 7", lambda: L.kh_median_blur_u8(S, P, Q, 8, 8, 3, 7, 1, 0, 0), INVALID, "kernel length 7"),
    ("median: 5 channels", lambda: L.kh_median_blur_u8(S, P, Q, 8, 8, 5, 3, 1, 0, 0), UNSUPPORTED, "5 channels"),
    ("median: in place", lambda: L.kh_median_blur_u8(S, P, P, 8, 8, 3, 3, 1, 0, 0), INVALID, "aliased"),
    ("bilateral: zero-sized", lambda: L.kh_bilateral_filter_u8(S, P, Q, 0, 8, 5, 50.0, 50.0, 1, 0, 0), INVALID, "zero-sized"),
    ("bilateral: null dst", lambda: L.kh_bilateral_filter_u8(S, P, None, 8, 8, 5, 50.0, 50.0, 1, 0, 0), INVALID, "null"),
    ("bilateral: absurd radius", lambda: L.kh_bilateral_filter_u8(S, P, Q, 8, 8, 0, 50.0, 1e9, 1, 0, 0), TOO_LARGE, "radius"),
    ("bilateral_tables: absurd radius", lambda: L.kh_bilateral_tables(4000, 50.0, 50.0, 0, None, C.byref(C.c_int32()), None, None, None, None, None), TOO_LARGE, "radius"),
    ("bilateral_tables: null ntaps", lambda: L.kh_bilateral_tables(5, 50.0, 50.0, 0, None, None, None, None, None, None, None), INVALID, "ntaps"),
    # ---- u8 fixed-point twins
    ("gaussian_u8: 2 channels", lambda: L.kh_gaussian_blur_u8(S, P, Q, 8, 8, 2, 3, 3, 1.0, 1.0, 1, 0, 0), UNSUPPORTED, "2 channels"),
    ("box_u8: even kernel", lambda: L.kh_box_blur_u8(S, P, Q, 8, 8, 3, 2, 3, 1, 0, 0), INVALID, "odd"),
    ("remap_u8: bicubic", lambda: L.kh_remap_u8(S, P, P, P, Q, 8, 8, 8, 8, 3, 2, 1, 0, 0), UNSUPPORTED, "mode 2"),
    ("warp_affine_u8: 5 channels", lambda: L.kh_warp_affine_u8(S, P, Q, 8, 8, 8, 8, 5, M6, 1, 0, 0), UNSUPPORTED, "1, 2, 3, 4"),
    ("warp_affine_u8: negative stride", lambda: L.kh_warp_affine_u8(S, P, Q, 8, 8, 8, 8, 3, M6, 2, -5, 0), INVALID, "stride"),
    ("warp_perspective_u8: singular", lambda: L.kh_warp_perspective_u8(S, P, Q, 8, 8, 8, 8, 3, SING, 1, 0, 0), SINGULAR, "determinant"),
    ("resize_fast_u8: 2ch bilinear", lambda: L.kh_resize_fast_u8(S, P, Q, 8, 8, 4, 4, 2, 1, 1, 1, 0, 0), UNSUPPORTED, "channel count 2"),
    ("resize_fast_u8: 1-px-wide bilinear", lambda: L.kh_resize_fast_u8(S, P, Q, 1, 16, 8, 8, 3, 1, 1, 1, 0, 0), INVALID, "2x2"),
    ("resize_normalize: null scale", lambda: L.kh_resize_normalize_to_chw_u8_f32(S, P, Q, 8, 8, 4, 4, None, None, 1, 1, 1, 0, 0), INVALID, "scale"),
    ("resize_opencv_u8: bicubic", lambda: L.kh_resize_opencv_u8(S, P, Q, 8, 8, 4, 4, 3, 2, 1, 0, 0), UNSUPPORTED, "mode 2"),
    ("resize_opencv_f32: zero-sized", lambda: L.kh_resize_opencv_f32(S, P, Q, 0, 8, 4, 4, 3, 1, 1, 0, 0), INVALID, "zero-sized"),
    # ---- pyramid / morphology
    ("pyrdown_u8: 2 channels", lambda: L.kh_pyrdown_u8(S, P, Q, 8, 8, 2, 1, 0, 0), UNSUPPORTED, "2 channels"),
    ("pyrdown_f32: zero-sized", lambda: L.kh_pyrdown_f32(S, P, Q, 0, 8, 3, 1, 0, 0), INVALID, "zero-sized"),
    ("pyrup_u8: > 2^31 bytes", lambda: L.kh_pyrup_u8(S, P, Q, 40000, 40000, 3, 1, 0, 0), TOO_LARGE, "32-bit"),
    ("morphology: unknown op", lambda: L.kh_morphology_u8(S, P, Q, 8, 8, 3, 7, MASK, 3, 3, 0, None, 1, 0, 0), INVALID, "op 7"),
    ("morphology: 33-wide element", lambda: L.kh_morphology_u8(S, P, Q, 8, 8, 3, 0, MASK, 33, 3, 0, None, 1, 0, 0), UNSUPPORTED, "32x32"),
    ("morphology: null element", lambda: L.kh_morphology_u8(S, P, Q, 8, 8, 3, 0, None, 3, 3, 1, None, 1, 0, 0), INVALID, "structuring"),
    # ---- colour / pointwise
    ("gray: null", lambda: L.kh_gray_from_rgb_u8(S, None, Q, 10), INVALID, "null"),
    ("gray: negative count", lambda: L.kh_gray_from_rgb_u8(S, P, Q, -1), INVALID, "negative"),
    ("ycc: unknown chroma order", lambda: L.kh_ycc_from_rgb_u8(S, P, Q, 10, 7), INVALID, "order 7"),
    ("cie: unknown conversion", lambda: L.kh_cie_convert_f32(S, P, Q, 10, 99), INVALID, "conversion 99"),
    ("color f64: unknown conversion", lambda: L.kh_color_convert_f64(S, P, Q, 10, 18), INVALID, "conversion 18"),
    ("color f64: null", lambda: L.kh_color_convert_f64(S, None, Q, 10, 8), INVALID, "null"),
    ("color f64: negative count", lambda: L.kh_color_convert_f64(S, P, Q, -3, 8), INVALID, "negative"),
    ("yuyv mode decode: unknown mode", lambda: L.kh_yuyv_to_rgb_mode_u8(S, P, Q, 8, 4, 3), INVALID, "mode 3"),
    ("yuyv mode decode: null", lambda: L.kh_yuyv_to_rgb_mode_u8(S, None, Q, 8, 4, 0), INVALID, "null"),
    ("yuyv mode decode: > 2^31 bytes", lambda: L.kh_yuyv_to_rgb_mode_u8(S, P, Q, 40000, 40000, 0), TOO_LARGE, "32-bit"),
    ("bayer: unknown pattern", lambda: L.kh_rgb_from_bayer_u8(S, P, Q, 8, 4, 4), INVALID, "pattern 4"),
    ("bayer: null", lambda: L.kh_rgb_from_bayer_u8(S, P, None, 8, 4, 0), INVALID, "null"),
    ("planar 4:2:0: odd width", lambda: L.kh_rgb_from_planar420_u8(S, P, Q, 7, 4, 0), INVALID, "even"),
    ("planar 4:2:0: unknown layout", lambda: L.kh_rgb_from_planar420_u8(S, P, Q, 8, 4, 9), INVALID, "layout 9"),
    ("packed 4:2:2: odd width", lambda: L.kh_rgb_from_packed422_u8(S, P, Q, 7, 4, 0), INVALID, "even"),
    ("nv12_from_rgb: odd height", lambda: L.kh_nv12_from_rgb_u8(S, P, Q, 8, 5), INVALID, "even"),
    ("colormap: null LUT", lambda: L.kh_apply_colormap_u8(S, P, Q, 10, None), INVALID, "LUT"),
    ("normalize: 5 channels", lambda: L.kh_normalize_mean_std_f32(S, P, Q, 10, 5, K3, K3), UNSUPPORTED, "5 channels"),
    ("normalize: null mean", lambda: L.kh_normalize_mean_std_f32(S, P, Q, 10, 3, None, K3), INVALID, "mean"),
    ("normalize_rgb_u8: null scale", lambda: L.kh_normalize_rgb_u8_f32(S, P, Q, 10, None, None), INVALID, "scale"),
    ("find_min_max: empty image", lambda: L.kh_find_min_max_f32(S, P, 0, Q, Q), INVALID, "empty image"),
    ("find_min_max: null outputs", lambda: L.kh_find_min_max_f32(S, P, 10, None, None), INVALID, "null"),
    ("crop: window outside", lambda: L.kh_crop(S, P, Q, 8, 8, 4, 4, 6, 6, 3), INVALID, "out of bounds"),
    ("crop: negative origin", lambda: L.kh_crop(S, P, Q, 8, 8, 4, 4, -1, 0, 3), INVALID, "negative"),
    ("flip: negative width", lambda: L.kh_flip(S, P, Q, -1, 8, 3, 1), INVALID, "geometry"),
    # ---- runtime: graphs
    ("graph capture: default stream", lambda: L.kh_graph_capture_begin(None), INVALID, "non-default stream"),
    ("graph capture end: null out", lambda: L.kh_graph_capture_end(C.c_void_p(P), None), INVALID, "null"),
    ("graph launch: null graph", lambda: L.kh_graph_launch(None, None), INVALID, "null graph"),
    ("mem_get_info: null outputs", lambda: L.kh_mem_get_info(None, None), INVALID, "null"),
    # ---- fused pipelines
    ("fused pipeline: no stages", lambda: L.kh_fused_pipeline_build(None, 3, 8, 8, 1, 0, C.byref(C.c_void_p())), INVALID, "stage"),
]


@pytest.mark.parametrize("name,call,code,fragment", CASES, ids=[c[0] for c in CASES])
def test_rejected_before_any_device_work(name, call, code, fragment):
    rc = call()
    assert rc != _ffi.KH_ERR_HIP, f"{name}: reached the HIP runtime ({_ffi.last_error()})"
    assert rc == code, f"{name}: rc {rc}, message {_ffi.last_error()!r}"
    assert fragment in _ffi.last_error(), _ffi.last_error()


def test_empty_batches_and_images_are_no_ops_without_a_device():
    """batch == 0 / npixels == 0 returns KH_OK before touching the runtime (empty inputs are legal, null pointers allowed)."""
    assert L.kh_resize_f32(S, None, None, 8, 8, 4, 4, 3, 1, 0, 0, 0) == 0
    assert L.kh_warp_affine_u8(S, None, None, 8, 8, 8, 8, 3, M6, 0, 0, 0) == 0
    assert L.kh_gaussian_blur_u8(S, None, None, 8, 8, 3, 3, 3, 1.0, 1.0, 0, 0, 0) == 0
    assert L.kh_resize_fast_u8(S, None, None, 8, 8, 4, 4, 3, 1, 1, 0, 0, 0) == 0
    assert L.kh_pyrdown_u8(S, None, None, 8, 8, 3, 0, 0, 0) == 0
    assert L.kh_gray_from_rgb_u8(S, None, None, 0) == 0
    assert L.kh_color_convert_f64(S, None, None, 0, 12) == 0
    assert L.kh_yuyv_to_rgb_mode_u8(S, None, None, 1, 9, 0) == 0  # width 1: no whole pixel pair
    assert L.kh_flip(S, P, Q, 0, 8, 3, 1) == 0
    assert L.kh_graph_destroy(None) == 0
    assert L.kh_rgb_from_bayer_u8(S, None, None, 0, 7, 1) == 0
    assert L.kh_spatial_gradient_f32(S, None, None, None, 8, 8, 3, 1, 0, 0, 0) == 0
    assert L.kh_median_blur_u8(S, None, None, 8, 8, 3, 5, 0, 0, 0) == 0
    assert L.kh_bilateral_filter_u8(S, None, None, 8, 8, 5, 50.0, 50.0, 0, 0, 0) == 0
    assert L.kh_box_blur_fast_f32(S, None, None, None, 8, 8, 3, 0.5, 0.5, 0, 0, 0) == 0
