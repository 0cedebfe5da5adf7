"""ctypes binding of ``libkornia_hip.so`` — the C ABI declared in ``include/kornia_hip.h``.

This is the Python twin of the thin ``extern "C"`` FFI crate the Rust host would use
(INTEGRATION.md).  There is deliberately no fallback: if the shared library is missing or a
symbol does not resolve, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("KORNIA_HIP_LIB", _HERE.parent / "lib" / "libkornia_hip.so"))

# status codes (kornia_hip.h)
KH_OK = 0
KH_ERR_INVALID_ARG = -1
KH_ERR_HIP = -2
KH_ERR_UNSUPPORTED = -3
KH_ERR_TOO_LARGE = -4
KH_ERR_SINGULAR = -5
KH_ERR_SLICE_TOO_SMALL = -6

KH_DOMAIN_HOST, KH_DOMAIN_DEVICE, KH_DOMAIN_UNIFIED, KH_DOMAIN_HOST_PINNED = 0, 1, 2, 3

KH_FMT_RGB, KH_FMT_BGR, KH_FMT_GRAY, KH_FMT_NV12, KH_FMT_YUYV = 0, 1, 2, 3, 4
KH_SAMPLE_NEAREST, KH_SAMPLE_BILINEAR, KH_SAMPLE_LANCZOS = 0, 1, 2
KH_OUT_F32, KH_OUT_F16 = 0, 1
KH_PRE_FORCE_GENERIC = 1
KH_YCC_YCRCB, KH_YCC_YUV = 0, 1
KH_INTERP_NEAREST, KH_INTERP_BILINEAR, KH_INTERP_BICUBIC, KH_INTERP_LANCZOS = 0, 1, 2, 3
KH_MORPH_DILATE, KH_MORPH_ERODE = 0, 1
KH_CIE = {"linear_rgb_from_rgb": 0, "rgb_from_linear_rgb": 1, "xyz_from_rgb": 2, "rgb_from_xyz": 3, "lab_from_rgb": 4,
          "rgb_from_lab": 5, "luv_from_rgb": 6, "rgb_from_luv": 7}
KH_FUSE_READ_U8RGB_BILINEAR, KH_FUSE_NORMALIZE, KH_FUSE_RGB_TO_GRAY, KH_FUSE_WRITE_CHW_F32, KH_FUSE_WRITE_C1_F32 = 1, 16, 17, 32, 33


class FusedStage(C.Structure):
    """kh_fused_stage (include/kornia_hip.h)."""
    _fields_ = [("kind", C.c_int32), ("u", C.c_int32 * 4), ("f", C.c_float * 6)]
KH_BORDER = {"constant": 0, "replicate": 1, "reflect101": 2, "reflect": 3, "wrap": 4}
KH_MORPH_SHAPE = {"box": 0, "cross": 1, "ellipse": 2}
KH_PIXEL_MAPPING = {"half_pixel": 0, "align_corners": 1}  # PixelMapping, P/cuda/resize.rs:438-454
KH_BAYER = {"rggb": 0, "bggr": 1, "grbg": 2, "gbrg": 3}  # BayerPattern, I/color_spaces.rs:853-862
KH_YUV_MODE = {"bt601_full": 0, "bt709_full": 1, "bt601_limited": 2}  # YuvToRgbMode, P/color/yuv/mod.rs:319-327
# kh_color_convert_f64 codes: 0..7 = KH_CIE, then the gray / hsv / hls / YCbCr / YUV f64 twins
KH_F64 = {**KH_CIE, "gray_from_rgb": 8, "rgb_from_gray": 9, "hsv_from_rgb": 10, "rgb_from_hsv": 11, "hls_from_rgb": 12,
          "rgb_from_hls": 13, "ycbcr_from_rgb": 14, "rgb_from_ycbcr": 15, "yuv_from_rgb": 16, "rgb_from_yuv": 17}
KH_GRAD_SOBEL, KH_GRAD_SCHARR = 0, 1


class KorniaHipError(RuntimeError):
    """A failed C-ABI call; ``code`` is the KH_ERR_* status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


class PreprocessParams(C.Structure):
    """``kh_preprocess_params`` (include/kornia_hip.h)."""

    _fields_ = [
        ("scale_x", C.c_float), ("scale_y", C.c_float), ("pad_x", C.c_float), ("pad_y", C.c_float),
        ("src_w", C.c_int32), ("src_h", C.c_int32), ("src_pitch", C.c_int32), ("src_bpp", C.c_int32),
        ("fmt", C.c_int32), ("dst_w", C.c_int32), ("dst_h", C.c_int32),
        ("mean", C.c_float * 3), ("inv_std", C.c_float * 3), ("pad_value", C.c_float),
        ("sampling", C.c_int32), ("out_dtype", C.c_int32), ("nframes", C.c_int32),
        ("flags", C.c_int32), ("src_frame_stride", C.c_int64), ("dst_frame_stride", C.c_int64),
    ]


_vp, _i32, _u64, _sz, _f32 = C.c_void_p, C.c_int32, C.c_uint64, C.c_size_t, C.c_float
_i64 = C.c_int64
_f64 = C.c_double
_P = C.POINTER

# name -> (restype, argtypes); every symbol kornia_hip.h declares must appear here
# (tests/test_abi.py cross-checks the header against this table and the built library).
SIGNATURES = {
    "kh_last_error": (_sz, [C.c_char_p, _sz]),
    "kh_debug_fast_quot": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "kh_debug_set_option": (_i32, [C.c_char_p, _i32]),
    "kh_version": (C.c_char_p, []),
    "kh_hip_runtime_images": (_i32, [C.c_char_p, C.c_size_t]),
    "kh_hip_versions": (_i32, [_P(_i32), _P(_i32)]),
    "kh_dlpack_noop_deleter": (None, [_vp]),
    "kh_device_count": (_i32, [_P(_i32)]),
    "kh_set_device": (_i32, [_i32]),
    "kh_get_device": (_i32, [_P(_i32)]),
    "kh_device_info": (_i32, [_i32, C.c_char_p, _sz, _P(_i32), _P(_u64)]),
    "kh_mem_get_info": (_i32, [_P(_u64), _P(_u64)]),
    "kh_graph_capture_begin": (_i32, [_vp]),
    "kh_graph_capture_end": (_i32, [_vp, _P(_vp)]),
    "kh_graph_launch": (_i32, [_vp, _vp]),
    "kh_graph_destroy": (_i32, [_vp]),
    "kh_stream_create": (_i32, [_P(_vp)]),
    "kh_stream_destroy": (_i32, [_vp]),
    "kh_stream_synchronize": (_i32, [_vp]),
    "kh_stream_wait_event": (_i32, [_vp, _vp]),
    "kh_stream_set_workspace": (_i32, [_vp, _vp, _sz]),
    "kh_last_workspace_bytes": (_i32, [_P(_sz)]),
    "kh_stream_workspace_bytes": (_i32, [_vp, _P(_sz)]),
    "kh_event_create": (_i32, [_P(_vp), _i32]),
    "kh_event_destroy": (_i32, [_vp]),
    "kh_event_record": (_i32, [_vp, _vp]),
    "kh_event_synchronize": (_i32, [_vp]),
    "kh_event_elapsed_ms": (_i32, [_vp, _vp, _P(_f32)]),
    "kh_stream_fence": (_i32, [_vp, _vp]),
    "kh_malloc_async": (_i32, [_P(_vp), _sz, _i32, _vp]),
    "kh_free_async": (_i32, [_vp, _vp]),
    "kh_mempool_set_release_threshold": (_i32, [_i32, _u64]),
    "kh_host_alloc": (_i32, [_P(_vp), _sz]),
    "kh_host_free": (_i32, [_vp]),
    "kh_malloc_managed": (_i32, [_P(_vp), _sz]),
    "kh_free": (_i32, [_vp]),
    "kh_memcpy_h2d_async": (_i32, [_vp, _vp, _sz, _vp]),
    "kh_memcpy_d2h_async": (_i32, [_vp, _vp, _sz, _vp]),
    "kh_memcpy_d2d_async": (_i32, [_vp, _vp, _sz, _vp]),
    "kh_memset_async": (_i32, [_vp, _i32, _sz, _vp]),
    "kh_pointer_domain": (_i32, [_vp, _P(_i32), _P(_i32)]),
    "kh_preprocess_to_chw": (_i32, [_vp, _vp, _vp, _P(PreprocessParams)]),
    "kh_preprocess_to_chw_list": (_i32, [_vp, _P(_vp), _vp, _P(PreprocessParams)]),
    "kh_preprocess_variant": (C.c_char_p, [_P(PreprocessParams)]),
    # colour
    **{n: (_i32, [_vp, _vp, _vp, _i64]) for n in (
        "kh_gray_from_rgb_u8", "kh_gray_from_rgb_f32", "kh_rgb_from_gray_u8", "kh_rgb_from_gray_f32",
        "kh_bgr_from_rgb_u8", "kh_bgr_from_rgb_f32", "kh_hsv_from_rgb_f32", "kh_rgb_from_hsv_f32",
        "kh_hls_from_rgb_f32", "kh_rgb_from_hls_f32", "kh_sepia_from_rgb_u8", "kh_sepia_from_rgb_f32")},
    **{n: (_i32, [_vp, _vp, _vp, _i64, _i32]) for n in (
        "kh_rgba_from_rgb_u8", "kh_rgba_from_rgb_f32", "kh_ycc_from_rgb_u8", "kh_rgb_from_ycc_u8",
        "kh_ycc_from_rgb_f32", "kh_rgb_from_ycc_f32")},
    "kh_cie_convert_f32": (_i32, [_vp, _vp, _vp, _i64, _i32]),
    "kh_color_convert_f64": (_i32, [_vp, _vp, _vp, _i64, _i32]),
    "kh_yuyv_to_rgb_mode_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32]),
    "kh_rgb_from_bayer_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32]),
    "kh_rgb_from_rgba_u8": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "kh_apply_colormap_u8": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "kh_rgb_from_planar420_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32]),
    "kh_rgb_from_packed422_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32]),
    "kh_nv12_from_rgb_u8": (_i32, [_vp, _vp, _vp, _i32, _i32]),
    "kh_yuyv_from_rgb_u8": (_i32, [_vp, _vp, _vp, _i32, _i32]),
    # geometry
    "kh_resize_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_pixel_mapping_coeffs": (_i32, [_i32, _i32, _i32, _P(_f32)]),
    "kh_resize_mapped_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_resize_bilinear_normalize_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _P(_f32), _P(_f32), _i32, _i32, _i64, _i64]),
    "kh_warp_affine_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _P(_f32), _i32, _i32, _i64, _i64]),
    "kh_warp_perspective_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _P(_f32), _i32, _i32, _i64, _i64]),
    "kh_remap_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    # pointer-list batches: (stream, srcs[n], dsts[n], n, ...) — host arrays of device pointers (`pointer_array`)
    "kh_resize_f32_list": (_i32, [_vp, _P(_vp), _P(_vp), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "kh_resize_bilinear_normalize_f32_list": (_i32, [_vp, _P(_vp), _P(_vp), _i32, _i32, _i32, _i32, _i32, _P(_f32), _P(_f32), _i32]),
    "kh_warp_affine_f32_list": (_i32, [_vp, _P(_vp), _P(_vp), _i32, _i32, _i32, _i32, _i32, _i32, _P(_f32), _i32]),
    "kh_warp_perspective_f32_list": (_i32, [_vp, _P(_vp), _P(_vp), _i32, _i32, _i32, _i32, _i32, _i32, _P(_f32), _i32]),
    "kh_remap_f32_list": (_i32, [_vp, _P(_vp), _vp, _vp, _P(_vp), _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "kh_correction_map_polynomial_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _P(C.c_double), _P(C.c_double)]),
    "kh_invert_affine_transform": (None, [_P(_f32), _P(_f32)]),
    "kh_get_rotation_matrix2d": (None, [_f32, _f32, _f32, _f32, _P(_f32)]),
    "kh_invert_homography": (_i32, [_P(_f32), _P(_f32)]),
    # filters
    "kh_separable_filter_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _P(_f32), _i32, _P(_f32), _i32, _i32, _i64, _i64]),
    "kh_gaussian_blur_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _i64, _i64]),
    "kh_box_blur_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_gradient_magnitude_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_separable_filter_f32_list": (_i32, [_vp, _P(_vp), _P(_vp), _i32, _i32, _i32, _i32, _P(_f32), _i32, _P(_f32), _i32]),
    "kh_gaussian_blur_f32_list": (_i32, [_vp, _P(_vp), _P(_vp), _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32]),
    "kh_box_blur_f32_list": (_i32, [_vp, _P(_vp), _P(_vp), _i32, _i32, _i32, _i32, _i32, _i32]),
    "kh_gradient_magnitude_f32_list": (_i32, [_vp, _P(_vp), _P(_vp), _i32, _i32, _i32, _i32, _i32, _i32]),
    "kh_spatial_gradient_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_box_blur_fast_kernels_1d": (_i32, [_f32, _i32, _P(_i32)]),
    "kh_fast_horizontal_filter_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_box_blur_fast_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _i64, _i64]),
    "kh_median_blur_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_bilateral_filter_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f64, _f64, _i32, _i64, _i64]),
    "kh_bilateral_tables": (_i32, [_i32, _f64, _f64, _i32, _P(_i32), _P(_i32), _vp, _vp, _vp, _vp, _vp]),
    "kh_box_blur_kernel_1d": (_i32, [_i32, _P(_f32)]),
    "kh_gaussian_kernel_1d": (_i32, [_i32, _f32, _P(_f32)]),
    "kh_gaussian_resolve": (_i32, [_P(_i32), _P(_f32)]),
    # u8 fixed-point twins
    "kh_quantize_kernel_256": (None, [_P(_f32), _i32, _P(C.c_uint8)]),
    "kh_gaussian_blur_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _i64, _i64]),
    "kh_box_blur_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_remap_u8": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_warp_affine_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _P(_f32), _i32, _i64, _i64]),
    "kh_warp_perspective_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _P(_f32), _i32, _i64, _i64]),
    "kh_resize_fast_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_resize_normalize_to_chw_u8_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _P(_f32), _P(_f32), _i32, _i32, _i32, _i64, _i64]),
    "kh_resize_opencv_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "kh_resize_opencv_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    # pyramid + morphology
    **{n: (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64]) for n in (
        "kh_pyrdown_f32", "kh_pyrup_f32", "kh_pyrdown_u8", "kh_pyrup_u8")},
    "kh_morph_kernel": (_i32, [_i32, _i32, _i32, _P(C.c_uint8)]),
    "kh_morphology_u8": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _P(C.c_uint8), _i32, _i32, _i32, _P(C.c_uint8),
                                _i32, _i64, _i64]),
    # fused pipelines
    "kh_fused_pipeline_build": (_i32, [_vp, _i32, _i32, _i32, _i32, _i64, _P(_vp)]),
    "kh_fused_pipeline_launch": (_i32, [_vp, _vp, _P(_vp), _i32, _i64, _vp, _i64]),
    "kh_fused_pipeline_describe": (_i32, [_vp, C.c_char_p, C.c_size_t]),
    "kh_fused_pipeline_destroy": (None, [_vp]),
    # pointwise
    "kh_normalize_mean_std_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _P(_f32), _P(_f32)]),
    "kh_normalize_rgb_u8_f32": (_i32, [_vp, _vp, _vp, _i64, _P(_f32), _P(_f32)]),
    "kh_find_min_max_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "kh_normalize_min_max_f32": (_i32, [_vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp]),
    "kh_crop": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "kh_flip": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32]),
}


def pointer_array(ptrs):
    """A host ``void*[n]`` for the ``*_list`` entry points: the device pointers of separately allocated images / frames.  The
    library reads it during the call only (the bases travel in the kernel arguments), so a temporary is fine."""
    return (C.c_void_p * len(ptrs))(*ptrs)


def mapped_hip_runtimes() -> dict:
    """{"libamdhip64": [paths], "libhsa-runtime64": [paths]} of the HIP / HSA runtime images mapped into this process
    (from /proc/self/maps).  More than one of either is the root cause of the round-1 "SDMA" finding: two HSA runtimes
    in one process share the process's single KFD event page / copy-engine queues, and a host<->device copy of runtime A
    was seen incomplete at ``hipStreamSynchronize`` once runtime B had initialised (profiles/r01zy_sdma.log,
    profiles/r02*_runtimes.log)."""
    import re
    found = {"libamdhip64": set(), "libhsa-runtime64": set()}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                m = re.search(r"(/\S*/(libamdhip64|libhsa-runtime64)[^/\s]*)", line)
                if m:
                    found[m.group(2)].add(os.path.realpath(m.group(1)))
    except OSError:
        pass
    return {k: sorted(v) for k, v in found.items()}


_RUNTIME_CHECKED_AT = None  # loaded_objects_key() at the last check that found a single runtime

_PHDR_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)


def loaded_objects_key():
    """(number of shared objects mapped into the process, xor of their load addresses) from ``dl_iterate_phdr`` — tens of
    microseconds, and it changes with EVERY dlopen / dlclose, whoever performs it (a Python import, ``ctypes.CDLL`` from an
    already-imported module, a lazy native load inside another library).  ``None`` where the libc has no ``dl_iterate_phdr``."""
    try:
        libc = C.CDLL(None)
        acc = [0, 0]

        def cb(info, _size, _data):
            acc[0] += 1
            acc[1] ^= C.cast(info, C.POINTER(C.c_size_t))[0]   # dl_phdr_info.dlpi_addr is the first member
            return 0

        libc.dl_iterate_phdr(_PHDR_CB(cb), None)
        return tuple(acc)
    except (OSError, AttributeError, ValueError):
        return None


class MultipleHipRuntimes(RuntimeError):
    """Two HIP (or HSA) runtime images are mapped into the process; device work through either is unsafe."""


def assert_single_runtime() -> None:
    """Raise if the process holds more than one HIP or HSA runtime image.  Called wherever device memory crosses to
    or from another library (DLPack / ``__cuda_array_interface__`` import and export): that is the only way a second
    runtime can enter a process that loaded this package first with ``KORNIA_HIP_RUNTIME=system``."""
    # /proc/self/maps has thousands of lines once torch is loaded and this sits on the per-frame zero-copy path: re-parse only
    # when the set of MAPPED IMAGES has changed since the last clean check.  (Round 3 keyed this on len(sys.modules); a second
    # runtime can arrive without that changing — ctypes.CDLL from an imported module, a lazy native load — ADVICE r03.)
    global _RUNTIME_CHECKED_AT
    key = loaded_objects_key()
    if key is not None and _RUNTIME_CHECKED_AT == key:
        return
    rt = mapped_hip_runtimes()
    dup = {k: v for k, v in rt.items() if len(v) > 1}
    if not dup:
        _RUNTIME_CHECKED_AT = key
    if dup:
        raise MultipleHipRuntimes(
            f"two HIP/HSA runtimes are mapped into this process: {dup}.  Copies and stream waits issued through one do not "
            "order against the other (observed: host<->device copies incomplete at hipStreamSynchronize).  Import torch "
            "before kornia_rs, or leave KORNIA_HIP_RUNTIME unset so that kornia_rs binds to torch's bundled runtime.")


def elf_dynamic_names(path) -> dict:
    """{"soname": str | None, "needed": [str]} of an ELF64 little-endian shared object, read from its dynamic section
    (no subprocess, nothing is loaded).  Used to check that a runtime image about to be preloaded is the one
    ``libkornia_hip.so`` asks for by DT_NEEDED name."""
    import struct
    out = {"soname": None, "needed": []}
    try:
        data = Path(path).read_bytes()
        if data[:6] != b"\x7fELF\x02\x01":
            return out
        e_phoff, = struct.unpack_from("<Q", data, 0x20)
        e_phentsize, e_phnum = struct.unpack_from("<HH", data, 0x36)
        loads, dyn = [], None
        for i in range(e_phnum):
            p_type, _flags, p_offset, p_vaddr, _paddr, p_filesz = struct.unpack_from("<IIQQQQ", data, e_phoff + i * e_phentsize)
            if p_type == 1:
                loads.append((p_vaddr, p_offset, p_filesz))
            elif p_type == 2:
                dyn = (p_offset, p_filesz)
        if dyn is None:
            return out
        tags = [struct.unpack_from("<qQ", data, dyn[0] + k) for k in range(0, dyn[1], 16)]
        strtab = next((v for t, v in tags if t == 5), None)
        if strtab is None:
            return out
        off = next((strtab - va + fo for va, fo, sz in loads if va <= strtab < va + sz), None)
        if off is None:
            return out

        def name(idx):
            end = data.index(b"\0", off + idx)
            return data[off + idx:end].decode("utf-8", "replace")
        for t, v in tags:
            if t == 0:
                break
            if t == 1:
                out["needed"].append(name(v))
            elif t == 14:
                out["soname"] = name(v)
    except (OSError, ValueError, IndexError, struct.error):
        pass
    return out


def _preload_hip_runtime() -> str:
    """Make sure the process ends up with ONE HIP runtime, whatever the import order.

    ``libkornia_hip.so`` needs ``libamdhip64.so.7`` (RUNPATH /opt/rocm/lib).  torch-ROCm wheels bundle their own copy as
    ``torch/lib/libamdhip64.so`` (same SONAME) and ``libtorch_hip.so`` asks for it by FILE name — so if this package is
    loaded first the system runtime is mapped, and a later ``import torch`` maps a second HIP + HSA runtime beside it
    (verified from /proc/self/maps on CPU and on the GPU box).  Policy (``KORNIA_HIP_RUNTIME``):
      * unset / ``auto``: a runtime already mapped is used as is; otherwise, if a torch-ROCm wheel is installed, its
        bundled runtime is loaded first (by path, RTLD_GLOBAL) so that both libraries bind to that one image; otherwise
        the system runtime.
      * ``system``: never touch torch's copy (C / Rust hosts, torch-free deployments).
      * a path: load that ``libamdhip64`` image.
    Returns a short description of the choice (``hip.runtime_info()``)."""
    choice = os.environ.get("KORNIA_HIP_RUNTIME", "auto")
    already = mapped_hip_runtimes()["libamdhip64"]
    if already:
        return f"already mapped: {already[0]}"
    if choice == "system":
        return "system (KORNIA_HIP_RUNTIME=system)"
    if choice not in ("", "auto"):
        C.CDLL(choice, mode=C.RTLD_GLOBAL)
        return f"explicit: {choice}"
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")  # locates the wheel without importing it
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = Path(spec.origin).resolve().parent / "lib" / "libamdhip64.so"
        if cand.exists():
            # Only a bundle whose SONAME is the name libkornia_hip.so asks for can serve both libraries as ONE image: with a
            # different SONAME (another ROCm major) the dynamic linker would still load the system runtime for DT_NEEDED and
            # the preload itself would create the two-runtime process this function exists to prevent (round-2 ADVICE).
            want = [n for n in elf_dynamic_names(LIB_PATH)["needed"] if n.startswith("libamdhip64")]
            have = elf_dynamic_names(cand)["soname"]
            if want and have != want[0]:
                return f"system (torch bundle {cand} is {have!r}, libkornia_hip.so needs {want[0]!r})"
            try:
                C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
                return f"torch bundle: {cand}"
            except OSError:
                pass
    return "system"


RUNTIME_CHOICE = ""


def _load() -> C.CDLL:
    global RUNTIME_CHOICE
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found — build it with `make -C kornia-rs_amd` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the device path."
        )
    RUNTIME_CHOICE = _preload_hip_runtime()
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    # Whatever was chosen, the process must hold ONE HIP and ONE HSA runtime now.  Checked here, at load, and not only when
    # device memory crosses a DLPack boundary: a torch-free process can end up with two images through the preload alone.
    dup = {k: v for k, v in mapped_hip_runtimes().items() if len(v) > 1}
    if dup:
        # Two HIP runtimes are the failure that was observed (two sets of streams / copy queues).  A second HSA image alone is what
        # profilers and tools bring (rocprofv3 links the system libhsa-runtime64 beside torch's bundled one; measured runs of
        # rounds 1-2 were made that way): loud warning, not an error.
        only_hsa = list(dup) == ["libhsa-runtime64"]
        msg = (f"libkornia_hip.so was loaded into a process that now maps two HIP/HSA runtime images: {dup} (choice: {RUNTIME_CHOICE}).  "
               "Copies and stream waits issued through one do not order against the other.  Set KORNIA_HIP_RUNTIME=system or to the "
               "path of the one runtime every HIP user of this process should share; KORNIA_HIP_RUNTIME_CHECK=warn downgrades this to a warning.")
        if only_hsa or os.environ.get("KORNIA_HIP_RUNTIME_CHECK", "raise") == "warn":
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
        else:
            raise MultipleHipRuntimes(msg)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def hip_versions() -> tuple:
    """((major, minor, patch) the library was built against, (major, minor, patch) of the runtime it is bound to)."""
    b, r = C.c_int32(0), C.c_int32(0)
    lib.kh_hip_versions(C.byref(b), C.byref(r))
    split = lambda v: (v // 10_000_000, v // 100_000 % 100, v % 100_000)
    return split(b.value), split(r.value)


RUNTIME_VERSION_NOTE = ""


def _check_runtime_version() -> None:
    """The torch-bundle preload (one runtime per process) can bind a library built with ROCm 7.2 to the wheel's 7.0 runtime.  Same major
    = same SONAME (libamdhip64.so.7) and ABI: recorded in ``hip.runtime_info()["version"]``, no noise.  Another MAJOR is a warning."""
    global RUNTIME_VERSION_NOTE
    try:
        build, run = hip_versions()
    except Exception:
        return
    RUNTIME_VERSION_NOTE = f"built against HIP {build[0]}.{build[1]}.{build[2]}, runtime reports {run[0]}.{run[1]}.{run[2]}"
    if run != (0, 0, 0) and build != (0, 0, 0) and run[0] != build[0]:
        import warnings
        warnings.warn(f"libkornia_hip.so: {RUNTIME_VERSION_NOTE} ({RUNTIME_CHOICE}) — different major versions; set KORNIA_HIP_RUNTIME=system "
                      "or to a matching libamdhip64", RuntimeWarning, stacklevel=2)


_check_runtime_version()


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib.kh_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(code: int) -> None:
    if code != KH_OK:
        raise KorniaHipError(code, last_error())


def version() -> str:
    return lib.kh_version().decode()
