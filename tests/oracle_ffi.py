"""ctypes binding of the CPU oracle (oracle/libkornia_oracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB_PATH = ROOT / "oracle" / "libkornia_oracle.so"
if not LIB_PATH.exists():
    import subprocess

    subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)
ko = C.CDLL(str(LIB_PATH))

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C")


class KoPreprocessParams(C.Structure):
    _fields_ = [
        ("scale_x", C.c_float), ("scale_y", C.c_float), ("pad_x", C.c_float), ("pad_y", C.c_float),
        ("src_w", C.c_int32), ("src_h", C.c_int32), ("src_pitch", C.c_int32), ("src_bpp", C.c_int32),
        ("fmt", C.c_int32), ("dst_w", C.c_int32), ("dst_h", C.c_int32),
        ("mean", C.c_float * 3), ("inv_std", C.c_float * 3), ("pad_value", C.c_float),
        ("sampling", C.c_int32), ("out_dtype", C.c_int32), ("nframes", C.c_int32),
        ("flags", C.c_int32), ("src_frame_stride", C.c_int64), ("dst_frame_stride", C.c_int64),
    ]


ko.ko_max_threads.restype = C.c_int
ko.ko_set_threads.argtypes = [C.c_int]
ko.ko_pattern_u8.argtypes = [_u8p, C.c_size_t]
ko.ko_pattern_f32.argtypes = [_f32p, C.c_size_t]
ko.ko_preprocess_to_chw.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(KoPreprocessParams)]
ko.ko_preprocess_affine.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float * 4)]
ko.ko_f2h.argtypes = [C.c_float]
ko.ko_f2h.restype = C.c_uint16
ko.ko_gray_from_rgb_u8.argtypes = [_u8p, _u8p, C.c_size_t]
ko.ko_gray_from_rgb_f32.argtypes = [_f32p, _f32p, C.c_size_t]
ko.ko_rgb_from_gray_u8.argtypes = [_u8p, _u8p, C.c_size_t]
ko.ko_rgb_from_gray_f32.argtypes = [_f32p, _f32p, C.c_size_t]
ko.ko_rgb_from_planar420.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
ko.ko_rgb_from_packed422.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
ko.ko_nv12_from_rgb.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
ko.ko_yuyv_from_rgb.argtypes = [_u8p, _u8p, C.c_int, C.c_int]


def pattern_u8(n: int) -> np.ndarray:
    out = np.empty(n, np.uint8)
    ko.ko_pattern_u8(out, n)
    return out


def pattern_f32(n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    ko.ko_pattern_f32(out, n)
    return out


def affine(mode: str, sw, sh, dw, dh):
    out = (C.c_float * 4)()
    ko.ko_preprocess_affine(0 if mode == "letterbox" else 1, sw, sh, dw, dh, C.byref(out))
    return tuple(np.float32(v) for v in out)


FMT = {"rgb": (0, 3), "bgr": (1, 3), "rgba": (0, 4), "bgra": (1, 4), "gray": (2, 1), "nv12": (3, 1),
       "yuyv": (4, 2)}
SAMPLING = {"nearest": 0, "bilinear": 1, "lanczos": 2}


def preprocess(src: np.ndarray, sw: int, sh: int, dw: int, dh: int, *, fmt="rgb", mode="letterbox",
               sampling="bilinear", mean=None, std=None, pad_value=114.0, f16=False, nframes=1,
               src_frame_stride=0, pitch=None) -> np.ndarray:
    """Oracle for the fused kernel.  Returns [nframes, 3, dh, dw] (f32 or uint16 f16 bits)."""
    code, bpp = FMT[fmt]
    p = KoPreprocessParams()
    p.scale_x, p.scale_y, p.pad_x, p.pad_y = affine(mode, sw, sh, dw, dh)
    p.src_w, p.src_h = sw, sh
    p.src_pitch = pitch if pitch is not None else sw * bpp
    p.src_bpp, p.fmt, p.dst_w, p.dst_h = bpp, code, dw, dh
    m = np.zeros(3, np.float32) if mean is None else np.asarray(mean, np.float32)
    s = np.ones(3, np.float32) if std is None else np.asarray(std, np.float32)
    inv = (np.float32(1.0) / s).astype(np.float32)
    for c in range(3):
        p.mean[c] = m[c]
        p.inv_std[c] = inv[c]
    p.pad_value = pad_value
    p.sampling = SAMPLING[sampling]
    p.out_dtype = 1 if f16 else 0
    p.nframes = nframes
    p.src_frame_stride = src_frame_stride
    p.dst_frame_stride = 3 * dw * dh
    src = np.ascontiguousarray(src, dtype=np.uint8)
    out = np.empty((nframes, 3, dh, dw), np.uint16 if f16 else np.float32)
    ko.ko_preprocess_to_chw(src.ctypes.data, out.ctypes.data, C.byref(p))
    return out


def rgb_from_nv12(buf: np.ndarray, w: int, h: int, layout: int = 0) -> np.ndarray:
    out = np.empty((h, w, 3), np.uint8)
    ko.ko_rgb_from_planar420(np.ascontiguousarray(buf, np.uint8).reshape(-1), out.reshape(-1), w, h, layout)
    return out


def rgb_from_yuyv(buf: np.ndarray, w: int, h: int, layout: int = 0) -> np.ndarray:
    out = np.empty((h, w, 3), np.uint8)
    ko.ko_rgb_from_packed422(np.ascontiguousarray(buf, np.uint8).reshape(-1), out.reshape(-1), w, h, layout)
    return out


def nv12_from_rgb(rgb: np.ndarray) -> np.ndarray:
    h, w, _ = rgb.shape
    out = np.empty(w * h * 3 // 2, np.uint8)
    ko.ko_nv12_from_rgb(np.ascontiguousarray(rgb).reshape(-1), out, w, h)
    return out


def yuyv_from_rgb(rgb: np.ndarray) -> np.ndarray:
    h, w, _ = rgb.shape
    out = np.empty(w * h * 2, np.uint8)
    ko.ko_yuyv_from_rgb(np.ascontiguousarray(rgb).reshape(-1), out, w, h)
    return out


def gray_from_rgb_u8(rgb: np.ndarray) -> np.ndarray:
    n = rgb.size // 3
    out = np.empty(n, np.uint8)
    ko.ko_gray_from_rgb_u8(np.ascontiguousarray(rgb).reshape(-1), out, n)
    return out.reshape(rgb.shape[:-1] + (1,))


def gray_from_rgb_f32(rgb: np.ndarray) -> np.ndarray:
    n = rgb.size // 3
    out = np.empty(n, np.float32)
    ko.ko_gray_from_rgb_f32(np.ascontiguousarray(rgb, np.float32).reshape(-1), out, n)
    return out.reshape(rgb.shape[:-1] + (1,))


# ---- colour family helpers -----------------------------------------------------------------------
for _n in ("ko_ycc_from_rgb_u8", "ko_rgb_from_ycc_u8"):
    getattr(ko, _n).argtypes = [_u8p, _u8p, C.c_size_t, C.c_int]
for _n in ("ko_ycc_from_rgb_f32", "ko_rgb_from_ycc_f32"):
    getattr(ko, _n).argtypes = [_f32p, _f32p, C.c_size_t, C.c_int]
for _n in ("ko_hsv_from_rgb_f32", "ko_rgb_from_hsv_f32", "ko_hls_from_rgb_f32", "ko_rgb_from_hls_f32",
           "ko_bgr_from_rgb_f32", "ko_sepia_from_rgb_f32"):
    getattr(ko, _n).argtypes = [_f32p, _f32p, C.c_size_t]
for _n in ("ko_bgr_from_rgb_u8", "ko_sepia_from_rgb_u8"):
    getattr(ko, _n).argtypes = [_u8p, _u8p, C.c_size_t]
ko.ko_rgba_from_rgb_u8.argtypes = [_u8p, _u8p, C.c_size_t, C.c_int]
ko.ko_rgba_from_rgb_f32.argtypes = [_f32p, _f32p, C.c_size_t, C.c_int]
ko.ko_rgb_from_rgba_u8.argtypes = [_u8p, _u8p, C.c_size_t, C.c_int, C.c_void_p]
ko.ko_apply_colormap_u8.argtypes = [_u8p, _u8p, C.c_size_t, _u8p]


def color_map(name: str, src: np.ndarray, cout: int, *extra) -> np.ndarray:
    """Run oracle function ko_<name> on a flat interleaved pixel array; returns [npixels*cout]."""
    src = np.ascontiguousarray(src).reshape(-1)
    fn = getattr(ko, "ko_" + name)
    cin = {"gray_from_rgb": 3, "rgb_from_gray": 1, "apply_colormap": 1, "rgb_from_rgba": 4}.get(
        name.rsplit("_", 1)[0], 3)
    n = src.size // cin
    out = np.empty(n * cout, src.dtype)
    fn(src, out, n, *extra)
    return out


# ---- geometry + filters ---------------------------------------------------------------------------
MODE = {"nearest": 0, "bilinear": 1, "bicubic": 2, "lanczos": 3}
_f6, _f9 = C.c_float * 6, C.c_float * 9
ko.ko_resize_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int]
ko.ko_invert_affine_transform.argtypes = [C.POINTER(_f6), C.POINTER(_f6)]
ko.ko_warp_affine_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.POINTER(_f6), C.c_int]
ko.ko_invert_homography.argtypes = [C.POINTER(_f9), C.POINTER(_f9)]
ko.ko_warp_perspective_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.POINTER(_f9), C.c_int]
ko.ko_remap_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int]
ko.ko_correction_map_polynomial.argtypes = [C.POINTER(C.c_double * 4), C.POINTER(C.c_double * 8), C.c_int, C.c_int, _f32p, _f32p]
ko.ko_box_blur_kernel_1d.argtypes = [C.c_int, _f32p]
ko.ko_gaussian_kernel_1d.argtypes = [C.c_int, C.c_float, _f32p]
ko.ko_gradient_kernels_1d.argtypes = [C.c_int, C.c_int, _f32p, _f32p]
ko.ko_gaussian_resolve.argtypes = [C.POINTER(C.c_int * 2), C.POINTER(C.c_float * 2)]
ko.ko_separable_filter_f32.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int]
ko.ko_gradient_magnitude_f32.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int]


def _img(a):
    a = np.ascontiguousarray(a, np.float32)
    if a.ndim == 2:
        a = a[:, :, None]
    return a


def resize(src, dw, dh, mode="bilinear"):
    src = _img(src)
    sh, sw, c = src.shape
    out = np.empty((dh, dw, c), np.float32)
    ko.ko_resize_f32(src.reshape(-1), sw, sh, out.reshape(-1), dw, dh, c, MODE[mode])
    return out


def invert_affine(m):
    a, o = _f6(*[float(v) for v in m]), _f6()
    ko.ko_invert_affine_transform(C.byref(a), C.byref(o))
    return np.array(list(o), np.float32)


def warp_affine(src, m, dw, dh, mode="bilinear"):
    src = _img(src)
    sh, sw, c = src.shape
    out = np.empty((dh, dw, c), np.float32)
    mm = _f6(*[float(v) for v in m])
    ko.ko_warp_affine_f32(src.reshape(-1), sw, sh, out.reshape(-1), dw, dh, c, C.byref(mm), MODE[mode])
    return out


def invert_homography(m):
    a, o = _f9(*[float(v) for v in m]), _f9()
    ok = ko.ko_invert_homography(C.byref(a), C.byref(o))
    return np.array(list(o), np.float32) if ok else None


def warp_perspective(src, m, dw, dh, mode="bilinear"):
    src = _img(src)
    sh, sw, c = src.shape
    out = np.empty((dh, dw, c), np.float32)
    mm = _f9(*[float(v) for v in m])
    ok = ko.ko_warp_perspective_f32(src.reshape(-1), sw, sh, out.reshape(-1), dw, dh, c, C.byref(mm), MODE[mode])
    return out if ok else None


def remap(src, map_x, map_y, mode="bilinear"):
    src = _img(src)
    sh, sw, c = src.shape
    map_x = np.ascontiguousarray(map_x, np.float32)
    map_y = np.ascontiguousarray(map_y, np.float32)
    dh, dw = map_x.shape[:2]
    out = np.empty((dh, dw, c), np.float32)
    ko.ko_remap_f32(src.reshape(-1), sw, sh, map_x.reshape(-1), map_y.reshape(-1), out.reshape(-1), dw, dh, c, MODE[mode])
    return out


def correction_map(intr, dist, w, h):
    mx, my = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
    a, d = (C.c_double * 4)(*intr), (C.c_double * 8)(*dist)
    ko.ko_correction_map_polynomial(C.byref(a), C.byref(d), w, h, mx.reshape(-1), my.reshape(-1))
    return mx, my


def gaussian_kernel_1d(n, sigma):
    out = np.empty(n, np.float32)
    ko.ko_gaussian_kernel_1d(n, sigma, out)
    return out


def box_kernel_1d(n):
    out = np.empty(n, np.float32)
    ko.ko_box_blur_kernel_1d(n, out)
    return out


def gradient_kernels(kind, n):
    kx, ky = np.empty(n, np.float32), np.empty(n, np.float32)
    return (kx, ky) if ko.ko_gradient_kernels_1d(kind, n, kx, ky) else None


def gaussian_resolve(ksize, sigma):
    k, s = (C.c_int * 2)(*ksize), (C.c_float * 2)(*sigma)
    if not ko.ko_gaussian_resolve(C.byref(k), C.byref(s)):
        return None
    return (k[0], k[1]), (np.float32(s[0]), np.float32(s[1]))


def separable_filter(src, kx, ky):
    src = _img(src)
    h, w, c = src.shape
    out = np.empty_like(src)
    kx, ky = np.ascontiguousarray(kx, np.float32), np.ascontiguousarray(ky, np.float32)
    ko.ko_separable_filter_f32(src.reshape(-1), out.reshape(-1), w, h, c, kx, len(kx), ky, len(ky))
    return out


def gaussian_blur(src, ksize, sigma):
    (kx, ky), (sx, sy) = gaussian_resolve(ksize, sigma)
    return separable_filter(src, gaussian_kernel_1d(kx, sx), gaussian_kernel_1d(ky, sy))


def gradient_magnitude(src, kind, n):
    src = _img(src)
    h, w, c = src.shape
    kx, ky = gradient_kernels(kind, n)
    out = np.empty_like(src)
    ko.ko_gradient_magnitude_f32(src.reshape(-1), out.reshape(-1), w, h, c, kx, ky, n)
    return out


# ---- Lanczos-3 helpers ------------------------------------------------------------------------------
ko.ko_sin_pi.argtypes = [C.c_float]
ko.ko_sin_pi.restype = C.c_float
ko.ko_lanczos3.argtypes = [C.c_float]
ko.ko_lanczos3.restype = C.c_float
ko.ko_lanczos3_weights.argtypes = [C.c_float, C.POINTER(C.c_float * 6)]
ko.ko_lanczos_axis.argtypes = [C.c_int, C.c_int, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"), _f32p]


def lanczos3_weights(frac):
    w = (C.c_float * 6)()
    ko.ko_lanczos3_weights(float(frac), C.byref(w))
    return np.array(list(w), np.float32)


def lanczos_axis(src_len, dst_len):
    x0 = np.empty(dst_len, np.int32)
    w = np.empty(dst_len * 6, np.float32)
    ko.ko_lanczos_axis(src_len, dst_len, x0, w)
    return x0, w.reshape(dst_len, 6)


# ---- u8 fixed-point twins (ko_u8.c) --------------------------------------------------------------------
ko.ko_quantize_kernel_256.argtypes = [_f32p, C.c_int, _u8p]
ko.ko_separable_blur_u8.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int]
ko.ko_binomial3_u8.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
ko.ko_gaussian_blur_u8.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
ko.ko_box_blur_u8.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
ko.ko_remap_u8.argtypes = [_u8p, C.c_int, C.c_int, _f32p, _f32p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int]
ko.ko_warp_affine_u8.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.POINTER(_f6)]
ko.ko_warp_perspective_u8.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.POINTER(_f9)]


def _img8(a):
    a = np.ascontiguousarray(a, np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return a


def quantize_kernel_256(k):
    k = np.ascontiguousarray(k, np.float32)
    out = np.empty(k.size, np.uint8)
    ko.ko_quantize_kernel_256(k, k.size, out)
    return out


def separable_blur_u8(src, qx, qy):
    src = _img8(src)
    h, w, c = src.shape
    out = np.empty_like(src)
    qx, qy = np.ascontiguousarray(qx, np.uint8), np.ascontiguousarray(qy, np.uint8)
    ko.ko_separable_blur_u8(src.reshape(-1), out.reshape(-1), w, h, c, qx, qx.size, qy, qy.size)
    return out


def binomial3_u8(src):
    src = _img8(src)
    h, w, c = src.shape
    out = np.empty_like(src)
    ko.ko_binomial3_u8(src.reshape(-1), out.reshape(-1), w, h, c)
    return out


def gaussian_blur_u8(src, ksize, sigma):
    """-> (image, path) with path 1 = binomial, 2 = general Q8; raises on invalid parameters."""
    src = _img8(src)
    h, w, c = src.shape
    out = np.empty_like(src)
    path = ko.ko_gaussian_blur_u8(src.reshape(-1), out.reshape(-1), w, h, c, ksize[0], ksize[1], sigma[0], sigma[1])
    if path == 0:
        raise ValueError("invalid gaussian parameters")
    return out, path


def box_blur_u8(src, ksize):
    src = _img8(src)
    h, w, c = src.shape
    out = np.empty_like(src)
    if not ko.ko_box_blur_u8(src.reshape(-1), out.reshape(-1), w, h, c, ksize[0], ksize[1]):
        raise ValueError("invalid box kernel")
    return out


def remap_u8(src, map_x, map_y, mode="bilinear"):
    src = _img8(src)
    sh, sw, c = src.shape
    map_x, map_y = np.ascontiguousarray(map_x, np.float32), np.ascontiguousarray(map_y, np.float32)
    dh, dw = map_x.shape
    out = np.empty((dh, dw, c), np.uint8)
    ko.ko_remap_u8(src.reshape(-1), sw, sh, map_x.reshape(-1), map_y.reshape(-1), out.reshape(-1), dw, dh, c, MODE[mode])
    return out


def warp_affine_u8(src, m, dw, dh):
    src = _img8(src)
    sh, sw, c = src.shape
    out = np.empty((dh, dw, c), np.uint8)
    mm = _f6(*[float(v) for v in m])
    ko.ko_warp_affine_u8(src.reshape(-1), sw, sh, out.reshape(-1), dw, dh, c, C.byref(mm))
    return out


def warp_perspective_u8(src, m, dw, dh):
    src = _img8(src)
    sh, sw, c = src.shape
    out = np.empty((dh, dw, c), np.uint8)
    mm = _f9(*[float(v) for v in m])
    if not ko.ko_warp_perspective_u8(src.reshape(-1), sw, sh, out.reshape(-1), dw, dh, c, C.byref(mm)):
        raise ValueError("singular homography")
    return out


# ---- u8 resize cascade + OpenCV-compatible resize (ko_resize_u8.c) ---------------------------------------
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
ko.ko_resize_contribs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
ko.ko_resize_fast_u8.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
ko.ko_resize_opencv_u8.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int]
ko.ko_resize_opencv_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int]
RESIZE_U8_PATHS = {1: "pyrdown2x", 2: "pyrup2x", 3: "nearest", 4: "bilinear", 5: "separable"}


def resize_contribs(src_size, dst_size, filt, antialias):
    """filt: 'cubic' | 'lanczos3' -> (offsets[dst], weights[dst, ksize])"""
    f = {"cubic": 0, "lanczos3": 1}[filt]
    k = ko.ko_resize_contribs(src_size, dst_size, f, int(antialias), None, None, 0)
    ofs, w = np.empty(dst_size, np.int32), np.empty(dst_size * k, np.int32)
    ko.ko_resize_contribs(src_size, dst_size, f, int(antialias), ofs.ctypes.data, w.ctypes.data, k)
    return ofs, w.reshape(dst_size, k)


def resize_fast_u8(src, dw, dh, mode="bilinear", antialias=True):
    """-> (image, path name); raises ValueError for the reference's typed errors."""
    src = np.ascontiguousarray(src, np.uint8)
    if src.ndim == 2:
        src = src[:, :, None]
    sh, sw, c = src.shape
    out = np.empty((dh, dw, c), np.uint8)
    path = ko.ko_resize_fast_u8(src.reshape(-1), sw, sh, out.reshape(-1), dw, dh, c, MODE[mode], int(antialias))
    if path == 0:
        raise ValueError("unsupported channel count or source smaller than 2x2")
    return out, RESIZE_U8_PATHS[path]


def resize_opencv(src, dw, dh, mode="bilinear"):
    src = np.ascontiguousarray(src)
    if src.ndim == 2:
        src = src[:, :, None]
    sh, sw, c = src.shape
    out = np.empty((dh, dw, c), src.dtype)
    fn = ko.ko_resize_opencv_u8 if src.dtype == np.uint8 else ko.ko_resize_opencv_f32
    if not fn(src.reshape(-1), sw, sh, out.reshape(-1), dw, dh, c, MODE[mode]):
        raise ValueError("unsupported interpolation")
    return out


# ---- fused RGB8 -> normalised CHW (P/resize/fused.rs) -------------------------------------------------------
_f3 = C.c_float * 3
ko.ko_normalize_params.argtypes = [C.POINTER(_f3), C.POINTER(_f3), C.POINTER(_f3), C.POINTER(_f3)]
ko.ko_resize_normalize_to_chw.argtypes = [_u8p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.POINTER(_f3), C.POINTER(_f3),
                                          C.c_int, C.c_int]
FUSED_PATHS = {1: "box2x", 2: "bilinear", 3: "nearest", 4: "separable"}


def normalize_params(mean, std):
    m, s, sc, bi = _f3(*mean), _f3(*std), _f3(), _f3()
    ko.ko_normalize_params(C.byref(m), C.byref(s), C.byref(sc), C.byref(bi))
    return np.array(list(sc), np.float32), np.array(list(bi), np.float32)


def resize_normalize_to_chw(src, dw, dh, scale, bias, mode="bilinear", antialias=True):
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw, c = src.shape
    assert c == 3
    out = np.empty((3, dh, dw), np.float32)
    sc, bi = _f3(*[float(v) for v in scale]), _f3(*[float(v) for v in bias])
    path = ko.ko_resize_normalize_to_chw(src.reshape(-1), sw, sh, out.reshape(-1), dw, dh, C.byref(sc), C.byref(bi),
                                         MODE[mode], int(antialias))
    return out, FUSED_PATHS[path]


# ---- pyramid + morphology (ko_pyramid_morph.c) ------------------------------------------------------------------
for _n, _p in (("ko_pyrdown_f32", _f32p), ("ko_pyrup_f32", _f32p), ("ko_pyrdown_u8", _u8p), ("ko_pyrup_u8", _u8p)):
    getattr(ko, _n).argtypes = [_p, C.c_int, C.c_int, _p, C.c_int]
ko.ko_morph_kernel.argtypes = [C.c_int, C.c_int, C.c_int, _u8p]
ko.ko_morphology_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, _u8p]
BORDER = {"constant": 0, "replicate": 1, "reflect101": 2, "reflect": 3, "wrap": 4}


def _pyr(src, up):
    src = np.ascontiguousarray(src)
    if src.ndim == 2:
        src = src[:, :, None]
    sh, sw, c = src.shape
    dh, dw = (2 * sh, 2 * sw) if up else ((sh + 1) // 2, (sw + 1) // 2)
    out = np.empty((dh, dw, c), src.dtype)
    fn = getattr(ko, f"ko_{'pyrup' if up else 'pyrdown'}_{'u8' if src.dtype == np.uint8 else 'f32'}")
    fn(src.reshape(-1), sw, sh, out.reshape(-1), c)
    return out


def pyrdown(src):
    return _pyr(src, False)


def pyrup(src):
    return _pyr(src, True)


def morph_kernel(shape, width, height=None):
    height = width if height is None else height
    out = np.empty(width * height, np.uint8)
    ko.ko_morph_kernel({"box": 0, "cross": 1, "ellipse": 2}[shape], width, height, out)
    return out.reshape(height, width)


def morphology_u8(src, op, mask, border="constant", cval=None):
    src = np.ascontiguousarray(src, np.uint8)
    if src.ndim == 2:
        src = src[:, :, None]
    h, w, c = src.shape
    out = np.empty_like(src)
    mask = np.ascontiguousarray(mask, np.uint8)
    cv = np.zeros(4, np.uint8)
    if cval is not None:
        cv[:c] = cval
    ko.ko_morphology_u8(src.reshape(-1), w, h, c, out.reshape(-1), {"dilate": 0, "erode": 1}[op], mask.reshape(-1),
                        mask.shape[1], mask.shape[0], BORDER[border], cv)
    return out


# ---- fused per-pixel pipelines (P/cuda/fusion.rs) -------------------------------------------------------------
ko.ko_fused_pipeline.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _f32p, C.c_int, C.c_int, _f32p]


def fused_pipeline(src, dw, dh, maps, sink, read_dst=None):
    """maps: list of ("normalize", scale, bias) | ("gray",); sink: "chw" | "c1"."""
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw, _ = src.shape
    rdw, rdh = read_dst or (dw, dh)
    kinds = np.array([16 if m[0] == "normalize" else 17 for m in maps] or [0], np.int32)
    params = np.zeros((max(len(maps), 1), 6), np.float32)
    for i, m in enumerate(maps):
        if m[0] == "normalize":
            params[i, :3], params[i, 3:] = m[1], m[2]
    out = np.empty((3 if sink == "chw" else 1, dh, dw), np.float32)
    ko.ko_fused_pipeline(src.reshape(-1), sw, sh, rdw, rdh, dw, dh, kinds, params.reshape(-1), len(maps),
                         32 if sink == "chw" else 33, out.reshape(-1))
    return out


# ---- CIE colour spaces (ko_cie.c) ---------------------------------------------------------------------------
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
ko.ko_cie_f32.argtypes = [_f32p, _f32p, C.c_size_t, C.c_int]
ko.ko_cie_f64.argtypes = [_f64p, _f64p, C.c_size_t, C.c_int]
CIE = {"linear_rgb_from_rgb": 0, "rgb_from_linear_rgb": 1, "xyz_from_rgb": 2, "rgb_from_xyz": 3, "lab_from_rgb": 4,
       "rgb_from_lab": 5, "luv_from_rgb": 6, "rgb_from_luv": 7}


def cie(name, src):
    """f32 input -> the reference's f32 scalar path; f64 input -> its f64 exact-formula functions."""
    src = np.ascontiguousarray(src)
    out = np.empty_like(src)
    (ko.ko_cie_f32 if src.dtype == np.float32 else ko.ko_cie_f64)(src.reshape(-1), out.reshape(-1), src.size // 3, CIE[name])
    return out


# ---- f64 colour family (P/color/{gray,hsv,hls,yuv}/mod.rs f64 arms + the CIE scalar64 formulas) -------------------
F64_CONV = {"linear_rgb_from_rgb": 0, "rgb_from_linear_rgb": 1, "xyz_from_rgb": 2, "rgb_from_xyz": 3, "lab_from_rgb": 4,
            "rgb_from_lab": 5, "luv_from_rgb": 6, "rgb_from_luv": 7, "gray_from_rgb": 8, "rgb_from_gray": 9, "hsv_from_rgb": 10,
            "rgb_from_hsv": 11, "hls_from_rgb": 12, "rgb_from_hls": 13, "ycbcr_from_rgb": 14, "rgb_from_ycbcr": 15,
            "yuv_from_rgb": 16, "rgb_from_yuv": 17}
ko.ko_color_f64.argtypes = [_f64p, _f64p, C.c_size_t, C.c_int]
ko.ko_color_f64.restype = C.c_int


def color_f64(name, src):
    """src: float64 [..., cin] -> float64 [..., cout] through the reference's f64 arms."""
    src = np.ascontiguousarray(src, np.float64)
    conv = F64_CONV[name]
    cin, cout = (1 if conv == 9 else 3), (1 if conv == 8 else 3)
    n = src.size // cin
    out = np.empty(n * cout, np.float64)
    assert ko.ko_color_f64(src.reshape(-1), out, n, conv) == 0
    return out.reshape(src.shape[:-1] + (cout,)) if src.ndim > 1 else out


YUV_MODE = {"bt601_full": 0, "bt709_full": 1, "bt601_limited": 2}
ko.ko_yuyv_to_rgb_mode.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
ko.ko_yuyv_to_rgb_mode.restype = C.c_int


def yuyv_to_rgb_mode(buf, w, h, mode, fill=0):
    """convert_yuyv_to_rgb_u8; pixels the reference leaves untouched (odd widths) keep `fill`."""
    out = np.full((h, w, 3), fill, np.uint8)
    assert ko.ko_yuyv_to_rgb_mode(np.ascontiguousarray(buf, np.uint8).reshape(-1), out.reshape(-1), w, h, YUV_MODE[mode]) == 0
    return out


# ---- resize launchers' PixelMapping + fused resize/normalise (P/cuda/resize.rs:184-236, 433-473, 580-650) -----------
PIXEL_MAPPING = {"half_pixel": 0, "align_corners": 1}
ko.ko_pixel_mapping_coeffs.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
ko.ko_pixel_mapping_coeffs.restype = C.c_int
ko.ko_resize_mapped_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
ko.ko_resize_mapped_f32.restype = C.c_int
ko.ko_resize_bilinear_normalize_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.POINTER(C.c_float),
                                                C.POINTER(C.c_float), C.c_int]
ko.ko_resize_bilinear_normalize_f32.restype = C.c_int


def pixel_mapping_coeffs(mapping, src_len, dst_len):
    out = (C.c_float * 2)()
    assert ko.ko_pixel_mapping_coeffs(PIXEL_MAPPING[mapping], src_len, dst_len, out) == 0
    return float(out[0]), float(out[1])


def resize_mapped(src, dw, dh, mode="bilinear", mapping="half_pixel"):
    src = _img(src)
    sh, sw, c = src.shape
    out = np.empty((dh, dw, c), np.float32)
    assert ko.ko_resize_mapped_f32(src.reshape(-1), sw, sh, out.reshape(-1), dw, dh, c, MODE[mode], PIXEL_MAPPING[mapping]) == 0
    return out


def resize_bilinear_normalize(src, dw, dh, mean, std, mapping="half_pixel"):
    src = _img(src)
    sh, sw, c = src.shape
    assert c == 3
    out = np.empty((dh, dw, 3), np.float32)
    rc = ko.ko_resize_bilinear_normalize_f32(src.reshape(-1), sw, sh, out.reshape(-1), dw, dh, (C.c_float * 3)(*mean),
                                             (C.c_float * 3)(*std), PIXEL_MAPPING[mapping])
    if rc:
        raise ValueError(f"ko_resize_bilinear_normalize_f32 -> {rc}")
    return out


BAYER = {"rggb": 0, "bggr": 1, "grbg": 2, "gbrg": 3}
ko.ko_rgb_from_bayer.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
ko.ko_rgb_from_bayer.restype = C.c_int


def rgb_from_bayer(mosaic, pattern):
    m = np.ascontiguousarray(mosaic, np.uint8)
    if m.ndim == 3:
        m = m[:, :, 0]
    h, w = m.shape
    out = np.empty((h, w, 3), np.uint8)
    assert ko.ko_rgb_from_bayer(np.ascontiguousarray(m).reshape(-1), out.reshape(-1), w, h, BAYER[pattern]) == 0
    return out


# ---- the rest of the filter module (oracle/ko_filter_extra.c) ------------------------------------------------------------
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
ko.ko_spatial_gradient_f32.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int]
ko.ko_box_blur_fast_kernels_1d.argtypes = [C.c_float, C.c_int, _i32p]
ko.ko_fast_horizontal_filter.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int]
ko.ko_box_blur_fast_f32.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
ko.ko_median_blur_u8.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int]
ko.ko_v_exp_f32.argtypes = [C.c_float]
ko.ko_v_exp_f32.restype = C.c_float
ko.ko_bilateral_tables.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
ko.ko_bilateral_filter_u8.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]

GRADIENT_KINDS = {"sobel": 0, "scharr": 1}


def spatial_gradient(src, kind="sobel"):
    src = _img(src)
    h, w, c = src.shape
    gx, gy = np.empty_like(src), np.empty_like(src)
    ko.ko_spatial_gradient_f32(src.reshape(-1), gx.reshape(-1), gy.reshape(-1), w, h, c, GRADIENT_KINDS[kind])
    return gx, gy


def box_blur_fast_kernels_1d(sigma, kernels):
    out = np.empty(kernels, np.int32)
    ko.ko_box_blur_fast_kernels_1d(sigma, kernels, out)
    return [int(v) for v in out]


def fast_horizontal_filter(src, half):
    """Returns the TRANSPOSED image (W, H, C), or None where the reference would index out of bounds."""
    src = _img(src)
    h, w, c = src.shape
    out = np.empty((w, h, c), np.float32)
    return out if ko.ko_fast_horizontal_filter(src.reshape(-1), out.reshape(-1), w, h, c, half) == 0 else None


def box_blur_fast(src, sigma):
    src = _img(src)
    h, w, c = src.shape
    out = np.empty_like(src)
    return out if ko.ko_box_blur_fast_f32(src.reshape(-1), out.reshape(-1), w, h, c, sigma[0], sigma[1]) == 0 else None


def median_blur(src, ksize):
    src = _img8(src)
    h, w, c = src.shape
    out = np.empty_like(src)
    return out if ko.ko_median_blur_u8(src.reshape(-1), out.reshape(-1), w, h, c, ksize) == 0 else None


def bilateral_tables(d, sigma_color, sigma_space):
    radius = C.c_int(0)
    n = ko.ko_bilateral_tables(d, sigma_color, sigma_space, 0, C.byref(radius), None, None, None, None, None)
    dy, dx, order = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
    sw, cw = np.empty(n, np.float32), np.empty(256, np.float32)
    ko.ko_bilateral_tables(d, sigma_color, sigma_space, n, C.byref(radius), dy.ctypes.data, dx.ctypes.data, sw.ctypes.data, cw.ctypes.data,
                           order.ctypes.data)
    return {"radius": radius.value, "dy": dy, "dx": dx, "space_weight": sw, "color_weight": cw, "simd_order": order}


def bilateral_filter(src, d, sigma_color, sigma_space):
    src = _img8(src)
    h, w, c = src.shape
    assert c == 1
    out = np.empty_like(src)
    assert ko.ko_bilateral_filter_u8(src.reshape(-1), out.reshape(-1), w, h, d, sigma_color, sigma_space) == 0
    return out


# ---- pointwise (ko_pointwise.c) --------------------------------------------------------------------------
ko.ko_normalize_mean_std_f32.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p]


def normalize_mean_std(src, mean, std):
    """normalize_mean_std (P/normalize.rs:56-87) on an HWC f32 image."""
    src = _img(src)
    h, w, c = src.shape
    out = np.empty_like(src)
    ko.ko_normalize_mean_std_f32(src.reshape(-1), out.reshape(-1), w, h, c, np.ascontiguousarray(mean, np.float32),
                                 np.ascontiguousarray(std, np.float32))
    return out
