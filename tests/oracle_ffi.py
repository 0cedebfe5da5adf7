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
