"""The Rust crate's spellings — ``kornia_imgproc::{color, resize, filter, morphology}::*`` free functions with their
``(src, &mut dst, params...)`` argument order and typed ``_u8`` / ``_f32`` suffixes — over the same device entry points as
``kornia_rs.imgproc`` (which follows kornia-py's ``imgproc.pyi``).  A caller porting Rust call sites finds every name here;
each function validates the element type its suffix promises (``Image<u8, C>`` vs ``Image<f32, C>`` is a compile-time fact in
Rust, a typed ``ImageError`` here) and returns ``None`` like the Rust ``Result<(), ImageError>``.

Host-side kernel builders are the reference's tables: ``sobel_kernel_1d`` / ``scharr_kernel_1d`` (P/filter/kernels.rs:55-100),
``normalized_sobel_kernel3`` / ``normalized_scharr_kernel3`` (:107-140), ``box_blur_kernel_1d`` / ``gaussian_kernel_1d``
(:10-43, built by the library's host builders), ``box_kernel`` / ``cross_kernel`` / ``ellipse_kernel``
(P/morphology/kernels.rs:113-185).
"""
import ctypes as C
from typing import List, Sequence, Tuple

from . import imgproc
from ._ffi import lib
from .image import Image, ImageError
from .imgproc import Kernel, _check, _require


def _typed(fn, dtype: str, channels: Tuple[int, ...], name: str):
    def call(src: Image, dst: Image, *args) -> None:
        _require(src, dtype, channels, name)
        fn(src, *args, dst)
    call.__name__ = call.__qualname__ = name
    call.__doc__ = f"``{name}(src, dst, ...)``: the {dtype} instantiation of ``imgproc.{fn.__name__}``."
    return call


_C3, _C1, _ANY = (3,), (1,), (1, 2, 3, 4)
for _base, _cin in (("gray_from_rgb", _C3), ("rgb_from_gray", _C1), ("bgr_from_rgb", _C3), ("rgba_from_rgb", _C3), ("bgra_from_rgb", _C3),
                    ("sepia_from_rgb", _C3)):
    for _sfx, _dt in (("u8", "uint8"), ("f32", "float32")):
        globals()[f"{_base}_{_sfx}"] = _typed(getattr(imgproc, _base), _dt, _cin, f"{_base}_{_sfx}")
for _base in ("hsv_from_rgb", "rgb_from_hsv", "hls_from_rgb", "rgb_from_hls"):
    globals()[f"{_base}_f32"] = _typed(getattr(imgproc, _base), "float32", _C3, f"{_base}_f32")


def _ycc(direction: str, dtype: str, name: str):
    def call(src: Image, dst: Image, order: str = "ycrcb") -> None:
        """``order``: "ycrcb" (YCbCr family) or "yuv" — the ``YccOrder`` of P/color/yuv/kernels.rs:23-62."""
        _require(src, dtype, _C3, name)
        table = {("ycc_from_rgb", "ycrcb"): imgproc.ycbcr_from_rgb, ("ycc_from_rgb", "yuv"): imgproc.yuv_from_rgb,
                 ("rgb_from_ycc", "ycrcb"): imgproc.rgb_from_ycbcr, ("rgb_from_ycc", "yuv"): imgproc.rgb_from_yuv}
        fn = table.get((direction, str(order).lower()))
        if fn is None:
            raise ImageError("InvalidArgument", f"{name}: unknown order {order!r} (ycrcb, yuv)")
        fn(src, dst)
    call.__name__ = call.__qualname__ = name
    return call


for _dir in ("ycc_from_rgb", "rgb_from_ycc"):
    for _sfx, _dt in (("u8", "uint8"), ("f32", "float32")):
        globals()[f"{_dir}_{_sfx}"] = _ycc(_dir, _dt, f"{_dir}_{_sfx}")


# ---- resize (P/resize/mod.rs:245-440, P/resize/opencv_compat.rs:76-112) -----------------------------------------------
def resize_fast_u8_aa(src: Image, dst: Image, interpolation: str, antialias: bool) -> None:
    _require(src, "uint8", _ANY, "resize_fast_u8_aa")
    imgproc.resize_fast(src, (dst.height, dst.width), interpolation, antialias, dst)


def resize_fast_u8(src: Image, dst: Image, interpolation: str) -> None:
    resize_fast_u8_aa(src, dst, interpolation, True)


def resize_fast_rgb_aa(src: Image, dst: Image, interpolation: str, antialias: bool) -> None:
    _require(src, "uint8", _C3, "resize_fast_rgb_aa")
    resize_fast_u8_aa(src, dst, interpolation, antialias)


def resize_fast_rgb(src: Image, dst: Image, interpolation: str) -> None:
    resize_fast_rgb_aa(src, dst, interpolation, True)


def resize_fast_mono_aa(src: Image, dst: Image, interpolation: str, antialias: bool) -> None:
    _require(src, "uint8", _C1, "resize_fast_mono_aa")
    resize_fast_u8_aa(src, dst, interpolation, antialias)


def resize_fast_mono(src: Image, dst: Image, interpolation: str) -> None:
    resize_fast_mono_aa(src, dst, interpolation, True)


def resize_opencv_u8(src: Image, dst: Image, interpolation: str) -> None:
    _require(src, "uint8", _ANY, "resize_opencv_u8")
    imgproc.resize_opencv(src, (dst.height, dst.width), interpolation, dst)


def resize_opencv_f32(src: Image, dst: Image, interpolation: str) -> None:
    _require(src, "float32", _ANY, "resize_opencv_f32")
    imgproc.resize_opencv(src, (dst.height, dst.width), interpolation, dst)


# ---- decode of raw camera buffers: the generic names behind rgb_from_nv12 & co (P/color/yuv/kernels.rs:766, 986) -------------
def rgb_from_planar420(data, width: int, height: int, dst: Image, layout: str = "nv12") -> None:
    fn = {"nv12": imgproc.rgb_from_nv12, "nv21": imgproc.rgb_from_nv21, "i420": imgproc.rgb_from_i420, "yv12": imgproc.rgb_from_yv12}.get(layout)
    if fn is None:
        raise ImageError("InvalidArgument", f"rgb_from_planar420: unknown layout {layout!r} (nv12, nv21, i420, yv12)")
    fn(data, width, height, dst)


def rgb_from_packed422(data, width: int, height: int, dst: Image, layout: str = "yuyv") -> None:
    fn = {"yuyv": imgproc.rgb_from_yuyv, "uyvy": imgproc.rgb_from_uyvy, "yvyu": imgproc.rgb_from_yvyu}.get(layout)
    if fn is None:
        raise ImageError("InvalidArgument", f"rgb_from_packed422: unknown layout {layout!r} (yuyv, uyvy, yvyu)")
    fn(data, width, height, dst)


def rgb_from_bayer8(src, dst: Image, pattern: str = None) -> None:
    imgproc.rgb_from_bayer(src, pattern, dst)


# ---- filters: the parallel spellings share the arithmetic (P/filter/ops.rs:355-509) -----------------------------------
def spatial_gradient_float(src: Image, dx: Image, dy: Image) -> None:
    imgproc.spatial_gradient_float(src, dx, dy)


spatial_gradient_float_parallel_row = spatial_gradient_float
spatial_gradient_float_parallel = spatial_gradient_float


def scharr_spatial_gradient_float(src: Image, dx: Image, dy: Image) -> None:
    imgproc.scharr_spatial_gradient_float(src, dx, dy)


def box_blur_kernel_1d(kernel_size: int) -> List[float]:
    out = (C.c_float * max(int(kernel_size), 1))()
    _check(lib.kh_box_blur_kernel_1d(kernel_size, out))
    return [float(v) for v in out[:kernel_size]]


def gaussian_kernel_1d(kernel_size: int, sigma: float) -> List[float]:
    out = (C.c_float * max(int(kernel_size), 1))()
    _check(lib.kh_gaussian_kernel_1d(kernel_size, sigma, out))
    return [float(v) for v in out[:kernel_size]]


def sobel_kernel_1d(kernel_size: int) -> Tuple[List[float], List[float]]:
    table = {3: ([-1.0, 0.0, 1.0], [1.0, 2.0, 1.0]), 5: ([-1.0, -2.0, 0.0, 2.0, 1.0], [1.0, 4.0, 6.0, 4.0, 1.0])}
    if kernel_size not in table:
        raise ImageError("InvalidKernelLength", f"invalid kernel length ({kernel_size}, {kernel_size})")
    return table[kernel_size]


def scharr_kernel_1d(kernel_size: int) -> Tuple[List[float], List[float]]:
    if kernel_size != 3:
        raise ImageError("InvalidKernelLength", f"invalid kernel length ({kernel_size}, {kernel_size})")
    return [-1.0, 0.0, 1.0], [3.0, 10.0, 3.0]


def _kernel3(a: float, b: float):
    return ([[-a, 0.0, a], [-b, 0.0, b], [-a, 0.0, a]], [[-a, -b, -a], [0.0, 0.0, 0.0], [a, b, a]])


def normalized_sobel_kernel3():
    return _kernel3(0.125, 0.25)


def normalized_scharr_kernel3():
    return _kernel3(0.09375, 0.3125)


box_blur_fast_kernels_1d = imgproc.box_blur_fast_kernels_1d


# ---- morphology kernels (P/morphology/kernels.rs:113-185) --------------------------------------------------------------
def box_kernel(size: int) -> Kernel:
    return Kernel("box", size)


def cross_kernel(size: int) -> Kernel:
    return Kernel("cross", size)


def ellipse_kernel(width: int, height: int) -> Kernel:
    return Kernel("ellipse", (width, height))
