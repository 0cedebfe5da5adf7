"""``imgproc`` free functions with the reference's residency dispatch, running on the HIP backend.

Mirrors the public ops of ``kornia-imgproc`` (``color::*``, ``resize::resize``, ``warp::*``,
``interpolation::remap``, ``filter::*``, ``normalize::*``, ``crop``, ``flip``) and their Python
spellings (kornia-py/python/kornia_rs/imgproc.pyi:23-158).  Every op starts with the residency
classification of crates/kornia-imgproc/src/cuda/dispatch.rs:105-130:

* device/device pair  -> the HIP kernel, launched on the SOURCE image's stream; if the destination
  carries a different stream it is fenced in first (``DeviceExec::for_streams``, :50-67);
* mixed host/device   -> ``ImageError('MixedResidency')`` — never an implicit transfer (:128);
* different devices   -> ``ImageError('DeviceMismatch')`` (:51-53);
* host/host           -> ``ImageError('HostPathUnavailable')``: this build ships the device backend
  only; the CPU implementation stays in the reference crate.  It is NOT silently computed
  elsewhere.
* unsupported dtype / channel count -> ``ImageError('NoDeviceKernel')``, never a fallback
  (``no_gpu_kernel_err``, dispatch.rs:203-211).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from ._ffi import lib
from .hip import DeviceBuffer, Stream
from .image import Image, ImageError
from . import colormap as _colormap
from .tensor import Tensor

_INTERP = {"nearest": 0, "bilinear": 1, "bicubic": 2, "lanczos": 3}


def _check(rc: int) -> None:
    if rc == _ffi.KH_OK:
        return
    msg = _ffi.last_error()
    kind = {_ffi.KH_ERR_SINGULAR: "CannotComputeDeterminant", _ffi.KH_ERR_UNSUPPORTED: "NoDeviceKernel",
            _ffi.KH_ERR_HIP: "Hip"}.get(rc, "InvalidArgument")
    raise ImageError(kind, msg)


class _DeviceExec(Stream):
    """``DeviceExec`` (P/cuda/dispatch.rs:28-82): the stream a device op launches on — the source image's — plus the other
    streams its operands carry.  ``for_streams`` fences those INTO the launch stream before the launch; ``check`` is the
    reference's ``run``: it takes the status of the launch that has just been enqueued and then fences the launch stream
    BACK into every other stream, so a destination's own stream (its ``numpy()``, the next op that reads it, its
    stream-ordered free) is ordered after the kernel that writes it.  Same-stream operands cost nothing."""

    def __init__(self, launch: Stream, others: Sequence[Stream] = ()):
        super().__init__(launch.cuda_stream_ptr, launch.device, owned=False)
        self._launch = launch  # keeps an owned stream alive for as long as the exec
        self._others = []
        for o in others:
            self.join(o)

    def join(self, other: Optional[Stream]) -> None:
        """Order ``other``'s queued work before the launch, and remember it for the fence back."""
        if other is None or other.cuda_stream_ptr == self.cuda_stream_ptr:
            return
        if all(o.cuda_stream_ptr != other.cuda_stream_ptr for o in self._others):
            _check(lib.kh_stream_fence(other.cuda_stream_ptr, self.cuda_stream_ptr))
            self._others.append(other)

    def check(self, rc: int) -> None:
        _check(rc)
        for o in self._others:
            _check(lib.kh_stream_fence(self.cuda_stream_ptr, o.cuda_stream_ptr))


def _pair_residency(src: Image, *dsts: Image) -> _DeviceExec:
    """Host/Device/Mixed classification + same-device check + cross-stream fences (pair_residency, P/cuda/dispatch.rs:105-133).
    Returns the ``_DeviceExec`` to launch on: ``exec.check(lib.kh_...(exec.cuda_stream_ptr, ...))``."""
    for dst in dsts:
        if src.is_device != dst.is_device:
            raise ImageError("MixedResidency", "source and destination images must both be on the host or both on the "
                                               "device; there is no implicit transfer")
    if not src.is_device:
        raise ImageError("HostPathUnavailable", "host images: this build provides the HIP device backend only — move "
                                                "the image with .to_hip(stream) (the CPU path lives in the reference crate)")
    for dst in dsts:
        if src.device_id != dst.device_id:
            raise ImageError("DeviceMismatch", f"images live on different devices ({src.device} vs {dst.device})")
    if src.stream is None or any(dst.stream is None for dst in dsts):
        raise ImageError("UnsupportedDevice", "device image without a stream (untyped foreign memory); re-wrap it "
                                              "with Image.from_dlpack(obj, stream=...)")
    return _DeviceExec(src.stream, [dst.stream for dst in dsts])


def _require(img: Image, dtype: str, channels: Tuple[int, ...], what: str) -> None:
    if img.dtype != dtype or img.channels not in channels:
        raise ImageError("NoDeviceKernel", f"{what}: no device kernel for {img.dtype} x {img.channels} channels "
                                           f"(supported: {dtype} x {channels})")


def _same_size(a: Image, b: Image) -> None:
    if a.size != b.size:
        raise ImageError("InvalidImageSize", f"image sizes differ: {a.width}x{a.height} vs {b.width}x{b.height}")


def _new_like(src: Image, channels: Optional[int] = None, dtype: Optional[str] = None,
              size: Optional[Tuple[int, int]] = None) -> Image:
    if not src.is_device:
        raise ImageError("HostPathUnavailable", "host images: this build provides the HIP device backend only — move "
                                                "the image with .to_hip(stream)")
    w, h = size if size is not None else (src.width, src.height)
    return Image.uninit(w, h, channels or src.channels, dtype or src.dtype, src.stream)


# ---- colour maps --------------------------------------------------------------------------------

def _map_f64(code_name: str, src: Image, dst: Optional[Image], cin: int, cout: int) -> Image:
    """float64 images: the f64 launchers of P/color/cuda_dispatch.rs:48-61,111-135 == kh_color_convert_f64."""
    _require(src, "float64", (cin,), code_name)
    out = dst if dst is not None else _new_like(src, channels=cout)
    _require(out, "float64", (cout,), code_name)
    _same_size(src, out)
    stream = _pair_residency(src, out)
    stream.check(lib.kh_color_convert_f64(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width * src.height,
                                          _ffi.KH_F64[code_name]))
    return out


def _map(name: str, src: Image, dst: Optional[Image], cin: int, cout: int, dtypes: Sequence[str], *extra,
         f64: Optional[str] = None) -> Image:
    if f64 is not None and src.dtype == "float64":
        return _map_f64(f64, src, dst, cin, cout)
    if src.dtype not in dtypes:
        raise ImageError("NoDeviceKernel", f"{name}: no device kernel for dtype {src.dtype} (supported: {tuple(dtypes)})")
    _require(src, src.dtype, (cin,), name)
    out = dst if dst is not None else _new_like(src, channels=cout)
    _require(out, src.dtype, (cout,), name)
    _same_size(src, out)
    stream = _pair_residency(src, out)
    suffix = {"uint8": "u8", "float32": "f32"}[src.dtype]
    fn = getattr(lib, f"kh_{name}_{suffix}")
    stream.check(fn(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width * src.height, *extra))
    return out


def gray_from_rgb(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("gray_from_rgb", src, dst, 3, 1, ("uint8", "float32"), f64="gray_from_rgb")


def gray_from_rgb_f32(src: Image, dst: Optional[Image] = None) -> Image:
    """float32-only spelling kept by the Python stubs (imgproc.pyi:25)."""
    _require(src, "float32", (3,), "gray_from_rgb_f32")
    return gray_from_rgb(src, dst)


def rgb_from_gray(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("rgb_from_gray", src, dst, 1, 3, ("uint8", "float32"), f64="rgb_from_gray")


def bgr_from_rgb(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("bgr_from_rgb", src, dst, 3, 3, ("uint8", "float32"))


def rgba_from_rgb(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("rgba_from_rgb", src, dst, 3, 4, ("uint8", "float32"), 0)


def bgra_from_rgb(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("rgba_from_rgb", src, dst, 3, 4, ("uint8", "float32"), 1)


def _rgb_from_4(src: Image, dst: Optional[Image], swap: int, background) -> Image:
    bg = None
    if background is not None:
        arr = (C.c_uint8 * 3)(*[int(v) for v in background])
        bg = C.cast(arr, C.c_void_p)
    return _map("rgb_from_rgba", src, dst, 4, 3, ("uint8",), swap, bg)


def rgb_from_rgba(src: Image, dst: Optional[Image] = None, background: Optional[Sequence[int]] = None) -> Image:
    return _rgb_from_4(src, dst, 0, background)


def rgb_from_bgra(src: Image, dst: Optional[Image] = None, background: Optional[Sequence[int]] = None) -> Image:
    return _rgb_from_4(src, dst, 1, background)


def ycbcr_from_rgb(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("ycc_from_rgb", src, dst, 3, 3, ("uint8", "float32"), _ffi.KH_YCC_YCRCB, f64="ycbcr_from_rgb")


def rgb_from_ycbcr(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("rgb_from_ycc", src, dst, 3, 3, ("uint8", "float32"), _ffi.KH_YCC_YCRCB, f64="rgb_from_ycbcr")


def yuv_from_rgb(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("ycc_from_rgb", src, dst, 3, 3, ("uint8", "float32"), _ffi.KH_YCC_YUV, f64="yuv_from_rgb")


def rgb_from_yuv(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("rgb_from_ycc", src, dst, 3, 3, ("uint8", "float32"), _ffi.KH_YCC_YUV, f64="rgb_from_yuv")


def hsv_from_rgb(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("hsv_from_rgb", src, dst, 3, 3, ("float32",), f64="hsv_from_rgb")


def rgb_from_hsv(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("rgb_from_hsv", src, dst, 3, 3, ("float32",), f64="rgb_from_hsv")


def hls_from_rgb(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("hls_from_rgb", src, dst, 3, 3, ("float32",), f64="hls_from_rgb")


def rgb_from_hls(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("rgb_from_hls", src, dst, 3, 3, ("float32",), f64="rgb_from_hls")


def sepia_from_rgb(src: Image, dst: Optional[Image] = None) -> Image:
    return _map("sepia_from_rgb", src, dst, 3, 3, ("uint8", "float32"))


def apply_colormap(src: Image, lut, dst: Optional[Image] = None) -> Image:
    """``lut``: a colour-map name / ``colormap.ColormapType`` (P/color/colormap.rs:49-100) or a uint8 array of shape
    (3, 256) = r[256], g[256], b[256] (the reference's ``ColormapLut``)."""
    if isinstance(lut, (str, _colormap.ColormapType)):
        lut = _colormap.lut(lut)
    lut = np.ascontiguousarray(lut, np.uint8).reshape(-1)
    if lut.size != 768:
        raise ImageError("InvalidArgument", "colormap LUT must hold 3 x 256 bytes")
    if not src.is_device:
        raise ImageError("HostPathUnavailable", "host images: move the image with .to_hip(stream)")
    dlut = DeviceBuffer.from_numpy(lut, src.stream)
    out = _map("apply_colormap", src, dst, 1, 3, ("uint8",), dlut.ptr)
    out._lut_keepalive = dlut  # freed stream-ordered after the launch
    return out


# ---- video formats (raw buffers <-> RGB8) ----------------------------------------------------------

def _raw_ptr(buf, need: int, what: str) -> Tuple[int, Stream]:
    if hasattr(buf, "layout") and hasattr(buf, "as_slice"):  # color_spaces.Nv12 / Yuyv8 ... typed buffers
        n, ptr, st = buf.nbytes, buf.data_ptr, buf.stream    # .data_ptr raises for host-resident buffers
    elif isinstance(buf, (Tensor, Image)):
        if not buf.is_device:
            raise ImageError("HostPathUnavailable", f"{what}: host buffer; upload it first (DeviceBuffer.from_numpy)")
        n, ptr, st = buf.nbytes, buf.data_ptr, buf.stream
    elif isinstance(buf, DeviceBuffer):
        n, ptr, st = buf.nbytes, buf.ptr, buf.stream
    else:
        raise ImageError("HostPathUnavailable", f"{what}: expected a device buffer, got {type(buf).__name__}")
    if n < need:
        raise ImageError("InvalidImageSize", f"{what}: buffer holds {n} bytes, the format needs {need}")
    return ptr, st


def _decode(kind: str, layout: int, data, width: int, height: int, dst: Optional[Image]) -> Image:
    need = width * height * 3 // 2 if kind == "planar420" else width * height * 2
    ptr, st = _raw_ptr(data, need, f"rgb_from_{kind}")
    out = dst if dst is not None else Image.uninit(width, height, 3, "uint8", st)
    _require(out, "uint8", (3,), f"rgb_from_{kind}")
    if out.size != (width, height):
        raise ImageError("InvalidImageSize", f"destination is {out.width}x{out.height}, expected {width}x{height}")
    ex = _DeviceExec(st, [out.stream])  # device_exec_for (P/cuda/dispatch.rs:150-162): raw source stream + destination image
    ex.check(getattr(lib, f"kh_rgb_from_{kind}_u8")(ex.cuda_stream_ptr, ptr, out.data_ptr, width, height, layout))
    return out


def rgb_from_video(buf, dst: Optional[Image] = None) -> Image:
    """Decode a typed camera buffer (``color_spaces.Nv12`` / ``Nv21`` / ``I420`` / ``Yv12`` / ``Yuyv8`` / ``Uyvy8`` /
    ``Yvyu8``) to RGB8 — the typed entry points of P/color/yuv/mod.rs:209-273."""
    kinds = {"nv12": ("planar420", 0), "nv21": ("planar420", 1), "i420": ("planar420", 2), "yv12": ("planar420", 3),
             "yuyv": ("packed422", 0), "uyvy": ("packed422", 1), "yvyu": ("packed422", 2)}
    kind, layout = kinds[buf.layout]
    return _decode(kind, layout, buf, buf.width, buf.height, dst)


def rgb_from_nv12(data, width, height, dst=None): return _decode("planar420", 0, data, width, height, dst)
def rgb_from_nv21(data, width, height, dst=None): return _decode("planar420", 1, data, width, height, dst)
def rgb_from_i420(data, width, height, dst=None): return _decode("planar420", 2, data, width, height, dst)
def rgb_from_yv12(data, width, height, dst=None): return _decode("planar420", 3, data, width, height, dst)
def rgb_from_yuyv(data, width, height, dst=None): return _decode("packed422", 0, data, width, height, dst)
def rgb_from_uyvy(data, width, height, dst=None): return _decode("packed422", 1, data, width, height, dst)
def rgb_from_yvyu(data, width, height, dst=None): return _decode("packed422", 2, data, width, height, dst)


def rgb_from_bayer(src, pattern: Optional[str] = None, dst: Optional[Image] = None) -> Image:
    """Bilinear, cv2-compatible demosaic of a single-channel uint8 mosaic (``rgb_from_bayer``, P/color/bayer/mod.rs:37-70;
    imgproc.pyi:58).  ``src``: a 1-channel image plus ``pattern`` in ``rggb`` | ``bggr`` | ``grbg`` | ``gbrg``, or a
    ``color_spaces.Bayer8`` that carries its pattern."""
    if hasattr(src, "pattern") and hasattr(src, "as_image"):
        pattern, src = pattern or src.pattern, src.as_image()
    code = _ffi.KH_BAYER.get(str(pattern).lower())
    if code is None:
        raise ImageError("InvalidArgument", f"rgb_from_bayer: unknown pattern {pattern!r} ({', '.join(_ffi.KH_BAYER)})")
    _require(src, "uint8", (1,), "rgb_from_bayer")
    out = dst if dst is not None else _new_like(src, channels=3)
    _require(out, "uint8", (3,), "rgb_from_bayer")
    _same_size(src, out)
    stream = _pair_residency(src, out)
    stream.check(lib.kh_rgb_from_bayer_u8(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height, code))
    return out


def convert_yuyv_to_rgb_u8(data, width: int, height: int, mode: str = "bt601_limited", dst: Optional[Image] = None) -> Image:
    """YUYV -> RGB8 with a selectable matrix: ``mode`` in ``bt601_full`` | ``bt709_full`` | ``bt601_limited``
    (``YuvToRgbMode``, P/color/yuv/mod.rs:319-410).  ``data``: a device buffer / ``color_spaces.Yuyv8`` of
    ``width * height * 2`` bytes.  An odd width leaves the last pixel of each row as it was, like the reference."""
    code = _ffi.KH_YUV_MODE.get(str(mode).lower())
    if code is None:
        raise ImageError("InvalidArgument", f"convert_yuyv_to_rgb_u8: unknown mode {mode!r} ({', '.join(_ffi.KH_YUV_MODE)})")
    ptr, st = _raw_ptr(data, width * height * 2, "convert_yuyv_to_rgb_u8")
    out = dst if dst is not None else Image.uninit(width, height, 3, "uint8", st)
    _require(out, "uint8", (3,), "convert_yuyv_to_rgb_u8")
    if out.size != (width, height):
        raise ImageError("InvalidImageSize", f"destination is {out.width}x{out.height}, expected {width}x{height}")
    ex = _DeviceExec(st, [out.stream])
    ex.check(lib.kh_yuyv_to_rgb_mode_u8(ex.cuda_stream_ptr, ptr, out.data_ptr, width, height, code))
    return out


def _encode(name: str, src: Image, nbytes: int) -> Tensor:
    _require(src, "uint8", (3,), name)
    if not src.is_device:
        raise ImageError("HostPathUnavailable", "host images: move the image with .to_hip(stream)")
    out = Tensor.uninit((nbytes,), "uint8", src.stream)
    _check(getattr(lib, f"kh_{name}_u8")(src.stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height))
    return out


def nv12_from_rgb(src: Image) -> Tensor:
    return _encode("nv12_from_rgb", src, src.width * src.height * 3 // 2)


def yuyv_from_rgb(src: Image) -> Tensor:
    return _encode("yuyv_from_rgb", src, src.width * src.height * 2)


# ---- geometry ---------------------------------------------------------------------------------------

def _interp(name: str) -> int:
    mode = _INTERP.get(str(name).lower())
    if mode is None:
        raise ImageError("NoDeviceKernel", f"interpolation {name!r} has no device kernel (nearest, bilinear, bicubic, lanczos)")
    return mode


def _geom_pair(src: Image, dst: Optional[Image], new_size: Optional[Tuple[int, int]], what: str,
               dtype: str = "float32") -> Tuple[Image, Stream]:
    # the Q10 u8 gathers exist for 2-channel images as well (P/warp/cuda.rs:458-494)
    _require(src, dtype, (1, 2, 3, 4) if dtype == "uint8" else (1, 3, 4), what)
    if dst is None:
        h, w = new_size  # the Python API takes (height, width)
        dst = _new_like(src, size=(w, h))
    _require(dst, dtype, (src.channels,), what)
    return dst, _pair_residency(src, dst)


def _u8_bilinear_only(interpolation: str, what: str) -> None:
    if str(interpolation).lower() != "bilinear":
        raise ImageError("NoDeviceKernel", f"{what}: u8 images are sampled with the Q10 bilinear kernel only "
                                           f"(got {interpolation!r}); convert to float32 for other modes")


def resize(src: Image, new_size: Optional[Tuple[int, int]] = None, interpolation: str = "bilinear",
           antialias: bool = True, out: Optional[Image] = None) -> Image:
    """``resize(image, new_size, interpolation, antialias=True, out=None)`` with ``new_size`` = (height, width), as in
    kornia_rs (imgproc.pyi:80-93).  uint8 images take the u8 cascade (resize_fast_u8_aa, P/resize/mod.rs:348; ``antialias``
    shapes its bicubic / lanczos kernels), float32 the per-pixel resize (``antialias`` is ignored there, as in the reference)."""
    if src.dtype == "uint8":
        return resize_fast(src, new_size, interpolation, bool(antialias), out)
    mode = _interp(interpolation)
    dst, stream = _geom_pair(src, out, new_size, "resize")
    stream.check(lib.kh_resize_f32(stream.cuda_stream_ptr, src.data_ptr, dst.data_ptr, src.width, src.height, dst.width,
                                   dst.height, src.channels, mode, 1, 0, 0))
    return dst


def _mapping(name: str) -> int:
    code = _ffi.KH_PIXEL_MAPPING.get(str(name).lower())
    if code is None:
        raise ImageError("InvalidArgument", f"unknown pixel mapping {name!r} (half_pixel, align_corners)")
    return code


def resize_mapped(src: Image, new_size: Tuple[int, int], interpolation: str = "bilinear", mapping: str = "half_pixel",
                  out: Optional[Image] = None) -> Image:
    """The resize LAUNCHERS with their ``PixelMapping`` argument (launch_resize_*_cuda, P/cuda/resize.rs:433-930):
    ``half_pixel`` is what ``resize`` does; ``align_corners`` maps ``src = dst * (src_len-1)/(dst_len-1)``."""
    mode = _interp(interpolation)
    dst, stream = _geom_pair(src, out, new_size, "resize")
    stream.check(lib.kh_resize_mapped_f32(stream.cuda_stream_ptr, src.data_ptr, dst.data_ptr, src.width, src.height, dst.width,
                                          dst.height, src.channels, mode, _mapping(mapping), 1, 0, 0))
    return dst


def resize_bilinear_normalize(src: Image, new_size: Tuple[int, int], mean: Sequence[float], std: Sequence[float],
                              mapping: str = "half_pixel", out: Optional[Image] = None) -> Image:
    """launch_resize_bilinear_normalize_cuda (P/cuda/resize.rs:580-650): float32 RGB bilinear resize fused with
    ``(px - mean) * (1 / std)``, HWC float32 out (one pass over HBM instead of resize + normalize_mean_std)."""
    _require(src, "float32", (3,), "resize_bilinear_normalize")
    dst, stream = _geom_pair(src, out, new_size, "resize_bilinear_normalize")
    m, s_ = _matrix(mean, 3, "resize_bilinear_normalize"), _matrix(std, 3, "resize_bilinear_normalize")
    stream.check(lib.kh_resize_bilinear_normalize_f32(stream.cuda_stream_ptr, src.data_ptr, dst.data_ptr, src.width, src.height,
                                                      dst.width, dst.height, m, s_, _mapping(mapping), 1, 0, 0))
    return dst


def _matrix(m: Sequence[float], n: int, what: str):
    m = [float(v) for v in np.asarray(m, dtype=np.float32).reshape(-1)]
    if len(m) != n:
        raise ImageError("InvalidArgument", f"{what}: matrix needs {n} entries, got {len(m)}")
    return (C.c_float * n)(*m)


def resize_fast(src: Image, new_size: Optional[Tuple[int, int]] = None, interpolation: str = "bilinear",
                antialias: bool = True, out: Optional[Image] = None) -> Image:
    """resize_fast_u8_aa (P/resize/mod.rs:348): exact-2x RGB fast paths, nearest, Q14 bilinear, Q14
    separable bicubic / lanczos; ``antialias`` widens the separable kernels on downscale (PIL semantics)."""
    mode = _interp(interpolation)
    _require(src, "uint8", (1, 2, 3, 4), "resize_fast")
    if out is None:
        h, w = new_size
        out = _new_like(src, size=(w, h))
    _require(out, "uint8", (src.channels,), "resize_fast")
    stream = _pair_residency(src, out)
    if mode != _ffi.KH_INTERP_NEAREST and src.channels == 2:
        raise ImageError("UnsupportedChannelCount", "resize_fast: 2-channel images support nearest only")
    if mode == _ffi.KH_INTERP_BILINEAR and (src.width < 2 or src.height < 2):
        raise ImageError("InvalidImageSize", f"resize_fast: bilinear needs a source of at least 2x2, got {src.width}x{src.height}")
    stream.check(lib.kh_resize_fast_u8(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height, out.width,
                                       out.height, src.channels, mode, int(bool(antialias)), 1, 0, 0))
    return out


def normalize_params(mean: Sequence[float], std: Sequence[float]) -> Tuple[np.ndarray, np.ndarray]:
    """NormalizeParams::from_mean_std (P/resize/fused.rs:29-38), f32 arithmetic:
    scale = 1 / (std * 255), bias = -mean / std."""
    m, s = np.asarray(mean, np.float32), np.asarray(std, np.float32)
    if m.shape != (3,) or s.shape != (3,):
        raise ImageError("InvalidChannelShape", "mean / std must have 3 entries")
    return (np.float32(1.0) / (s * np.float32(255.0))).astype(np.float32), (-m / s).astype(np.float32)


def resize_normalize_to_tensor(src: Image, width: int, height: int, mean: Sequence[float], std: Sequence[float],
                               interpolation: str = "bilinear", antialias: bool = True,
                               out: Optional[Tensor] = None) -> Tensor:
    """Fused resize + per-channel normalise + HWC->CHW of an RGB8 image into a ``[3, height, width]``
    float32 tensor: ``(x/255 - mean) / std`` (``Image.resize_normalize_to_tensor``, image.pyi:216;
    resize_normalize_to_tensor_u8_to_f32{_bilinear,_nearest,_separable}, P/resize/fused.rs)."""
    mode = _interp(interpolation)
    _require(src, "uint8", (3,), "resize_normalize_to_tensor")
    if not src.is_device:
        raise ImageError("HostPathUnavailable", "host images: this build provides the HIP device backend only — move "
                                                "the image with .to_hip(stream)")
    scale, bias = normalize_params(mean, std)
    if out is None:
        out = Tensor.uninit((3, height, width), "float32", src.stream)
    elif tuple(out.shape) != (3, height, width) or out.dtype != "float32" or not out.is_device:
        raise ImageError("InvalidChannelShape", f"out must be a device float32 tensor of shape (3, {height}, {width})")
    ex = _DeviceExec(src.stream, [out.stream])
    ex.check(lib.kh_resize_normalize_to_chw_u8_f32(
        ex.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height, width, height,
        scale.ctypes.data_as(C.POINTER(C.c_float)), bias.ctypes.data_as(C.POINTER(C.c_float)), mode, int(bool(antialias)),
        1, 0, 0))
    return out


def resize_opencv(src: Image, new_size: Optional[Tuple[int, int]] = None, interpolation: str = "bilinear",
                  out: Optional[Image] = None) -> Image:
    """cv2.resize-compatible nearest / bilinear for uint8 and float32 (resize_opencv_{u8,f32},
    P/resize/opencv_compat.rs:76-112)."""
    mode = _interp(interpolation)
    if mode not in (_ffi.KH_INTERP_NEAREST, _ffi.KH_INTERP_BILINEAR):
        raise ImageError("UnsupportedInterpolation", f"resize_opencv: {interpolation!r} (nearest, bilinear)")
    if src.dtype not in ("uint8", "float32"):
        raise ImageError("NoDeviceKernel", f"resize_opencv: no device kernel for {src.dtype}")
    _require(src, src.dtype, (1, 2, 3, 4), "resize_opencv")
    if out is None:
        h, w = new_size
        out = _new_like(src, size=(w, h))
    _require(out, src.dtype, (src.channels,), "resize_opencv")
    stream = _pair_residency(src, out)
    fn = lib.kh_resize_opencv_u8 if src.dtype == "uint8" else lib.kh_resize_opencv_f32
    stream.check(fn(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height, out.width, out.height,
                    src.channels, mode, 1, 0, 0))
    return out


def warp_affine(src: Image, m: Sequence[float], new_size: Optional[Tuple[int, int]] = None,
                interpolation: str = "bilinear", out: Optional[Image] = None) -> Image:
    if src.dtype == "uint8":  # warp_affine_u8, P/warp/affine.rs:373
        _u8_bilinear_only(interpolation, "warp_affine")
        dst, stream = _geom_pair(src, out, new_size, "warp_affine_u8", "uint8")
        stream.check(lib.kh_warp_affine_u8(stream.cuda_stream_ptr, src.data_ptr, dst.data_ptr, src.width, src.height, dst.width,
                                           dst.height, src.channels, _matrix(m, 6, "warp_affine"), 1, 0, 0))
        return dst
    mode = _interp(interpolation)
    dst, stream = _geom_pair(src, out, new_size, "warp_affine")
    stream.check(lib.kh_warp_affine_f32(stream.cuda_stream_ptr, src.data_ptr, dst.data_ptr, src.width, src.height, dst.width,
                                        dst.height, src.channels, _matrix(m, 6, "warp_affine"), mode, 1, 0, 0))
    return dst


def warp_perspective(src: Image, m: Sequence[float], new_size: Optional[Tuple[int, int]] = None,
                     interpolation: str = "bilinear", out: Optional[Image] = None) -> Image:
    mode = _interp(interpolation)
    mm = _matrix(m, 9, "warp_perspective")
    inv = (C.c_float * 9)()
    _check(lib.kh_invert_homography(mm, inv))  # singular matrices are rejected before anything is allocated
    if src.dtype == "uint8":  # warp_perspective_u8, P/warp/perspective.rs:179
        _u8_bilinear_only(interpolation, "warp_perspective")
        dst, stream = _geom_pair(src, out, new_size, "warp_perspective_u8", "uint8")
        stream.check(lib.kh_warp_perspective_u8(stream.cuda_stream_ptr, src.data_ptr, dst.data_ptr, src.width, src.height,
                                                dst.width, dst.height, src.channels, mm, 1, 0, 0))
        return dst
    dst, stream = _geom_pair(src, out, new_size, "warp_perspective")
    stream.check(lib.kh_warp_perspective_f32(stream.cuda_stream_ptr, src.data_ptr, dst.data_ptr, src.width, src.height,
                                             dst.width, dst.height, src.channels, mm, mode, 1, 0, 0))
    return dst


def remap(src: Image, map_x: Image, map_y: Image, interpolation: str = "bilinear", out: Optional[Image] = None) -> Image:
    mode = _interp(interpolation)
    if map_x.size != map_y.size:
        raise ImageError("InvalidImageSize", "map_x and map_y must have the same size")
    for mp in (map_x, map_y):
        _require(mp, "float32", (1,), "remap map")
    u8 = src.dtype == "uint8"  # remap_u8, P/interpolation/remap.rs:157: nearest | bilinear
    if u8 and mode not in (_ffi.KH_INTERP_NEAREST, _ffi.KH_INTERP_BILINEAR):
        raise ImageError("NoDeviceKernel", f"remap: u8 images support nearest and bilinear only (got {interpolation!r})")
    dst, stream = _geom_pair(src, out, (map_x.height, map_x.width), "remap_u8" if u8 else "remap",
                             "uint8" if u8 else "float32")
    if dst.size != map_x.size:
        raise ImageError("InvalidImageSize", "dst must have the size of the maps")
    if not (map_x.is_device and map_y.is_device):
        raise ImageError("Hip", "remap: map_x and map_y must be device-resident when src/dst are on GPU")
    for mp in (map_x, map_y):  # inputs on their own streams: fenced in, and fenced back so a later rewrite of a map waits for this read
        stream.join(mp.stream)
    fn = lib.kh_remap_u8 if u8 else lib.kh_remap_f32
    stream.check(fn(stream.cuda_stream_ptr, src.data_ptr, map_x.data_ptr, map_y.data_ptr, dst.data_ptr,
                    src.width, src.height, dst.width, dst.height, src.channels, mode, 1, 0, 0))
    return dst


# ---- batches of separately allocated images ------------------------------------------------------------
# The reference's operators take one `&Image` per call (P/resize/mod.rs:114-132, P/warp/affine.rs:123, P/warp/perspective.rs:115,
# P/interpolation/remap.rs:43, P/filter/ops.rs:39,116), so a batch is a host loop of launches — 256 launches of a 2 us kernel for
# BASELINE configs[1] — which kornia-py amortises with a captured graph (kornia-py/src/cuda_ext/mod.rs:1684-1790; here `hip.Graph`).
# The `*_batch` forms take the SAME operands as N calls would — lists of independent `Image`s, wherever each was allocated — and hand
# their device pointers to the `kh_*_list` entry points: one launch per 128 images, the (src, dst) bases in the kernel arguments.
# Results equal those of the N single calls bit for bit (tests/test_list_batches_gpu.py).

class ImageBatch(tuple):
    """N same-sized float32 device Images of ONE device, validated once, with their device pointers laid out as the host array the
    ``kh_*_list`` entry points read — what a Rust host would keep as a ``Vec<*const f32>`` beside its images.  The ``*_batch``
    operators accept plain lists too (and build this per call: a Python loop over N images); a caller that runs the same batch
    repeatedly builds it once, and a call then costs what one launch costs."""

    def __new__(cls, images: Sequence[Image], what: str = "ImageBatch", channels: Optional[Tuple[int, ...]] = None):
        self = super().__new__(cls, images)
        if not len(self):
            raise ImageError("InvalidArgument", f"{what}: empty batch")
        first = self[0]
        for im in self:   # Host / Device / Mixed classification first (P/cuda/dispatch.rs:105-133): typed, before anything is allocated
            if im.is_device != first.is_device:
                raise ImageError("MixedResidency", "the images of a batch must all be on the host or all on the device; there is no implicit transfer")
        if not first.is_device:
            raise ImageError("HostPathUnavailable", "host images: this build provides the HIP device backend only — move the images with .to_hip(stream)")
        streams, seen = [], set()
        for im in self:
            _require(im, "float32", channels if channels is not None else tuple(range(1, 9)), what)
            if im.size != first.size or im.channels != first.channels:
                raise ImageError("InvalidImageSize", f"{what}: every image of a batch must be {first.width}x{first.height}x{first.channels}, "
                                                     f"got {im.width}x{im.height}x{im.channels}")
            if im.device_id != first.device_id:
                raise ImageError("DeviceMismatch", f"images live on different devices ({first.device} vs {im.device})")
            if im.stream is None:
                raise ImageError("UnsupportedDevice", "device image without a stream (untyped foreign memory); re-wrap it with Image.from_dlpack(obj, stream=...)")
            if im.stream.cuda_stream_ptr not in seen:
                seen.add(im.stream.cuda_stream_ptr)
                streams.append(im.stream)
        self.width, self.height, self.channels = first.width, first.height, first.channels
        self.size, self.device_id, self.streams = first.size, first.device_id, streams
        self.pointers = _ffi.pointer_array([im.data_ptr for im in self])
        self.stream = first.stream          # (what hip.on_operand_device looks at)
        self.is_device = True
        return self


def _batch_pairs(srcs: Sequence[Image], outs: Optional[Sequence[Image]], new_size: Optional[Tuple[int, int]], what: str,
                 channels: Tuple[int, ...] = (1, 3, 4)):
    """N (source, destination) pairs of one batch: same size / dtype / channel count on each side, every image device-resident
    on ONE device (``ImageBatch`` checks that once); missing destinations are allocated on their source's stream.  Returns
    ``(outs, exec, src_ptrs, dst_ptrs)``: the launch goes on the FIRST source's stream with every other operand stream fenced in
    (and back by ``exec.check``), exactly what N single calls on that stream would order."""
    sb = srcs if isinstance(srcs, ImageBatch) else ImageBatch(srcs, what)
    if sb.channels not in channels:
        raise ImageError("NoDeviceKernel", f"{what}: no device kernel for float32 x {sb.channels} channels (supported: {channels})")
    if outs is None:
        if new_size is None:
            new_size = (sb.height, sb.width)
        h, w = new_size  # the Python API takes (height, width)
        outs = [_new_like(im, size=(w, h)) for im in sb]
    ob = outs if isinstance(outs, ImageBatch) else ImageBatch(outs, what)
    if len(ob) != len(sb):
        raise ImageError("InvalidArgument", f"{what}: {len(sb)} sources but {len(ob)} destinations")
    if ob.channels != sb.channels:
        raise ImageError("NoDeviceKernel", f"{what}: destinations have {ob.channels} channels, sources {sb.channels}")
    if ob.device_id != sb.device_id:
        raise ImageError("DeviceMismatch", "sources and destinations live on different devices")
    ex = _DeviceExec(sb.streams[0], sb.streams[1:] + ob.streams)
    return ob, ex, sb.pointers, ob.pointers


def resize_batch(srcs: Sequence[Image], new_size: Optional[Tuple[int, int]] = None, interpolation: str = "bilinear",
                 outs: Optional[Sequence[Image]] = None, mapping: str = "half_pixel") -> list:
    """``resize`` of N separately allocated float32 images in ceil(N / 128) launches (``new_size`` = (height, width))."""
    mode = _interp(interpolation)
    outs, ex, sp, dp = _batch_pairs(srcs, outs, new_size, "resize_batch")
    s0, d0 = srcs[0], outs[0]
    ex.check(lib.kh_resize_f32_list(ex.cuda_stream_ptr, sp, dp, len(outs), s0.width, s0.height, d0.width, d0.height, s0.channels,
                                    mode, _mapping(mapping)))
    return outs


def warp_affine_batch(srcs: Sequence[Image], m: Sequence[float], new_size: Optional[Tuple[int, int]] = None,
                      interpolation: str = "bilinear", outs: Optional[Sequence[Image]] = None) -> list:
    mode = _interp(interpolation)
    outs, ex, sp, dp = _batch_pairs(srcs, outs, new_size, "warp_affine_batch")
    s0, d0 = srcs[0], outs[0]
    ex.check(lib.kh_warp_affine_f32_list(ex.cuda_stream_ptr, sp, dp, len(outs), s0.width, s0.height, d0.width, d0.height, s0.channels,
                                         _matrix(m, 6, "warp_affine_batch"), mode))
    return outs


def warp_perspective_batch(srcs: Sequence[Image], m: Sequence[float], new_size: Optional[Tuple[int, int]] = None,
                           interpolation: str = "bilinear", outs: Optional[Sequence[Image]] = None) -> list:
    mode = _interp(interpolation)
    mm = _matrix(m, 9, "warp_perspective_batch")
    inv = (C.c_float * 9)()
    _check(lib.kh_invert_homography(mm, inv))  # singular matrices are rejected before anything is allocated
    outs, ex, sp, dp = _batch_pairs(srcs, outs, new_size, "warp_perspective_batch")
    s0, d0 = srcs[0], outs[0]
    ex.check(lib.kh_warp_perspective_f32_list(ex.cuda_stream_ptr, sp, dp, len(outs), s0.width, s0.height, d0.width, d0.height,
                                              s0.channels, mm, mode))
    return outs


def remap_batch(srcs: Sequence[Image], map_x: Image, map_y: Image, interpolation: str = "bilinear",
                outs: Optional[Sequence[Image]] = None) -> list:
    """``remap`` of N images through ONE pair of maps (one camera): the maps are read once per four images."""
    mode = _interp(interpolation)
    if map_x.size != map_y.size:
        raise ImageError("InvalidImageSize", "map_x and map_y must have the same size")
    for mp in (map_x, map_y):
        _require(mp, "float32", (1,), "remap map")
    if not (map_x.is_device and map_y.is_device):
        raise ImageError("Hip", "remap: map_x and map_y must be device-resident when src/dst are on GPU")
    outs, ex, sp, dp = _batch_pairs(srcs, outs, (map_x.height, map_x.width), "remap_batch")
    s0, d0 = srcs[0], outs[0]
    if d0.size != map_x.size:
        raise ImageError("InvalidImageSize", "dst must have the size of the maps")
    for mp in (map_x, map_y):
        ex.join(mp.stream)
    ex.check(lib.kh_remap_f32_list(ex.cuda_stream_ptr, sp, map_x.data_ptr, map_y.data_ptr, dp, len(outs), s0.width, s0.height,
                                   d0.width, d0.height, s0.channels, mode))
    return outs


def _same_size_batch(srcs, outs, what):
    outs, ex, sp, dp = _batch_pairs(srcs, outs, None, what, channels=tuple(range(1, 9)))
    if outs[0].size != srcs[0].size:
        raise ImageError("InvalidImageSize", f"{what}: destinations must have the sources' size")
    return outs, ex, sp, dp


def gaussian_blur_batch(srcs: Sequence[Image], kernel_size: Tuple[int, int], sigma: Tuple[float, float],
                        outs: Optional[Sequence[Image]] = None) -> list:
    k = (C.c_int32 * 2)(*kernel_size)
    s = (C.c_float * 2)(*sigma)
    if lib.kh_gaussian_resolve(k, s) != _ffi.KH_OK:
        raise ImageError("InvalidSigmaValue", _ffi.last_error())
    outs, ex, sp, dp = _same_size_batch(srcs, outs, "gaussian_blur_batch")
    s0 = srcs[0]
    ex.check(lib.kh_gaussian_blur_f32_list(ex.cuda_stream_ptr, sp, dp, len(outs), s0.width, s0.height, s0.channels, kernel_size[0],
                                           kernel_size[1], sigma[0], sigma[1]))
    return outs


def box_blur_batch(srcs: Sequence[Image], kernel_size: Tuple[int, int], outs: Optional[Sequence[Image]] = None) -> list:
    outs, ex, sp, dp = _same_size_batch(srcs, outs, "box_blur_batch")
    s0 = srcs[0]
    ex.check(lib.kh_box_blur_f32_list(ex.cuda_stream_ptr, sp, dp, len(outs), s0.width, s0.height, s0.channels, kernel_size[0], kernel_size[1]))
    return outs


def sobel_batch(srcs: Sequence[Image], kernel_size: int = 3, outs: Optional[Sequence[Image]] = None) -> list:
    if kernel_size not in (3, 5):
        raise ImageError("InvalidKernelLength", f"sobel_batch: invalid kernel length ({kernel_size}, {kernel_size})")
    outs, ex, sp, dp = _same_size_batch(srcs, outs, "sobel_batch")
    s0 = srcs[0]
    ex.check(lib.kh_gradient_magnitude_f32_list(ex.cuda_stream_ptr, sp, dp, len(outs), s0.width, s0.height, s0.channels,
                                                _ffi.KH_GRAD_SOBEL, kernel_size))
    return outs


def separable_filter_batch(srcs: Sequence[Image], kernel_x: Sequence[float], kernel_y: Sequence[float],
                           outs: Optional[Sequence[Image]] = None) -> list:
    kx, ky = np.asarray(kernel_x, np.float32), np.asarray(kernel_y, np.float32)
    outs, ex, sp, dp = _same_size_batch(srcs, outs, "separable_filter_batch")
    s0 = srcs[0]
    ex.check(lib.kh_separable_filter_f32_list(ex.cuda_stream_ptr, sp, dp, len(outs), s0.width, s0.height, s0.channels,
                                              (C.c_float * kx.size)(*kx), kx.size, (C.c_float * ky.size)(*ky), ky.size))
    return outs


def generate_correction_map_polynomial(intrinsic: Sequence[float], distortion: Sequence[float], size: Tuple[int, int],
                                       stream: Stream) -> Tuple[Image, Image]:
    """``intrinsic`` = (fx, fy, cx, cy), ``distortion`` = (k1..k6, p1, p2), ``size`` = (width, height);
    returns device ``(map_x, map_y)`` (P/calibration/distortion.rs:135-152)."""
    w, h = size
    mx, my = Image.uninit(w, h, 1, "float32", stream), Image.uninit(w, h, 1, "float32", stream)
    _check(lib.kh_correction_map_polynomial_f32(stream.cuda_stream_ptr, mx.data_ptr, my.data_ptr, w, h,
                                                (C.c_double * 4)(*intrinsic), (C.c_double * 8)(*distortion)))
    return mx, my


def get_rotation_matrix2d(center: Tuple[float, float], angle: float, scale: float):
    out = (C.c_float * 6)()
    lib.kh_get_rotation_matrix2d(center[0], center[1], angle, scale, out)
    return [float(v) for v in out]


def invert_affine_transform(m: Sequence[float]):
    out = (C.c_float * 6)()
    lib.kh_invert_affine_transform(_matrix(m, 6, "invert_affine_transform"), out)
    return [float(v) for v in out]


# ---- filters ----------------------------------------------------------------------------------------

def _filter_pair(src: Image, dst: Optional[Image], what: str, dtype: str = "float32") -> Tuple[Image, Stream]:
    _require(src, dtype, tuple(range(1, 9)) if dtype == "float32" else (1, 3, 4), what)
    out = dst if dst is not None else _new_like(src)
    _require(out, dtype, (src.channels,), what)
    _same_size(src, out)
    return out, _pair_residency(src, out)


def gaussian_blur(src: Image, kernel_size: Tuple[int, int], sigma: Tuple[float, float], dst: Optional[Image] = None) -> Image:
    k = (C.c_int32 * 2)(*kernel_size)
    s = (C.c_float * 2)(*sigma)
    if lib.kh_gaussian_resolve(k, s) != _ffi.KH_OK:
        raise ImageError("InvalidSigmaValue", _ffi.last_error())
    if src.dtype == "uint8":  # gaussian_blur_u8, P/filter/ops.rs:639
        out, stream = _filter_pair(src, dst, "gaussian_blur_u8", "uint8")
        stream.check(lib.kh_gaussian_blur_u8(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height,
                                             src.channels, kernel_size[0], kernel_size[1], sigma[0], sigma[1], 1, 0, 0))
        return out
    out, stream = _filter_pair(src, dst, "gaussian_blur")
    stream.check(lib.kh_gaussian_blur_f32(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height,
                                          src.channels, kernel_size[0], kernel_size[1], sigma[0], sigma[1], 1, 0, 0))
    return out


def box_blur(src: Image, kernel_size: Tuple[int, int], dst: Optional[Image] = None) -> Image:
    if src.dtype == "uint8":  # box_blur_u8, P/filter/ops.rs:59
        if not all(int(k) > 0 and int(k) % 2 == 1 for k in kernel_size):
            raise ImageError("InvalidKernelLength", f"box_blur: invalid kernel length {tuple(kernel_size)} (u8 needs odd sizes)")
        out, stream = _filter_pair(src, dst, "box_blur_u8", "uint8")
        stream.check(lib.kh_box_blur_u8(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height, src.channels,
                                        kernel_size[0], kernel_size[1], 1, 0, 0))
        return out
    out, stream = _filter_pair(src, dst, "box_blur")
    stream.check(lib.kh_box_blur_f32(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height, src.channels,
                                     kernel_size[0], kernel_size[1], 1, 0, 0))
    return out


def _gradient(src: Image, kind: int, kernel_size: int, dst: Optional[Image], what: str) -> Image:
    ok = kernel_size in ((3, 5) if kind == _ffi.KH_GRAD_SOBEL else (3,))
    if not ok:
        raise ImageError("InvalidKernelLength", f"{what}: invalid kernel length ({kernel_size}, {kernel_size})")
    out, stream = _filter_pair(src, dst, what)
    stream.check(lib.kh_gradient_magnitude_f32(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height,
                                               src.channels, kind, kernel_size, 1, 0, 0))
    return out


def sobel(src: Image, kernel_size: int = 3, dst: Optional[Image] = None) -> Image:
    return _gradient(src, _ffi.KH_GRAD_SOBEL, kernel_size, dst, "sobel")


def scharr(src: Image, kernel_size: int = 3, dst: Optional[Image] = None) -> Image:
    return _gradient(src, _ffi.KH_GRAD_SCHARR, kernel_size, dst, "scharr")


def separable_filter(src: Image, kernel_x: Sequence[float], kernel_y: Sequence[float], dst: Optional[Image] = None) -> Image:
    kx, ky = np.asarray(kernel_x, np.float32), np.asarray(kernel_y, np.float32)
    out, stream = _filter_pair(src, dst, "separable_filter")
    stream.check(lib.kh_separable_filter_f32(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height,
                                             src.channels, (C.c_float * kx.size)(*kx), kx.size, (C.c_float * ky.size)(*ky),
                                             ky.size, 1, 0, 0))
    return out


def _spatial_gradient(src: Image, kind: int, dx: Optional[Image], dy: Optional[Image], what: str) -> Tuple[Image, Image]:
    _require(src, "float32", tuple(range(1, 9)), what)
    gx = dx if dx is not None else _new_like(src)
    gy = dy if dy is not None else _new_like(src)
    for g in (gx, gy):
        _require(g, "float32", (src.channels,), what)
        _same_size(src, g)
    stream = _pair_residency(src, gx, gy)
    stream.check(lib.kh_spatial_gradient_f32(stream.cuda_stream_ptr, src.data_ptr, gx.data_ptr, gy.data_ptr, src.width, src.height,
                                             src.channels, kind, 1, 0, 0))
    return gx, gy


def spatial_gradient_float(src: Image, dx: Optional[Image] = None, dy: Optional[Image] = None) -> Tuple[Image, Image]:
    """First-order derivatives with the normalised 3x3 Sobel operator (spatial_gradient_float and its _parallel twins,
    P/filter/ops.rs:287-509).  Returns ``(dx, dy)``."""
    return _spatial_gradient(src, _ffi.KH_GRAD_SOBEL, dx, dy, "spatial_gradient_float")


def scharr_spatial_gradient_float(src: Image, dx: Optional[Image] = None, dy: Optional[Image] = None) -> Tuple[Image, Image]:
    """The Scharr twin (scharr_spatial_gradient_float, P/filter/ops.rs:511-590)."""
    return _spatial_gradient(src, _ffi.KH_GRAD_SCHARR, dx, dy, "scharr_spatial_gradient_float")


def box_blur_fast_kernels_1d(sigma: float, kernels: int):
    out = (C.c_int32 * max(int(kernels), 1))()
    _check(lib.kh_box_blur_fast_kernels_1d(sigma, kernels, out))
    return [int(v) for v in out[:kernels]]


def box_blur_fast(src: Image, sigma: Tuple[float, float], dst: Optional[Image] = None) -> Image:
    """Three running-sum box passes per axis approximating a gaussian of ``sigma`` (box_blur_fast, P/filter/ops.rs:252-285).
    The transposed intermediate is a stream-ordered scratch image, released on the stream after the last pass."""
    out, stream = _filter_pair(src, dst, "box_blur_fast")
    scratch = _new_like(src)
    stream.check(lib.kh_box_blur_fast_f32(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, scratch.data_ptr, src.width, src.height,
                                          src.channels, sigma[0], sigma[1], 1, 0, 0))
    return out


def median_blur(image: Image, kernel_size: int = 3, dst: Optional[Image] = None) -> Image:
    """cv2.medianBlur-compatible median (replicate border), kernel 3 or 5, uint8 x 1..4 channels (median_blur,
    P/filter/median.rs:174; kornia-py imgproc.pyi:169)."""
    if kernel_size not in (3, 5):
        raise ImageError("InvalidKernelLength", f"median_blur: invalid kernel length ({kernel_size}, {kernel_size})")
    _require(image, "uint8", (1, 2, 3, 4), "median_blur")
    out = dst if dst is not None else _new_like(image)
    _require(out, "uint8", (image.channels,), "median_blur")
    _same_size(image, out)
    stream = _pair_residency(image, out)
    stream.check(lib.kh_median_blur_u8(stream.cuda_stream_ptr, image.data_ptr, out.data_ptr, image.width, image.height, image.channels,
                                       kernel_size, 1, 0, 0))
    return out


def bilateral_filter(image: Image, d: int = 5, sigma_color: float = 50.0, sigma_space: float = 50.0, dst: Optional[Image] = None) -> Image:
    """cv2.bilateralFilter-compatible bilateral filter for single-channel uint8 (bilateral_filter, P/filter/bilateral.rs:172;
    kornia-py imgproc.pyi:174)."""
    _require(image, "uint8", (1,), "bilateral_filter")
    out = dst if dst is not None else _new_like(image)
    _require(out, "uint8", (1,), "bilateral_filter")
    _same_size(image, out)
    stream = _pair_residency(image, out)
    stream.check(lib.kh_bilateral_filter_u8(stream.cuda_stream_ptr, image.data_ptr, out.data_ptr, image.width, image.height, int(d),
                                            float(sigma_color), float(sigma_space), 1, 0, 0))
    return out


def bilateral_tables(d: int, sigma_color: float, sigma_space: float) -> dict:
    """The tables cv2 would build (BilateralTables / build_tables, P/filter/bilateral.rs:80-170); host-only."""
    radius, n = C.c_int32(0), C.c_int32(0)
    _check(lib.kh_bilateral_tables(d, sigma_color, sigma_space, 0, C.byref(radius), C.byref(n), None, None, None, None, None))
    dy, dx, order = (np.empty(n.value, np.int32) for _ in range(3))
    space, color = np.empty(n.value, np.float32), np.empty(256, np.float32)
    _check(lib.kh_bilateral_tables(d, sigma_color, sigma_space, n.value, C.byref(radius), C.byref(n), dy.ctypes.data, dx.ctypes.data,
                                   space.ctypes.data, color.ctypes.data, order.ctypes.data))
    return {"radius": radius.value, "taps": list(zip(dy.tolist(), dx.tolist())), "space_weight": space, "color_weight": color,
            "simd_order": order.tolist()}


# ---- normalize / crop / flip ----------------------------------------------------------------------------

def normalize_mean_std(src: Image, mean: Sequence[float], std: Sequence[float], dst: Optional[Image] = None) -> Image:
    _require(src, "float32", (1, 2, 3, 4), "normalize_mean_std")
    if len(mean) != src.channels or len(std) != src.channels:
        raise ImageError("InvalidChannelShape", "mean/std need one entry per channel")
    out = dst if dst is not None else _new_like(src)
    _require(out, "float32", (src.channels,), "normalize_mean_std")
    _same_size(src, out)
    stream = _pair_residency(src, out)
    stream.check(lib.kh_normalize_mean_std_f32(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width * src.height,
                                               src.channels, (C.c_float * src.channels)(*mean), (C.c_float * src.channels)(*std)))
    return out


def normalize_rgb_u8(src: Image, scale: Sequence[float], offset: Sequence[float], dst: Optional[Image] = None) -> Image:
    """uint8 RGB -> float32 RGB, ``x * scale[c] + offset[c]`` (normalize_rgb_u8, P/normalize.rs:235-260)."""
    _require(src, "uint8", (3,), "normalize_rgb_u8")
    out = dst if dst is not None else _new_like(src, dtype="float32")
    _require(out, "float32", (3,), "normalize_rgb_u8")
    _same_size(src, out)
    stream = _pair_residency(src, out)
    stream.check(lib.kh_normalize_rgb_u8_f32(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width * src.height,
                                             _matrix(scale, 3, "normalize_rgb_u8"), _matrix(offset, 3, "normalize_rgb_u8")))
    return out


def find_min_max(src: Image) -> Tuple[float, float]:
    _require(src, "float32", tuple(range(1, 9)), "find_min_max")
    if not src.is_device:
        raise ImageError("HostPathUnavailable", "host images: move the image with .to_hip(stream)")
    n = src.width * src.height * src.channels
    if n == 0:
        raise ImageError("ImageDataNotInitialized", "image data is not initialized")
    mm, scratch = Tensor.uninit((2,), "float32", src.stream), Tensor.uninit((2,), "int32", src.stream)
    _check(lib.kh_find_min_max_f32(src.stream.cuda_stream_ptr, src.data_ptr, n, mm.data_ptr, scratch.data_ptr))
    lo, hi = mm.numpy()
    return float(lo), float(hi)


def normalize_min_max(src: Image, min: float, max: float, dst: Optional[Image] = None) -> Image:
    _require(src, "float32", tuple(range(1, 9)), "normalize_min_max")
    out = dst if dst is not None else _new_like(src)
    _require(out, "float32", (src.channels,), "normalize_min_max")
    _same_size(src, out)
    stream = _pair_residency(src, out)
    n = src.width * src.height * src.channels
    if n == 0:
        raise ImageError("ImageDataNotInitialized", "image data is not initialized")
    mm, scratch = Tensor.uninit((2,), "float32", stream), Tensor.uninit((2,), "int32", stream)
    stream.check(lib.kh_normalize_min_max_f32(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, n, min, max, mm.data_ptr,
                                              scratch.data_ptr))
    out._scratch_keepalive = (mm, scratch)
    return out


def crop_image(src: Image, x: int, y: int, width: int, height: int, dst: Optional[Image] = None) -> Image:
    out = dst if dst is not None else _new_like(src, size=(width, height))
    if out.dtype != src.dtype or out.channels != src.channels or out.size != (width, height):
        raise ImageError("InvalidImageSize", "crop destination must be width x height with the source's type")
    if x + width > src.width or y + height > src.height:
        raise ImageError("PixelIndexOutOfBounds", f"pixel index out of bounds: ({x + width}, {y + height}) exceeds "
                                                  f"{src.width}x{src.height}")
    stream = _pair_residency(src, out)
    pb = src.channels * np.dtype(src.dtype).itemsize
    stream.check(lib.kh_crop(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height, width, height, x, y, pb))
    return out


def _flip(src: Image, dst: Optional[Image], horizontal: int) -> Image:
    out = dst if dst is not None else _new_like(src)
    if out.dtype != src.dtype or out.channels != src.channels:
        raise ImageError("InvalidChannelShape", "flip destination must have the source's type")
    _same_size(src, out)
    stream = _pair_residency(src, out)
    pb = src.channels * np.dtype(src.dtype).itemsize
    stream.check(lib.kh_flip(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height, pb, horizontal))
    return out


def horizontal_flip(src: Image, dst: Optional[Image] = None) -> Image:
    return _flip(src, dst, 1)


def vertical_flip(src: Image, dst: Optional[Image] = None) -> Image:
    return _flip(src, dst, 0)


# Rust-API names of the u8 twins (P/warp/affine.rs:373, P/warp/perspective.rs:179,
# P/interpolation/remap.rs:157, P/filter/ops.rs:59,639): same functions, dtype-dispatched above.
def _u8_only(fn, name):
    def wrapper(src: Image, *args, **kwargs):
        if src.dtype != "uint8":
            raise ImageError("NoDeviceKernel", f"{name}: expects a uint8 image, got {src.dtype}")
        return fn(src, *args, **kwargs)
    wrapper.__name__ = name
    return wrapper


warp_affine_u8 = _u8_only(warp_affine, "warp_affine_u8")
warp_perspective_u8 = _u8_only(warp_perspective, "warp_perspective_u8")
remap_u8 = _u8_only(remap, "remap_u8")
gaussian_blur_u8 = _u8_only(gaussian_blur, "gaussian_blur_u8")
box_blur_u8 = _u8_only(box_blur, "box_blur_u8")


# ---- pyramid + morphology -----------------------------------------------------------------------------------

def _pyr(src: Image, dst: Optional[Image], up: bool, what: str) -> Image:
    if src.dtype not in ("float32", "uint8"):
        raise ImageError("NoDeviceKernel", f"{what}: no device kernel for {src.dtype}")
    _require(src, src.dtype, (1, 3, 4), what)
    w, h = (src.width * 2, src.height * 2) if up else ((src.width + 1) // 2, (src.height + 1) // 2)
    out = dst if dst is not None else _new_like(src, size=(w, h))
    _require(out, src.dtype, (src.channels,), what)
    if out.size != (w, h):  # P/pyramid.rs:216-224, 318-326
        raise ImageError("InvalidImageSize", f"{what}: expected a {w}x{h} destination, got {out.width}x{out.height}")
    stream = _pair_residency(src, out)
    fn = getattr(lib, f"kh_{'pyrup' if up else 'pyrdown'}_{'f32' if src.dtype == 'float32' else 'u8'}")
    stream.check(fn(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height, src.channels, 1, 0, 0))
    return out


def pyrdown(src: Image, dst: Optional[Image] = None) -> Image:
    """pyrdown_f32 / pyrdown_u8 (P/pyramid.rs:312, 469): 5x5 Gaussian + 2x decimation, reflect-101."""
    return _pyr(src, dst, False, "pyrdown")


def pyrup(src: Image, dst: Optional[Image] = None) -> Image:
    """pyrup_f32 / pyrup_u8 (P/pyramid.rs:210, 804): 2x Gaussian upsampling."""
    return _pyr(src, dst, True, "pyrup")


def build_pyramid(src: Image, max_level: int) -> list:
    """build_pyramid (P/pyramid.rs:431-452): [src, pyrdown(src), ...] with max_level + 1 entries."""
    levels = [src]
    for _ in range(max_level):
        levels.append(pyrdown(levels[-1]))
    return levels


class Kernel:
    """Morphological structuring element (P/morphology/kernels.rs:60-110): ``Kernel("box", 3)``,
    ``Kernel("cross", 5)``, ``Kernel("ellipse", (w, h))`` or ``Kernel.from_mask(array)``."""

    def __init__(self, shape: str, size):
        code = _ffi.KH_MORPH_SHAPE.get(str(shape).lower())
        if code is None:
            raise ImageError("InvalidKernelShape", f"unknown kernel shape {shape!r} (box, cross, ellipse)")
        w, h = (size, size) if np.isscalar(size) else size
        self.width, self.height = int(w), int(h)
        buf = (C.c_uint8 * (self.width * self.height))()
        _check(lib.kh_morph_kernel(code, self.width, self.height, buf))
        self.data = np.frombuffer(buf, np.uint8).reshape(self.height, self.width).copy()

    @staticmethod
    def from_mask(mask: np.ndarray) -> "Kernel":
        k = Kernel.__new__(Kernel)
        k.data = np.ascontiguousarray(mask, np.uint8)
        k.height, k.width = k.data.shape
        return k

    def pad(self) -> Tuple[int, int]:
        return self.height // 2, self.width // 2


def _morph(src: Image, kernel: Kernel, op: int, padding_mode: str, constant_value, dst: Optional[Image], what: str) -> Image:
    _require(src, "uint8", (1, 3, 4), what)  # only u8 has a device kernel (P/morphology/cuda.rs:62-64)
    out = dst if dst is not None else _new_like(src)
    _require(out, "uint8", (src.channels,), what)
    if out.size != src.size:
        raise ImageError("InvalidImageSize", f"{what}: image sizes differ: {src.width}x{src.height} vs {out.width}x{out.height}")
    border = _ffi.KH_BORDER.get(str(padding_mode).lower())
    if border is None:
        raise ImageError("InvalidPaddingMode", f"unknown padding mode {padding_mode!r}")
    cv = np.zeros(4, np.uint8)
    vals = np.atleast_1d(np.asarray(constant_value if constant_value is not None else 0))
    cv[: src.channels] = np.broadcast_to(vals, (src.channels,)) if vals.size == 1 else vals[: src.channels]
    stream = _pair_residency(src, out)
    mask = np.ascontiguousarray(kernel.data, np.uint8)
    stream.check(lib.kh_morphology_u8(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width, src.height, src.channels, op,
                                      mask.ctypes.data_as(C.POINTER(C.c_uint8)), kernel.width, kernel.height, border,
                                      cv.ctypes.data_as(C.POINTER(C.c_uint8)), 1, 0, 0))
    return out


def _kernel_arg(kernel, size) -> Kernel:
    """``Kernel`` object (Rust API, P/morphology/ops.rs:120) or the kornia-py spelling: a shape name plus ``size`` =
    (height, width) (imgproc.pyi:127-147)."""
    if isinstance(kernel, Kernel):
        return kernel
    h, w = size
    if str(kernel).lower() in ("box", "cross") and h != w:  # parse_kernel, kornia-py/src/morphology.rs:9-27
        raise ImageError("InvalidArgument", f"{kernel} kernel requires a square size")
    return Kernel(kernel, (w, h))


def _border_arg(kernel, padding_mode: Optional[str], border: Optional[str]) -> str:
    """The Rust API always names its PaddingMode; the Python stubs default ``border`` to "replicate" (morphology.rs:60)."""
    return border or padding_mode or ("constant" if isinstance(kernel, Kernel) else "replicate")


def dilate(src: Image, kernel="box", padding_mode: Optional[str] = None, constant_value=0, dst: Optional[Image] = None, *,
           size: Tuple[int, int] = (3, 3), border: Optional[str] = None) -> Image:
    return _morph(src, _kernel_arg(kernel, size), _ffi.KH_MORPH_DILATE, _border_arg(kernel, padding_mode, border), constant_value,
                  dst, "dilate")


def erode(src: Image, kernel="box", padding_mode: Optional[str] = None, constant_value=0, dst: Optional[Image] = None, *,
          size: Tuple[int, int] = (3, 3), border: Optional[str] = None) -> Image:
    return _morph(src, _kernel_arg(kernel, size), _ffi.KH_MORPH_ERODE, _border_arg(kernel, padding_mode, border), constant_value,
                  dst, "erode")


def morph_open(src: Image, kernel: Kernel, padding_mode: str = "constant", constant_value=0, dst: Optional[Image] = None) -> Image:
    """open (P/morphology/ops.rs:227-240): erode into a temporary, then dilate."""
    return dilate(erode(src, kernel, padding_mode, constant_value), kernel, padding_mode, constant_value, dst)


def morph_close(src: Image, kernel: Kernel, padding_mode: str = "constant", constant_value=0, dst: Optional[Image] = None) -> Image:
    """close (P/morphology/ops.rs:255-268): dilate into a temporary, then erode."""
    return erode(dilate(src, kernel, padding_mode, constant_value), kernel, padding_mode, constant_value, dst)


# ---- CIE colour spaces (P/color/cie/mod.rs:58-160) ---------------------------------------------------------

def _cie(name: str):
    code = _ffi.KH_CIE[name]

    def conv(src: Image, dst: Optional[Image] = None) -> Image:
        if src.dtype == "float64":  # the `*_scalar64` formulas (P/color/cie/kernels.rs:64-215)
            return _map_f64(name, src, dst, 3, 3)
        _require(src, "float32", (3,), name)
        out = dst if dst is not None else _new_like(src)
        _require(out, "float32", (3,), name)
        _same_size(src, out)
        stream = _pair_residency(src, out)
        stream.check(lib.kh_cie_convert_f32(stream.cuda_stream_ptr, src.data_ptr, out.data_ptr, src.width * src.height, code))
        return out

    conv.__name__ = name
    conv.__doc__ = f"``{name}`` on float32 / float64 RGB in [0, 1] (D65, OpenCV coefficients)."
    return conv


linear_rgb_from_rgb = _cie("linear_rgb_from_rgb")
rgb_from_linear_rgb = _cie("rgb_from_linear_rgb")
xyz_from_rgb = _cie("xyz_from_rgb")
rgb_from_xyz = _cie("rgb_from_xyz")
lab_from_rgb = _cie("lab_from_rgb")
rgb_from_lab = _cie("rgb_from_lab")
luv_from_rgb = _cie("luv_from_rgb")
rgb_from_luv = _cie("rgb_from_luv")


crop = crop_image  # short name used by the Python stubs; `crop_image` is the Rust name (P/crop.rs:187)


def _bind_public_operators() -> None:
    """Every public operator of this module runs with its first device operand's device current (``hip.on_operand_device``): the
    launch, its scratch / table lookups and a NULL stream handle all refer to "the current device"."""
    import inspect
    from .hip import on_operand_device
    g, bound = globals(), {}
    for name, obj in list(g.items()):
        if not name.startswith("_") and inspect.isfunction(obj) and obj.__module__ == __name__:
            if obj not in bound:          # aliases (crop is crop_image) stay one object
                bound[obj] = on_operand_device(obj)
            g[name] = bound[obj]


_bind_public_operators()
