"""Colour-space tags and the packed / planar camera buffer types — host mirror of
``crates/kornia-image/src/color_spaces.rs`` (ColorSpace :19-80, typed wrappers :269-620, video buffers
:630-830).

``ColorSpace`` tags what the channels of an ``Image`` mean; the typed constructors (``Rgb8``, ``Grayf32`` ...)
are ``Image`` factories that check dtype and channel count.  ``Nv12`` / ``Nv21`` / ``I420`` / ``Yv12`` and
``Yuyv8`` / ``Uyvy8`` / ``Yvyu8`` carry one byte buffer with a sub-sampled layout and validate its length
and the even-dimension rules on construction; ``to_hip`` uploads the buffer for the device decoders
(``imgproc.rgb_from_*`` accept them directly).

``convert(src, to)`` is the ``ConvertColor`` trait (crates/kornia-imgproc/src/color/convert.rs:31-273): the source's
colour-space tag, its dtype and the requested space select ONE conversion from the same table of impls; a pair
without an impl is an error, never a multi-hop route.
"""
from __future__ import annotations

import enum
from typing import Optional

import numpy as np

from .hip import DeviceBuffer, Stream
from .image import Image, ImageError


class ColorSpace(enum.Enum):
    RGB = "rgb"; BGR = "bgr"; GRAY = "gray"; RGBA = "rgba"; BGRA = "bgra"; HSV = "hsv"; HLS = "hls"
    LAB = "lab"; LUV = "luv"; XYZ = "xyz"; LINEAR_RGB = "linear_rgb"; YCBCR = "ycbcr"; YUV = "yuv"

    @property
    def channels(self) -> int:  # color_spaces.rs:48-58
        return {ColorSpace.GRAY: 1, ColorSpace.RGBA: 4, ColorSpace.BGRA: 4}.get(self, 3)

    @property
    def float_only(self) -> bool:  # :60-75
        return self in (ColorSpace.HSV, ColorSpace.HLS, ColorSpace.LAB, ColorSpace.LUV, ColorSpace.XYZ, ColorSpace.LINEAR_RGB)


def _typed(name: str, dtype: str, space: ColorSpace):
    def make(data: np.ndarray) -> Image:
        a = np.asarray(data)
        if a.ndim == 2 and space.channels == 1:
            a = a[:, :, None]
        if a.ndim != 3 or a.shape[2] != space.channels or a.dtype != np.dtype(dtype):
            raise ImageError("InvalidChannelShape",
                             f"{name} needs a [H, W, {space.channels}] {dtype} array, got {a.shape} {a.dtype}")
        return Image(Image.from_numpy(np.ascontiguousarray(a)).tensor, space)
    make.__name__ = name
    make.color_space = space
    return make


Rgb8, Bgr8, Gray8 = _typed("Rgb8", "uint8", ColorSpace.RGB), _typed("Bgr8", "uint8", ColorSpace.BGR), _typed("Gray8", "uint8", ColorSpace.GRAY)
Rgba8, Bgra8 = _typed("Rgba8", "uint8", ColorSpace.RGBA), _typed("Bgra8", "uint8", ColorSpace.BGRA)
YCbCr8, Yuv8 = _typed("YCbCr8", "uint8", ColorSpace.YCBCR), _typed("Yuv8", "uint8", ColorSpace.YUV)
Rgbf32, Bgrf32, Grayf32 = _typed("Rgbf32", "float32", ColorSpace.RGB), _typed("Bgrf32", "float32", ColorSpace.BGR), _typed("Grayf32", "float32", ColorSpace.GRAY)
Hsvf32, Hlsf32, Labf32 = _typed("Hsvf32", "float32", ColorSpace.HSV), _typed("Hlsf32", "float32", ColorSpace.HLS), _typed("Labf32", "float32", ColorSpace.LAB)
Luvf32, Xyzf32, LinearRgbf32 = _typed("Luvf32", "float32", ColorSpace.LUV), _typed("Xyzf32", "float32", ColorSpace.XYZ), _typed("LinearRgbf32", "float32", ColorSpace.LINEAR_RGB)
YCbCrf32, Yuvf32 = _typed("YCbCrf32", "float32", ColorSpace.YCBCR), _typed("Yuvf32", "float32", ColorSpace.YUV)
# f64 newtypes (color_spaces.rs:269-620; device conversions: P/color/convert.rs:110-218 f64 impls)
Rgbf64, Grayf64 = _typed("Rgbf64", "float64", ColorSpace.RGB), _typed("Grayf64", "float64", ColorSpace.GRAY)
Hsvf64, Hlsf64, Labf64 = _typed("Hsvf64", "float64", ColorSpace.HSV), _typed("Hlsf64", "float64", ColorSpace.HLS), _typed("Labf64", "float64", ColorSpace.LAB)
Luvf64, Xyzf64, LinearRgbf64 = _typed("Luvf64", "float64", ColorSpace.LUV), _typed("Xyzf64", "float64", ColorSpace.XYZ), _typed("LinearRgbf64", "float64", ColorSpace.LINEAR_RGB)
YCbCrf64, Yuvf64 = _typed("YCbCrf64", "float64", ColorSpace.YCBCR), _typed("Yuvf64", "float64", ColorSpace.YUV)


class _VideoBuffer:
    """One byte buffer + (width, height); subclasses fix the layout.  Host- or device-resident."""
    layout = ""
    _num, _den = 3, 2           # bytes = width * height * num / den
    _even_height = True

    def __init__(self, width: int, height: int, data, *, _device: Optional[DeviceBuffer] = None):
        self.width, self.height = int(width), int(height)
        expected = self.width * self.height * self._num // self._den
        n = _device.nbytes if _device is not None else np.asarray(data).size
        if n != expected or self.width % 2 or (self._even_height and self.height % 2) or self.width <= 0 or self.height <= 0:
            raise ImageError("InvalidImageSize", f"{type(self).__name__}: {n} bytes for {self.width}x{self.height} "
                                                 f"(expected {expected}; width{' and height' if self._even_height else ''} must be even)")
        self._dev = _device
        self._host = None if _device is not None else np.ascontiguousarray(np.asarray(data, np.uint8)).reshape(-1)
        self.stream: Optional[Stream] = None

    @classmethod
    def from_size_vec(cls, size, data):  # color_spaces.rs:737-750, 660-672
        return cls(size[0], size[1], data)

    @property
    def size(self): return (self.width, self.height)

    @property
    def is_device(self) -> bool: return self._dev is not None

    def as_slice(self) -> np.ndarray:
        if self._host is None:
            raise ImageError("UnsupportedDevice", "host access to a device-resident buffer; call .cpu() first")
        return self._host

    def to_hip(self, stream: Stream):
        if self.is_device:
            return self
        out = type(self)(self.width, self.height, None, _device=DeviceBuffer.from_numpy(self._host, stream))
        out.stream = stream
        return out

    def cpu(self):
        if not self.is_device:
            return self
        return type(self)(self.width, self.height, self._dev.to_numpy(np.uint8, (self._dev.nbytes,)))

    # what imgproc's decoders consume
    @property
    def data_ptr(self) -> int:
        if self._dev is None:
            raise ImageError("HostPathUnavailable", "host buffers: this build provides the HIP device backend only — "
                                                    "move the buffer with .to_hip(stream)")
        return self._dev.ptr

    @property
    def nbytes(self) -> int: return self.width * self.height * self._num // self._den


class Nv12(_VideoBuffer): layout = "nv12"
class Nv21(_VideoBuffer): layout = "nv21"
class I420(_VideoBuffer): layout = "i420"
class Yv12(_VideoBuffer): layout = "yv12"


class _Packed422(_VideoBuffer):
    _num, _den, _even_height = 2, 1, False


class Yuyv8(_Packed422): layout = "yuyv"
class Uyvy8(_Packed422): layout = "uyvy"
class Yvyu8(_Packed422): layout = "yvyu"




class Bayer8:
    """Single-channel 8-bit mosaic tagged with its ``BayerPattern`` (color_spaces.rs:853-900): ``rggb`` | ``bggr`` | ``grbg`` |
    ``gbrg``.  ``convert(bayer, Rgb8)`` / ``imgproc.rgb_from_bayer(bayer)`` demosaic it."""
    PATTERNS = ("rggb", "bggr", "grbg", "gbrg")

    def __init__(self, data, pattern: str):
        self.pattern = str(pattern).lower()
        if self.pattern not in self.PATTERNS:
            raise ImageError("InvalidArgument", f"Bayer8: unknown pattern {pattern!r} ({', '.join(self.PATTERNS)})")
        if isinstance(data, Image):
            img = data
        else:
            a = np.asarray(data)
            if a.ndim == 2:
                a = a[:, :, None]
            if a.ndim != 3 or a.shape[2] != 1 or a.dtype != np.uint8:
                raise ImageError("InvalidChannelShape", f"Bayer8 needs a [H, W] or [H, W, 1] uint8 array, got {a.shape} {a.dtype}")
            img = Image.from_numpy(np.ascontiguousarray(a))
        if img.channels != 1 or img.dtype != "uint8":
            raise ImageError("InvalidChannelShape", "Bayer8 wraps a single-channel uint8 image")
        self._image = img

    def as_image(self) -> Image: return self._image
    @property
    def size(self): return self._image.size
    @property
    def is_device(self) -> bool: return self._image.is_device

    def to_hip(self, stream: Stream) -> "Bayer8":
        return Bayer8(self._image.to_hip(stream), self.pattern)

    def cpu(self) -> "Bayer8":
        return Bayer8(self._image.cpu(), self.pattern)


# ---- DeviceVideoFrame (P/cuda/color/video.rs:470-640) -------------------------------------------------------------
VIDEO_FORMATS = {"nv12": Nv12, "nv21": Nv21, "i420": I420, "yv12": Yv12, "yuyv": Yuyv8, "uyvy": Uyvy8, "yvyu": Yvyu8}


class DeviceVideoFrame:
    """A device-resident camera frame tagged with its layout — ``VideoFormat::{Packed422, Planar420}`` by name
    (``nv12`` ... ``yvyu``).  ``from_host`` uploads exactly ``buffer_len`` bytes (longer inputs are truncated, shorter ones
    rejected: video.rs:516-537); ``to_rgb`` / ``convert`` decode into an RGB8 image of the frame's size."""

    def __init__(self, buffer: _VideoBuffer):
        if not buffer.is_device:
            raise ImageError("UnsupportedDevice", "video frame is not device-backed")
        self._buf = buffer

    @staticmethod
    def buffer_len(fmt: str, width: int, height: int) -> int:  # VideoFormat::buffer_len, :485-492
        cls = DeviceVideoFrame._cls(fmt)
        return width * height * cls._num // cls._den

    @staticmethod
    def _cls(fmt: str):
        cls = VIDEO_FORMATS.get(str(fmt).lower())
        if cls is None:
            raise ImageError("InvalidArgument", f"unknown video format {fmt!r} ({', '.join(VIDEO_FORMATS)})")
        return cls

    @classmethod
    def from_host(cls, data, width: int, height: int, fmt: str, stream: Stream) -> "DeviceVideoFrame":
        need = cls.buffer_len(fmt, width, height)
        flat = np.ascontiguousarray(np.asarray(data, np.uint8)).reshape(-1)
        if flat.size < need:
            raise ImageError("InvalidImageSize", f"src holds {flat.size} bytes, a {width}x{height} {fmt} frame needs {need}")
        return cls(cls._cls(fmt)(width, height, flat[:need]).to_hip(stream))

    @classmethod
    def from_device_buffer(cls, buf: DeviceBuffer, width: int, height: int, fmt: str) -> "DeviceVideoFrame":
        """``from_cudaslice``: adopt device memory that already holds the frame (exactly ``buffer_len`` bytes)."""
        frame = cls._cls(fmt)(width, height, None, _device=buf)
        frame.stream = buf.stream
        return cls(frame)

    @property
    def width(self) -> int: return self._buf.width
    @property
    def height(self) -> int: return self._buf.height
    @property
    def format(self) -> str: return self._buf.layout
    @property
    def buffer(self) -> _VideoBuffer: return self._buf

    def to_rgb(self, dst: Optional[Image] = None) -> Image:
        from . import imgproc
        if dst is not None and dst.size != (self.width, self.height):
            raise ImageError("InvalidImageSize", f"destination is {dst.width}x{dst.height}, the frame {self.width}x{self.height}")
        out = imgproc.rgb_from_video(self._buf, dst)
        out.color_space = ColorSpace.RGB
        return out

    convert = to_rgb  # ConvertColor<Rgb8> for DeviceVideoFrame, video.rs:630-638


# ---- ConvertColor (P/color/convert.rs:101-273) -------------------------------------------------------------
# (source space, destination space) -> (imgproc function, dtypes the reference implements for a device pair)
_U8, _F32, _BOTH = ("uint8",), ("float32",), ("uint8", "float32")
_FLT, _ALL = ("float32", "float64"), ("uint8", "float32", "float64")
_CONVERSIONS = {
    (ColorSpace.RGB, ColorSpace.GRAY): ("gray_from_rgb", _ALL), (ColorSpace.GRAY, ColorSpace.RGB): ("rgb_from_gray", _ALL),
    (ColorSpace.RGB, ColorSpace.BGR): ("bgr_from_rgb", _BOTH), (ColorSpace.BGR, ColorSpace.RGB): ("bgr_from_rgb", _BOTH),
    (ColorSpace.RGB, ColorSpace.HSV): ("hsv_from_rgb", _FLT), (ColorSpace.HSV, ColorSpace.RGB): ("rgb_from_hsv", _FLT),
    (ColorSpace.RGB, ColorSpace.HLS): ("hls_from_rgb", _FLT), (ColorSpace.HLS, ColorSpace.RGB): ("rgb_from_hls", _FLT),
    (ColorSpace.RGB, ColorSpace.LINEAR_RGB): ("linear_rgb_from_rgb", _FLT),
    (ColorSpace.LINEAR_RGB, ColorSpace.RGB): ("rgb_from_linear_rgb", _FLT),
    (ColorSpace.RGB, ColorSpace.XYZ): ("xyz_from_rgb", _FLT), (ColorSpace.XYZ, ColorSpace.RGB): ("rgb_from_xyz", _FLT),
    (ColorSpace.RGB, ColorSpace.LAB): ("lab_from_rgb", _FLT), (ColorSpace.LAB, ColorSpace.RGB): ("rgb_from_lab", _FLT),
    (ColorSpace.RGB, ColorSpace.LUV): ("luv_from_rgb", _FLT), (ColorSpace.LUV, ColorSpace.RGB): ("rgb_from_luv", _FLT),
    (ColorSpace.RGB, ColorSpace.YCBCR): ("ycbcr_from_rgb", _ALL), (ColorSpace.YCBCR, ColorSpace.RGB): ("rgb_from_ycbcr", _ALL),
    (ColorSpace.RGB, ColorSpace.YUV): ("yuv_from_rgb", _ALL), (ColorSpace.YUV, ColorSpace.RGB): ("rgb_from_yuv", _ALL),
    (ColorSpace.RGBA, ColorSpace.RGB): ("rgb_from_rgba", _U8), (ColorSpace.BGRA, ColorSpace.RGB): ("rgb_from_bgra", _U8),
    (ColorSpace.RGB, ColorSpace.RGBA): ("rgba_from_rgb", _BOTH), (ColorSpace.RGB, ColorSpace.BGRA): ("bgra_from_rgb", _BOTH),
}


def _space_of(to) -> ColorSpace:
    if isinstance(to, ColorSpace):
        return to
    space = getattr(to, "color_space", None)  # a typed constructor (Gray8, Hsvf32 ...) or a tagged image
    if isinstance(space, ColorSpace):
        return space
    raise ImageError("InvalidChannelShape", f"convert: {to!r} does not name a colour space")


def convert(src, to, dst: Optional[Image] = None, background=None) -> Image:
    """``src.convert(&mut dst)`` (P/color/convert.rs:31-40).  ``src`` is a tagged image (``Rgb8(...)``, possibly moved with
    ``to_hip``) or a camera buffer (``Nv12`` ... decode to RGB8, :245-261); ``to`` a ``ColorSpace`` or a typed
    constructor.  ``background`` is ``convert_with_bg``'s optional RGB triple for RGBA/BGRA sources (:266-274).
    The result carries the destination tag."""
    from . import imgproc
    want = _space_of(to)
    if isinstance(src, Bayer8):  # impl ConvertColor<Rgb8> for Bayer8 (convert.rs:101-106)
        if want is not ColorSpace.RGB:
            raise ImageError("NoDeviceKernel", f"convert: a Bayer mosaic demosaics to RGB8 only (asked for {want.name})")
        out = imgproc.rgb_from_bayer(src, None, dst)
        out.color_space = want
        return out
    if isinstance(src, _VideoBuffer):
        if want is not ColorSpace.RGB:
            raise ImageError("NoDeviceKernel", f"convert: camera buffers decode to RGB8 only (asked for {want.name})")
        out = imgproc.rgb_from_video(src, dst)
        out.color_space = want
        return out
    have = getattr(src, "color_space", None)
    if have is None:
        raise ImageError("InvalidChannelShape", "convert: the source image carries no colour-space tag; build it with a typed "
                                                 "constructor (Rgb8, Bgrf32 ...) or call the imgproc function directly")
    entry = _CONVERSIONS.get((have, want))
    if entry is None or src.dtype not in entry[1]:
        raise ImageError("NoDeviceKernel", f"convert: no {have.name} -> {want.name} conversion for {src.dtype} images")
    fn = getattr(imgproc, entry[0])
    if background is not None:
        if have not in (ColorSpace.RGBA, ColorSpace.BGRA):
            raise ImageError("NoDeviceKernel", "convert: a background applies to RGBA / BGRA sources only")
        out = fn(src, dst, background)
    else:
        out = fn(src, dst)
    out.color_space = want
    return out
