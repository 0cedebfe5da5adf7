"""Colour-space tags and the packed / planar camera buffer types — host mirror of
``crates/kornia-image/src/color_spaces.rs`` (ColorSpace :19-80, typed wrappers :269-620, video buffers
:630-830).

``ColorSpace`` tags what the channels of an ``Image`` mean; the typed constructors (``Rgb8``, ``Grayf32`` ...)
are ``Image`` factories that check dtype and channel count.  ``Nv12`` / ``Nv21`` / ``I420`` / ``Yv12`` and
``Yuyv8`` / ``Uyvy8`` / ``Yvyu8`` carry one byte buffer with a sub-sampled layout and validate its length
and the even-dimension rules on construction; ``to_hip`` uploads the buffer for the device decoders
(``imgproc.rgb_from_*`` accept them directly).
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
        img = Image.from_numpy(np.ascontiguousarray(a))
        img.color_space = space
        return img
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
