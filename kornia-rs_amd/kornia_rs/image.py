"""``Image`` — typed HWC image over a residency-aware ``Tensor``.

Mirrors ``kornia_image::Image<T, C>`` (crates/kornia-image/src/image.rs:138: a newtype over
``Tensor3<T>`` with shape ``[H, W, C]``), its device helpers (crates/kornia-image/src/cuda.rs:53-221
``to_cuda`` / ``zeros_cuda`` / ``to_host_*``), ``ImageError`` (crates/kornia-image/src/error.rs)
and the Python class ``kornia_rs.image.Image`` (kornia-py/python/kornia_rs/image.pyi:49-300:
``from_numpy``, ``zeros(w, h, c, dtype, stream=)``, ``from_dlpack``, ``.numpy()``, ``.device``,
``.cpu()``, ``.to_cuda()``, ``__cuda_array_interface__``, ``__dlpack__``).

``to_hip`` is the native spelling; ``to_cuda`` is the same method under the reference's name so
user code written against kornia_rs keeps running — there is no CUDA code path behind it.
"""
from __future__ import annotations

from typing import Any, Optional, Tuple

import numpy as np

from .hip import Stream
from .tensor import Tensor


class ImageError(ValueError):
    """``kind`` names the reference variant (crates/kornia-image/src/error.rs): MixedResidency,
    DeviceMismatch, UnsupportedDevice, InvalidImageSize, InvalidChannelShape, Hip,
    CannotComputeDeterminant, PixelIndexOutOfBounds, InvalidSigmaValue, InvalidKernelLength,
    ImageDataNotInitialized, NoDeviceKernel, HostPathUnavailable."""

    def __init__(self, kind: str, message: str):
        super().__init__(message)
        self.kind = kind


class Image:
    def __init__(self, tensor: Tensor, color_space=None):
        if len(tensor.shape) != 3:
            raise ImageError("InvalidChannelShape", f"an image is [H, W, C], got shape {tensor.shape}")
        self._t = tensor
        # what the channels mean (color_spaces.ColorSpace) when the image came from a typed constructor
        # (Rgb8, Hsvf32 ... — the newtypes of I/color_spaces.rs:269-620); None = untyped.  Travels with transfers.
        self.color_space = color_space

    # -- constructors -------------------------------------------------------------------------
    @staticmethod
    def from_numpy(data: np.ndarray, copy: bool = False) -> "Image":
        a = np.asarray(data)
        if a.ndim == 2:
            a = a[:, :, None]
        if a.ndim != 3:
            raise ImageError("InvalidChannelShape", f"expected (H, W) or (H, W, C), got {a.shape}")
        a = np.array(a, copy=True) if copy else np.ascontiguousarray(a)
        return Image(Tensor.from_numpy(a))

    @staticmethod
    def zeros(width: int, height: int, channels: int, dtype: str = "uint8", stream: Optional[Stream] = None) -> "Image":
        return Image(Tensor.zeros((height, width, channels), dtype, stream=stream))

    @staticmethod
    def uninit(width: int, height: int, channels: int, dtype: str, stream: Stream) -> "Image":
        """Device image without the memset (uninit_cuda, I/cuda.rs:110-121): every device op fully
        overwrites its destination."""
        return Image(Tensor.uninit((height, width, channels), dtype, stream))

    @staticmethod
    def zeros_hip_unified(width: int, height: int, channels: int, dtype: str, stream: Stream) -> "Image":
        """Zero-filled MANAGED memory carrying ``stream`` (``zeros_cuda_unified``, I/cuda.rs:144-160): host slices and
        device kernels both work on it; residency dispatch routes it to the device kernels without an upload."""
        return Image(Tensor.zeros_unified((height, width, channels), dtype, stream))

    zeros_cuda_unified = zeros_hip_unified  # reference spelling

    @staticmethod
    def zeros_pinned(width: int, height: int, channels: int, dtype: str = "uint8") -> "Image":
        """A HOST image in page-locked memory (``zeros_pinned``, I/cuda.rs:122-142): uploads from it are direct DMA."""
        return Image(Tensor.zeros_pinned((height, width, channels), dtype))

    @staticmethod
    def from_dlpack(obj: Any, stream: Optional[Stream] = None) -> "Image":
        t = Tensor.from_dlpack(obj, stream)
        if len(t.shape) == 2:
            t = Tensor(t.shape + (1,), t._dtype, host=None if t.is_device else t._host.reshape(t.shape + (1,)),
                       device_ptr=t.data_ptr if t.is_device else 0, device=t.device_id, stream=t.stream, keepalive=t,
                       unified=t.is_unified, pinned=t.is_pinned)
        return Image(t)

    # -- geometry / type ------------------------------------------------------------------------
    @property
    def height(self) -> int: return self._t.shape[0]
    @property
    def width(self) -> int: return self._t.shape[1]
    @property
    def channels(self) -> int: return self._t.shape[2]
    @property
    def size(self) -> Tuple[int, int]: return (self.width, self.height)
    @property
    def shape(self) -> Tuple[int, int, int]: return self._t.shape
    @property
    def dtype(self) -> str: return self._t.dtype
    @property
    def tensor(self) -> Tensor: return self._t

    # -- residency ------------------------------------------------------------------------------
    @property
    def is_device(self) -> bool: return self._t.is_device
    @property
    def domain(self) -> str: return self._t.domain
    @property
    def is_unified(self) -> bool: return self._t.is_unified
    @property
    def device(self) -> str: return self._t.device
    @property
    def device_id(self) -> int: return self._t.device_id
    @property
    def stream(self) -> Optional[Stream]: return self._t.stream
    @property
    def data_ptr(self) -> int: return self._t.data_ptr
    @property
    def nbytes(self) -> int: return self._t.nbytes

    def to_hip(self, stream: Optional[Stream] = None) -> "Image":
        """H2D copy onto ``stream``'s device (to_cuda, I/cuda.rs:53-70); a device image is returned as is."""
        return self if self.is_device else Image(self._t.to_hip(stream), self.color_space)

    to_cuda = to_hip  # reference spelling (kornia_rs/image.pyi:199)

    def to_hip_unified(self, stream: Stream) -> "Image":
        """Copy this HOST image into a new managed-memory image carrying ``stream`` (``to_cuda_unified``, I/cuda.rs:62-75)."""
        if self.is_device:
            raise ImageError("UnsupportedDevice", "to_hip_unified: the image is not host-resident")
        return Image(self._t.to_hip_unified(stream), self.color_space)

    to_cuda_unified = to_hip_unified

    def cpu(self, stream: Optional[Stream] = None) -> "Image":
        if self.is_device:
            return Image(self._t.cpu(), self.color_space)
        return Image(Tensor.from_numpy(self._t.numpy_raw().copy()), self.color_space)

    def numpy(self) -> np.ndarray:
        """Host: zero-copy view.  Device: D2H copy, returned read-only (image.pyi:179-184)."""
        a = self._t.numpy_raw()
        if self.is_device and not self.is_unified:  # a managed image IS its host view: writable, like as_slice_mut
            a.flags.writeable = False
        return a

    def as_slice(self) -> np.ndarray:
        """Host access only — like ``TensorStorage::as_slice`` it refuses device memory (T/storage.rs:102-110)."""
        if not self._t.is_host_accessible:
            raise ImageError("UnsupportedDevice", "host access to device-resident image data; call .cpu() first")
        return self._t.numpy_raw().reshape(-1)

    def resize_normalize_to_tensor(self, width: int, height: int, mean, std) -> Tensor:
        """Fused bilinear resize + normalise + HWC->CHW (image.pyi:216-228); device images only here."""
        from . import imgproc
        return imgproc.resize_normalize_to_tensor(self, width, height, mean, std)

    @property
    def __cuda_array_interface__(self) -> dict:
        return self._t.__cuda_array_interface__

    def __dlpack__(self, **kw) -> Any:
        return self._t.__dlpack__(**kw)

    def __dlpack_device__(self) -> Tuple[int, int]:
        return self._t.__dlpack_device__()

    def __array__(self, dtype=None, copy=None) -> np.ndarray:
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self) -> str:
        return f"Image(width={self.width}, height={self.height}, channels={self.channels}, dtype={self.dtype}, device={self.device})"
