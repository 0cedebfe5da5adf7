"""Residency-aware N-d ``Tensor`` (host numpy or HIP device memory).

Mirrors ``kornia_rs.Tensor`` (kornia-py/python/kornia_rs/__init__.pyi:30-68;
kornia-py/src/cuda_ext/mod.rs:352-487) over the memory model of
crates/kornia-tensor/src/{tensor,storage,resource}.rs: location is a run-time property of the
storage (``MemoryDomain`` — resource.rs:19), host access to device memory is an error rather
than an implicit copy (storage.rs:102-110), and a device tensor carries the stream its producer
ran on (cuda.rs:89-108).
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Optional, Sequence, Tuple

import numpy as np

from . import hip
from .hip import DeviceBuffer, Stream

_DTYPES = {
    "uint8": np.uint8, "uint16": np.uint16, "int32": np.int32, "int64": np.int64,
    "float16": np.float16, "float32": np.float32, "float64": np.float64,
}


def _np_dtype(dtype) -> np.dtype:
    if isinstance(dtype, str):
        if dtype not in _DTYPES:
            raise ValueError(f"unsupported dtype {dtype!r}")
        return np.dtype(_DTYPES[dtype])
    return np.dtype(dtype)


class Tensor:
    """Contiguous row-major N-d array, host- or device-resident."""

    def __init__(self, shape: Sequence[int], dtype, *, host: Optional[np.ndarray] = None,
                 device_buf: Optional[DeviceBuffer] = None, device_ptr: int = 0,
                 device: int = 0, stream: Optional[Stream] = None, keepalive: Any = None):
        self._shape = tuple(int(s) for s in shape)
        self._dtype = _np_dtype(dtype)
        self._host = host
        self._buf = device_buf
        self._ptr = device_buf.ptr if device_buf is not None else int(device_ptr)
        self._device = device_buf.stream.device if device_buf is not None else int(device)
        self._stream = device_buf.stream if device_buf is not None else stream
        self._keepalive = keepalive  # foreign memory owner (Backing::Foreign, cuda.rs:139-169)
        if host is not None:
            assert host.flags["C_CONTIGUOUS"] and tuple(host.shape) == self._shape

    # -- constructors -------------------------------------------------------------------------
    @staticmethod
    def zeros(shape: Sequence[int], dtype="float32", stream: Optional[Stream] = None) -> "Tensor":
        """Host zeros when ``stream`` is None, else a zeroed stream-ordered device allocation
        (zeros_cuda, cuda.rs:860)."""
        dt = _np_dtype(dtype)
        if stream is None:
            return Tensor(shape, dt, host=np.zeros(shape, dtype=dt))
        n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        return Tensor(shape, dt, device_buf=DeviceBuffer(n, stream, zeroed=True))

    @staticmethod
    def uninit(shape: Sequence[int], dtype, stream: Stream) -> "Tensor":
        """Device allocation without the memset (uninit_cuda, cuda.rs:891): the producer kernel
        must overwrite every element."""
        dt = _np_dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        return Tensor(shape, dt, device_buf=DeviceBuffer(n, stream, zeroed=False))

    @staticmethod
    def from_numpy(a: np.ndarray) -> "Tensor":
        a = np.ascontiguousarray(a)
        return Tensor(a.shape, a.dtype, host=a)

    # -- properties ---------------------------------------------------------------------------
    @property
    def shape(self) -> Tuple[int, ...]:
        return self._shape

    @property
    def dtype(self) -> str:
        return self._dtype.name

    @property
    def is_device(self) -> bool:
        return self._host is None

    @property
    def device(self) -> str:
        # torch-ROCm names HIP devices "cuda:N"; keep that spelling so `torch.device(t.device)`
        # and reference user code keep working.
        return "cpu" if self._host is not None else f"cuda:{self._device}"

    @property
    def device_id(self) -> int:
        return self._device

    @property
    def stream(self) -> Optional[Stream]:
        return self._stream

    @property
    def nbytes(self) -> int:
        return int(np.prod(self._shape, dtype=np.int64)) * self._dtype.itemsize

    @property
    def data_ptr(self) -> int:
        if self._host is not None:
            return int(self._host.ctypes.data)
        return self._ptr

    @property
    def __cuda_array_interface__(self) -> dict:
        if self._host is not None:
            raise AttributeError("__cuda_array_interface__ is device-only; this Tensor is on the host")
        s = self._stream.cuda_stream_ptr if self._stream is not None else 0
        return {
            "shape": self._shape,
            "typestr": self._dtype.str,
            "data": (self._ptr, False),
            "version": 3,
            "strides": None,
            "stream": s if s != 0 else 1,  # CAI v3: 1 = legacy default stream, 0 is disallowed
        }

    # -- transfers ----------------------------------------------------------------------------
    def numpy(self) -> np.ndarray:
        """Host tensor: a no-copy view.  Device tensor: a D2H copy on its stream + sync
        (to_host, cuda.rs:1258-1300).  f16 tensors are widened to f32 like the reference."""
        if self._host is not None:
            out = self._host
        else:
            out = self.numpy_raw()
        return out.astype(np.float32) if self._dtype == np.float16 else out

    def numpy_raw(self) -> np.ndarray:
        """Like numpy() but never widens (binary16 stays binary16)."""
        if self._host is not None:
            return self._host
        out = np.empty(self._shape, dtype=self._dtype)
        hip.d2h(out, self._ptr, self._stream if self._stream is not None else Stream.default(self._device))
        return out

    def to_hip(self, stream: Optional[Stream] = None) -> "Tensor":
        if self._host is None:
            return self
        stream = stream if stream is not None else Stream.default(hip.current_device())
        return Tensor(self._shape, self._dtype, device_buf=DeviceBuffer.from_numpy(self._host, stream))

    def cpu(self) -> "Tensor":
        if self._host is not None:
            return self
        return Tensor(self._shape, self._dtype, host=self.numpy_raw())

    # -- DLPack (T/dlpack.rs:72-290, PY/cuda_ext/mod.rs:196-216) ---------------------------------
    def __dlpack_device__(self) -> Tuple[int, int]:
        from . import dlpack
        return (dlpack.kDLCPU, 0) if self._host is not None else (dlpack.kDLROCM, self._device)

    def __dlpack__(self, *, stream: Any = None, max_version: Any = None, dl_device: Any = None,
                   copy: Any = None) -> Any:
        """Zero-copy capsule; this tensor is the keepalive.  For a device tensor the consumer's
        ``stream`` (an integer handle, array-API convention) is fenced behind the producing stream
        so the consumer never reads before our kernels finish."""
        from . import dlpack
        if copy:
            raise BufferError("DLPack export never copies")
        if self._host is not None:
            return dlpack.export(self, int(self._host.ctypes.data), self._shape, self._dtype, dlpack.kDLCPU, 0)
        if stream is not None and stream != -1 and self._stream is not None:
            consumer = 0 if stream in (0, 1, 2) else int(stream)  # 0/1/2 = default-stream aliases
            hip.check(hip.lib.kh_stream_fence(self._stream.cuda_stream_ptr, consumer))
        return dlpack.export(self, self._ptr, self._shape, self._dtype, dlpack.kDLROCM, self._device)

    @staticmethod
    def from_dlpack(obj: Any, stream: Optional[Stream] = None) -> "Tensor":
        """Zero-copy alias of the producer's buffer, residency inferred from the DLPack device
        (T/dlpack.rs:267-290: any non-host type is Device).  The producer is kept alive for this
        tensor's lifetime (``Backing::Foreign`` with a keepalive, T/cuda.rs:139-169); a device
        import carries ``stream`` (default: the device's default stream) like
        ``from_foreign_cudaslice`` does (T/cuda.rs:1022-1029)."""
        from . import dlpack
        ptr, shape, dtype, dev_type, dev_id, keep = dlpack.from_object(
            obj, stream.cuda_stream_ptr if stream is not None else None)
        if dev_type in dlpack._HOST_TYPES:
            n = int(np.prod(shape, dtype=np.int64))
            buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr) if n else b""
            host = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
            return Tensor(shape, dtype, host=host, keepalive=keep)
        st = stream if stream is not None else Stream.default(dev_id)
        return Tensor(shape, dtype, device_ptr=ptr, device=dev_id, stream=st, keepalive=keep)

    def __repr__(self) -> str:
        return f"Tensor(shape={self._shape}, dtype={self.dtype}, device={self.device})"
