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

from . import allocator as _alloc
from . import hip
from .hip import DeviceBuffer, ManagedBuffer, PinnedBuffer, Stream


class MemoryDomain:
    """Where a buffer can be legally dereferenced (``MemoryDomain``, T/resource.rs:19-60): ``HOST``, ``DEVICE`` or
    ``UNIFIED`` (managed memory: host AND device accessible)."""

    HOST, DEVICE, UNIFIED = "host", "device", "unified"

    @staticmethod
    def is_host_accessible(domain: str) -> bool:
        return domain in (MemoryDomain.HOST, MemoryDomain.UNIFIED)

    @staticmethod
    def is_device_accessible(domain: str) -> bool:
        return domain in (MemoryDomain.DEVICE, MemoryDomain.UNIFIED)

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
                 device: int = 0, stream: Optional[Stream] = None, keepalive: Any = None,
                 managed_buf: Optional[ManagedBuffer] = None, unified: bool = False, pinned: bool = False,
                 alloc: Optional["_alloc.TensorAllocator"] = None):
        # the allocator handle this tensor's memory came from (TensorStorage.alloc, T/storage.rs:53-70); memory the tensor did
        # not allocate (from_numpy of a caller's array, DLPack / array-interface imports) carries the handle that cannot allocate
        self._alloc = alloc if alloc is not None else _alloc.foreign_alloc()
        self._shape = tuple(int(s) for s in shape)
        self._dtype = _np_dtype(dtype)
        self._buf = device_buf if device_buf is not None else managed_buf
        self._ptr = self._buf.ptr if self._buf is not None else int(device_ptr)
        self._device = self._buf.stream.device if self._buf is not None else int(device)
        self._stream = self._buf.stream if self._buf is not None else stream
        self._keepalive = keepalive  # foreign memory owner (Backing::Foreign, cuda.rs:139-169)
        self._pinned = bool(pinned)  # host domain, page-locked (zeros_pinned): exported as kDLROCMHost
        if managed_buf is not None or unified:
            # Unified: the SAME bytes seen from the host (a numpy view) and from kernels (the pointer).
            self._domain = MemoryDomain.UNIFIED
            n = int(np.prod(self._shape, dtype=np.int64))
            if host is None:
                raw = (C.c_uint8 * max(n * self._dtype.itemsize, 1)).from_address(self._ptr)
                host = np.frombuffer(raw, dtype=self._dtype, count=n).reshape(self._shape)
            self._host = None
            self._uview = host
        else:
            self._domain = MemoryDomain.HOST if host is not None else MemoryDomain.DEVICE
            self._host = host
            self._uview = None
        if host is not None:
            assert host.flags["C_CONTIGUOUS"] and tuple(host.shape) == self._shape

    # -- constructors -------------------------------------------------------------------------
    @staticmethod
    def zeros(shape: Sequence[int], dtype="float32", stream: Optional[Stream] = None) -> "Tensor":
        """Host zeros when ``stream`` is None, else a zeroed stream-ordered device allocation
        (zeros_cuda, cuda.rs:860)."""
        return Tensor.zeros_in(shape, dtype, _alloc.host_alloc() if stream is None else _alloc.HipAllocator(stream, zeroed=True))

    @staticmethod
    def zeros_in(shape: Sequence[int], dtype, allocator: "_alloc.TensorAllocator") -> "Tensor":
        """``Tensor::zeros(shape, alloc)`` (T/tensor.rs:300-330): the allocator decides where the tensor lives — ``CpuAllocator``
        / ``PinnedAllocator`` give a host tensor, ``HipAllocator`` a device tensor on its stream, ``HipUnifiedAllocator`` a
        unified one.  The tensor keeps the handle (``Tensor.alloc``)."""
        dt = _np_dtype(dtype)
        shape = tuple(int(s) for s in shape)
        n = int(np.prod(shape, dtype=np.int64))
        res = allocator.allocate(_alloc.Layout.array(dt, n))
        if res.len_bytes() != n * dt.itemsize:
            raise _alloc.TensorAllocatorError("NullPointer", f"allocator returned {res.len_bytes()} bytes for a {n * dt.itemsize}-byte tensor")
        back = res.as_any()
        if res.domain == MemoryDomain.DEVICE:
            return Tensor(shape, dt, device_buf=back, alloc=allocator)
        if res.domain == MemoryDomain.UNIFIED:
            return Tensor(shape, dt, managed_buf=back, alloc=allocator)
        if isinstance(back, PinnedBuffer):
            host = back.view().view(dt)[:n].reshape(shape) if n else np.zeros(shape, dt)
            return Tensor(shape, dt, host=host, keepalive=back, pinned=True, alloc=allocator)
        host = back.view(dt)[:n].reshape(shape) if n else np.zeros(shape, dt)
        return Tensor(shape, dt, host=host, keepalive=res, alloc=allocator)

    @staticmethod
    def uninit(shape: Sequence[int], dtype, stream: Stream) -> "Tensor":
        """Device allocation without the memset (uninit_cuda, cuda.rs:891): the producer kernel
        must overwrite every element."""
        return Tensor.zeros_in(shape, dtype, _alloc.HipAllocator(stream, zeroed=False))

    @staticmethod
    def zeros_unified(shape: Sequence[int], dtype, stream: Stream) -> "Tensor":
        """Zero-filled managed memory carrying ``stream`` (``zeros_cuda_unified``, T/cuda.rs:513-560): host slices AND
        device kernels work on it without a copy; dispatch treats it as device-resident (P/cuda/dispatch.rs:90-96)."""
        return Tensor.zeros_in(shape, dtype, _alloc.HipUnifiedAllocator(stream))

    zeros_hip_unified = zeros_unified
    zeros_cuda_unified = zeros_unified  # reference spelling

    @staticmethod
    def zeros_pinned(shape: Sequence[int], dtype="uint8") -> "Tensor":
        """A HOST tensor in page-locked memory (``zeros_pinned``, T/cuda.rs:382-410): an ordinary host tensor for every
        host path, but copies against it are direct, stream-ordered DMA.  Allocate once and reuse."""
        return Tensor.zeros_in(shape, dtype, _alloc.PinnedAllocator())

    @staticmethod
    def from_numpy(a: np.ndarray) -> "Tensor":
        a = np.ascontiguousarray(a)
        return Tensor(a.shape, a.dtype, host=a, alloc=_alloc.host_alloc())  # from_shape_vec(shape, data, CpuAllocator)

    # -- properties ---------------------------------------------------------------------------
    @property
    def shape(self) -> Tuple[int, ...]:
        return self._shape

    @property
    def dtype(self) -> str:
        return self._dtype.name

    @property
    def alloc(self) -> "_alloc.TensorAllocator":
        """The allocator handle the storage was created with (``TensorStorage::alloc``); ``ForeignAllocator`` for wrapped memory."""
        return self._alloc

    @property
    def domain(self) -> str:
        """``MemoryDomain.HOST`` / ``DEVICE`` / ``UNIFIED`` (TensorStorage::domain, T/storage.rs:86-100)."""
        return self._domain

    @property
    def is_device(self) -> bool:
        """Device- OR unified-resident: what residency dispatch asks (``is_device``, P/cuda/dispatch.rs:90-96)."""
        return self._domain != MemoryDomain.HOST

    @property
    def is_unified(self) -> bool:
        return self._domain == MemoryDomain.UNIFIED

    @property
    def is_host_accessible(self) -> bool:
        return MemoryDomain.is_host_accessible(self._domain)

    @property
    def is_pinned(self) -> bool:
        return self._pinned

    @property
    def device(self) -> str:
        # torch-ROCm names HIP devices "cuda:N"; keep that spelling so `torch.device(t.device)`
        # and reference user code keep working.
        return "cpu" if self._domain == MemoryDomain.HOST else f"cuda:{self._device}"

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
        """Like numpy() but never widens (binary16 stays binary16).  Unified memory: the carried stream is drained, then
        the managed bytes themselves are returned as a no-copy view (``as_slice`` on ``MemoryDomain::Unified``)."""
        if self._host is not None:
            return self._host
        if self._uview is not None:
            if self._stream is not None:
                self._stream.synchronize()
            return self._uview
        out = np.empty(self._shape, dtype=self._dtype)
        hip.d2h(out, self._ptr, self._stream if self._stream is not None else Stream.default(self._device))
        return out

    def to_hip_unified(self, stream: Stream) -> "Tensor":
        """Copy this HOST tensor into a new managed allocation carrying ``stream`` (``to_cuda_unified``, T/cuda.rs:1302-1340)."""
        if self._host is None:
            raise ValueError("to_hip_unified: the tensor is not host-resident")
        out = Tensor.zeros_unified(self._shape, self._dtype, stream)
        if self.nbytes:
            out._uview[...] = self._host  # managed memory is host-writable; first device touch migrates the pages
        return out

    to_cuda_unified = to_hip_unified

    def to_hip(self, stream: Optional[Stream] = None) -> "Tensor":
        if self._host is None:
            return self
        stream = stream if stream is not None else Stream.default(hip.current_device())
        return Tensor(self._shape, self._dtype, device_buf=DeviceBuffer.from_numpy(self._host, stream), alloc=_alloc.HipAllocator(stream, zeroed=False))

    def cpu(self) -> "Tensor":
        if self._host is not None:
            return self
        raw = self.numpy_raw()
        return Tensor(self._shape, self._dtype, host=raw.copy() if self._uview is not None else raw, alloc=_alloc.host_alloc())

    # -- DLPack (T/dlpack.rs:72-290, PY/cuda_ext/mod.rs:196-216) ---------------------------------
    def __dlpack_device__(self) -> Tuple[int, int]:
        from . import dlpack
        if self._host is not None:  # Host -> kDLCPU, page-locked host -> kDLROCMHost (T/dlpack.rs:76-84)
            return (dlpack.kDLROCMHost, 0) if self._pinned else (dlpack.kDLCPU, 0)
        return (dlpack.kDLCUDAManaged, self._device) if self._uview is not None else (dlpack.kDLROCM, self._device)

    def __dlpack__(self, *, stream: Any = None, max_version: Any = None, dl_device: Any = None,
                   copy: Any = None) -> Any:
        """Zero-copy capsule; this tensor is the keepalive.  For a device tensor the consumer's
        ``stream`` (an integer handle, array-API convention) is fenced behind the producing stream
        so the consumer never reads before our kernels finish."""
        from . import dlpack
        if copy:
            raise BufferError("DLPack export never copies")
        if self._host is not None:
            return dlpack.export(self, int(self._host.ctypes.data), self._shape, self._dtype,
                                 dlpack.kDLROCMHost if self._pinned else dlpack.kDLCPU, 0)
        hip._ffi.assert_single_runtime()  # device memory is about to cross to another library
        dev_code = dlpack.kDLCUDAManaged if self._uview is not None else dlpack.kDLROCM
        if dl_device is not None and self._uview is not None:
            # Managed memory is legal on both sides: a consumer that cannot take kDLCUDAManaged (torch-ROCm) may ask for
            # the device view, (kDLROCM, id), or the host view, (kDLCPU, 0) — same pointer, no copy.
            want = (int(dl_device[0]), int(dl_device[1]))
            if want == (dlpack.kDLCPU, 0):
                if self._stream is not None:
                    self._stream.synchronize()
                return dlpack.export(self, self._ptr, self._shape, self._dtype, dlpack.kDLCPU, 0)
            if want not in ((dlpack.kDLROCM, self._device), (dlpack.kDLCUDAManaged, self._device)):
                raise BufferError(f"cannot export managed memory of device {self._device} as DLPack device {want}")
            dev_code = want[0]
        elif dl_device is not None and (int(dl_device[0]), int(dl_device[1])) != (dev_code, self._device):
            raise BufferError(f"DLPack export never copies: the tensor lives on {(dev_code, self._device)}, not {tuple(dl_device)}")
        if stream is not None and stream != -1 and self._stream is not None:
            consumer = 0 if stream in (0, 1, 2) else int(stream)  # 0/1/2 = default-stream aliases
            hip.check(hip.lib.kh_stream_fence(self._stream.cuda_stream_ptr, consumer))
        # DLPack has no ROCm-specific managed code: kDLCUDAManaged (13) is THE managed type, as in the reference
        return dlpack.export(self, self._ptr, self._shape, self._dtype, dev_code, self._device)

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
            return Tensor(shape, dtype, host=host, keepalive=keep, pinned=dev_type != dlpack.kDLCPU)
        hip._ffi.assert_single_runtime()  # a foreign exporter of device memory: both sides must share one HIP runtime
        st = stream if stream is not None else Stream.default(dev_id)
        return Tensor(shape, dtype, device_ptr=ptr, device=dev_id, stream=st, keepalive=keep,
                      unified=dev_type == dlpack.kDLCUDAManaged)

    def __repr__(self) -> str:
        return f"Tensor(shape={self._shape}, dtype={self.dtype}, device={self.device})"
