"""Allocator abstraction of the host layer: ``TensorAllocator`` / ``CpuAllocator`` / ``host_alloc()`` and the HIP allocators that
replace ``CudaAllocator`` / ``PinnedAllocator`` / ``CudaUnifiedAllocator``.

Mirrors crates/kornia-tensor/src/allocator.rs:73-144 (the trait, ``CpuAllocator``, ``AllocHandle``, the process-global
``host_alloc()``) and the device allocators of crates/kornia-tensor/src/cuda.rs:214-262 (``CudaAllocator``: stream-ordered pool,
zeroed or uninitialised), :355-380 (``PinnedAllocator``), :440-511 (``CudaUnifiedAllocator``).  An allocator turns a ``Layout``
(size + alignment, Rust's ``std::alloc::Layout``) into an owning ``MemoryResource`` (T/resource.rs:73-101: ``as_ptr``,
``len_bytes``, ``domain``, ``is_readonly``); every ``Tensor`` constructor goes through one, and a tensor remembers the handle it
was allocated with (``Tensor.alloc``, the ``alloc: AllocHandle`` field of ``TensorStorage``, T/storage.rs:53-70).

Nothing here computes: the HIP allocators call the C ABI (``kh_malloc_async`` / ``kh_host_alloc`` / ``kh_malloc_managed``).
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Optional

import numpy as np

from .hip import DeviceBuffer, ManagedBuffer, PinnedBuffer, Stream


class TensorAllocatorError(Exception):
    """``TensorAllocatorError`` (T/allocator.rs:20-40).  ``kind`` is the reference variant name: ``LayoutError`` (bad size /
    alignment), ``NullPointer`` (the allocator returned nothing), ``CannotAllocateForeign`` (a wrapper of foreign memory cannot
    allocate)."""

    def __init__(self, kind: str, msg: str):
        super().__init__(f"{kind}: {msg}")
        self.kind = kind


class Layout:
    """Size and alignment of an allocation (``std::alloc::Layout``): the alignment is a power of two, the size fits isize."""

    __slots__ = ("size", "align")

    def __init__(self, size: int, align: int = 1):
        size, align = int(size), int(align)
        if size < 0 or align <= 0 or (align & (align - 1)) or size > (1 << 63) - align:
            raise TensorAllocatorError("LayoutError", f"invalid layout (size {size}, align {align})")
        self.size, self.align = size, align

    @staticmethod
    def array(dtype, count: int) -> "Layout":
        """``Layout::array::<T>(n)``."""
        dt = np.dtype(dtype)
        return Layout(dt.itemsize * int(count), max(dt.alignment, 1))

    def __repr__(self) -> str:
        return f"Layout(size={self.size}, align={self.align})"


class MemoryResource:
    """An owning handle to a block of memory (``MemoryResource``, T/resource.rs:73-101).  Dropping the last reference frees it."""

    domain: str = "host"

    def as_ptr(self) -> int:
        raise NotImplementedError

    def len_bytes(self) -> int:
        raise NotImplementedError

    def is_readonly(self) -> bool:
        return False

    def as_any(self) -> Any:
        """The backing object (``as_any`` downcast): a numpy array, ``PinnedBuffer``, ``DeviceBuffer`` or ``ManagedBuffer``."""
        raise NotImplementedError


class HostResource(MemoryResource):
    """Zeroed, aligned host memory (``HostResource::from_layout``, T/resource.rs:103-170).  A zero-size layout is legal and
    allocates nothing."""

    domain = "host"

    def __init__(self, layout: Layout):
        self._n = layout.size
        raw = np.zeros(layout.size + layout.align, np.uint8)  # over-allocate, then slide to the alignment
        off = (-raw.ctypes.data) % layout.align
        self._a = raw[off:off + layout.size]
        if layout.size and self._a.ctypes.data % layout.align:
            raise TensorAllocatorError("NullPointer", "host allocation could not be aligned")

    def as_ptr(self) -> int:
        return int(self._a.ctypes.data)

    def len_bytes(self) -> int:
        return self._n

    def as_any(self) -> np.ndarray:
        return self._a


class PinnedResource(MemoryResource):
    """Page-locked host memory (host domain; ``PinnedAllocator``, T/cuda.rs:355-380)."""

    domain = "host"

    def __init__(self, layout: Layout):
        self._buf = PinnedBuffer(layout.size)
        if layout.size and not self._buf.ptr:
            raise TensorAllocatorError("NullPointer", "kh_host_alloc returned null")
        if layout.size:
            C.memset(self._buf.ptr, 0, layout.size)

    def as_ptr(self) -> int:
        return int(self._buf.ptr or 0)

    def len_bytes(self) -> int:
        return self._buf.nbytes

    def as_any(self) -> PinnedBuffer:
        return self._buf


class DeviceResource(MemoryResource):
    """A stream-ordered device allocation carrying its stream (``CudaResource`` with ``Backing::Device``, T/cuda.rs:89-169)."""

    domain = "device"

    def __init__(self, layout: Layout, stream: Stream, zeroed: bool):
        self._buf = DeviceBuffer(layout.size, stream, zeroed=zeroed)
        if layout.size and not self._buf.ptr:
            raise TensorAllocatorError("NullPointer", "kh_malloc_async returned null")
        if layout.size and self._buf.ptr % layout.align:
            raise TensorAllocatorError("LayoutError", f"the device pool returned a pointer not aligned to {layout.align}")

    def as_ptr(self) -> int:
        return int(self._buf.ptr)

    def len_bytes(self) -> int:
        return self._buf.nbytes

    def as_any(self) -> DeviceBuffer:
        return self._buf

    @property
    def stream(self) -> Stream:
        return self._buf.stream


class ManagedResource(MemoryResource):
    """Managed (unified) memory: host AND device accessible (``Backing::Managed``, T/cuda.rs:440-511)."""

    domain = "unified"

    def __init__(self, layout: Layout, stream: Stream):
        self._buf = ManagedBuffer(layout.size, stream)
        if layout.size:
            C.memset(self._buf.ptr, 0, layout.size)

    def as_ptr(self) -> int:
        return int(self._buf.ptr)

    def len_bytes(self) -> int:
        return self._buf.nbytes

    def as_any(self) -> ManagedBuffer:
        return self._buf

    @property
    def stream(self) -> Stream:
        return self._buf.stream


class TensorAllocator:
    """``trait TensorAllocator`` (T/allocator.rs:73-90): ``allocate(layout) -> MemoryResource``, zero-filled unless the allocator
    says otherwise.  Stateless or internally synchronised: one handle may serve every thread (``Send + Sync``)."""

    def allocate(self, layout: Layout) -> MemoryResource:
        raise NotImplementedError


class CpuAllocator(TensorAllocator):
    """Zeroed host memory from the process heap (T/allocator.rs:107-130)."""

    def allocate(self, layout: Layout) -> MemoryResource:
        return HostResource(layout)

    def __repr__(self) -> str:
        return "CpuAllocator"


class PinnedAllocator(TensorAllocator):
    """Zeroed page-locked host memory (``PinnedAllocator``, T/cuda.rs:355-380): uploads from it are true stream-ordered DMA."""

    def allocate(self, layout: Layout) -> MemoryResource:
        return PinnedResource(layout)

    def __repr__(self) -> str:
        return "PinnedAllocator"


class HipAllocator(TensorAllocator):
    """The HIP allocator that replaces ``CudaAllocator`` (T/cuda.rs:214-262): stream-ordered pool memory on ``stream``'s device,
    zero-filled on that stream (``zeroed=False`` = ``uninit_cuda``: the producer must overwrite every byte)."""

    def __init__(self, stream: Stream, zeroed: bool = True):
        self.stream, self.zeroed = stream, bool(zeroed)

    def allocate(self, layout: Layout) -> MemoryResource:
        return DeviceResource(layout, self.stream, self.zeroed)

    def __repr__(self) -> str:
        return f"HipAllocator(device={self.stream.device}, zeroed={self.zeroed})"


class HipUnifiedAllocator(TensorAllocator):
    """Managed memory carrying ``stream`` (``CudaUnifiedAllocator``, T/cuda.rs:440-511)."""

    def __init__(self, stream: Stream):
        self.stream = stream

    def allocate(self, layout: Layout) -> MemoryResource:
        return ManagedResource(layout, self.stream)

    def __repr__(self) -> str:
        return f"HipUnifiedAllocator(device={self.stream.device})"


class ForeignAllocator(TensorAllocator):
    """The handle of a tensor that wraps memory it did not allocate (DLPack / array-interface imports): it cannot allocate
    (``CannotAllocateForeign``, T/allocator.rs:32-36)."""

    def allocate(self, layout: Layout) -> MemoryResource:
        raise TensorAllocatorError("CannotAllocateForeign", "this tensor wraps foreign memory; its allocator handle cannot allocate")

    def __repr__(self) -> str:
        return "ForeignAllocator"


_HOST_ALLOC: Optional[CpuAllocator] = None
_FOREIGN_ALLOC = ForeignAllocator()


def host_alloc() -> CpuAllocator:
    """The process-global host allocator handle (``host_alloc()``, T/allocator.rs:136-146): one shared stateless instance."""
    global _HOST_ALLOC
    if _HOST_ALLOC is None:
        _HOST_ALLOC = CpuAllocator()
    return _HOST_ALLOC


def foreign_alloc() -> ForeignAllocator:
    return _FOREIGN_ALLOC
