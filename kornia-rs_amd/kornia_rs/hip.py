"""Device runtime objects over the C ABI: ``Stream``, ``Event``, raw device/pinned buffers.

Mirrors ``kornia_rs.cuda`` of the reference (kornia-py/python/kornia_rs/cuda.pyi:22-75 —
``Stream.new/default/from_handle/from_cuda_stream/synchronize/cuda_stream_ptr/__cuda_stream__``,
``is_available``, ``mem_get_info``) with the HIP allocator of
crates/kornia-tensor/src/cuda.rs underneath.  The module is named for HIP; there is no CUDA
code path anywhere — ``__cuda_stream__`` / ``__cuda_array_interface__`` are kept only because
they are the protocol names torch-ROCm and cupy-ROCm speak.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Any, Optional, Tuple

import numpy as np

from . import _ffi
from ._ffi import check, lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def device_count() -> int:
    n = C.c_int32(0)
    rc = lib.kh_device_count(C.byref(n))
    return int(n.value) if rc == _ffi.KH_OK else 0


def is_available() -> bool:
    """True if a HIP device is usable in this process."""
    return device_count() > 0


def set_device(device: int) -> None:
    check(lib.kh_set_device(device))


def current_device() -> int:
    d = C.c_int32(0)
    check(lib.kh_get_device(C.byref(d)))
    return int(d.value)


class _device_guard:
    """``with _device_guard(d):`` — device ``d`` current inside, the previous one restored after.  Needed wherever a handle's meaning
    depends on the current device: the NULL stream (``Stream.default(d)`` has handle 0) and event / buffer creation."""

    __slots__ = ("device", "prev")

    def __init__(self, device: int):
        self.device, self.prev = int(device), None

    def __enter__(self):
        self.prev = current_device()
        if self.prev != self.device:
            set_device(self.device)
        return self

    def __exit__(self, *exc):
        if self.prev != self.device:
            set_device(self.prev)
        return False


def _operand_device(obj) -> Optional[int]:
    """The device an operand lives on, if it says so: a device Image / Tensor (``is_device`` + ``device_id``), a DeviceBuffer /
    anything carrying a ``stream``, a Stream."""
    if isinstance(obj, Stream):
        return obj.device
    if isinstance(obj, (list, tuple)):     # a batch of images / frames (imgproc.*_batch, run_raw_batch): its first device operand
        for item in obj:
            d = _operand_device(item)
            if d is not None:
                return d
        return None
    if getattr(obj, "is_device", False) and hasattr(obj, "device_id"):
        return int(obj.device_id)
    st = getattr(obj, "stream", None)
    if isinstance(st, Stream):
        return st.device
    return None


def on_operand_device(fn):
    """Decorator for the host layer's device operators: run ``fn`` with the device of its first device operand current (restored
    afterwards).  The C ABI keeps HIP's rule — the caller selects the device a stream belongs to before it launches on it (the table
    caches, the workspace registry and a NULL stream handle all mean "the current device") — and the reference binds its context per
    call the same way (``ctx.bind_to_thread()``, T/cuda.rs); this is that bind, so an operator on a device-1 image works from a thread
    whose current device is 0 (tests/test_multi_device_gpu.py)."""
    import functools

    @functools.wraps(fn)
    def bound(*args, **kwargs):
        dev = None
        for a in args:
            dev = _operand_device(a)
            if dev is not None:
                break
        else:
            for a in kwargs.values():
                dev = _operand_device(a)
                if dev is not None:
                    break
        if dev is None:
            return fn(*args, **kwargs)
        try:
            current_device()
        except _ffi.KorniaHipError:      # no usable device at all: let the operator report its own typed error (residency, KH_ERR_HIP)
            return fn(*args, **kwargs)
        # a device exists: failing to select the operand's one is an error of THIS layer — running the operator with another device
        # current would file its table caches, workspace lookups and NULL-stream meaning under the wrong device
        with _device_guard(dev):
            return fn(*args, **kwargs)

    return bound


def device_info(device: int = 0) -> Tuple[str, int, int]:
    """(name, compute units, total bytes)."""
    name = C.create_string_buffer(256)
    cu = C.c_int32(0)
    mem = C.c_uint64(0)
    check(lib.kh_device_info(device, name, 256, C.byref(cu), C.byref(mem)))
    return name.value.decode(), int(cu.value), int(mem.value)


def mem_get_info() -> Tuple[int, int]:
    """``(free, total)`` bytes of the current device (cuda.pyi:64-70; the default stream is synchronised first, so a
    loop can be bracketed with it to assert that no device memory leaked)."""
    check(lib.kh_stream_synchronize(None))
    free, total = C.c_uint64(0), C.c_uint64(0)
    check(lib.kh_mem_get_info(C.byref(free), C.byref(total)))
    return int(free.value), int(total.value)


class Stream:
    """A HIP stream handle.  The stream's device selects where ``Image.to_hip(stream)`` /
    ``Image.zeros(..., stream=stream)`` place data."""

    def __init__(self, handle: int, device: int, owned: bool):
        self._handle = int(handle)
        self.device = int(device)
        self._owned = owned
        self._workspace = None   # owned streams: the registered scratch buffer (set_workspace)

    @staticmethod
    def new(device: int = 0) -> "Stream":
        prev = current_device()
        set_device(device)
        try:
            h = C.c_void_p(0)
            check(lib.kh_stream_create(C.byref(h)))
        finally:
            set_device(prev)
        return Stream(h.value or 0, device, owned=True)

    @staticmethod
    def default(device: int = 0) -> "Stream":
        return Stream(0, device, owned=False)

    @staticmethod
    def from_handle(handle: int, device: Optional[int] = None) -> "Stream":
        """Adopt a raw ``hipStream_t`` (e.g. ``torch.cuda.current_stream().cuda_stream``).
        Not owned: never destroyed here."""
        return Stream(int(handle), current_device() if device is None else device, owned=False)

    @staticmethod
    def from_cuda_stream(obj: Any) -> "Stream":
        """Adopt a stream from any object speaking ``__cuda_stream__() -> (version, handle)``,
        exposing ``.cuda_stream`` (torch), ``.ptr`` / ``.handle`` (cupy / cuda-python), or an int."""
        if isinstance(obj, Stream):
            return obj
        if isinstance(obj, int):
            return Stream.from_handle(obj)
        if hasattr(obj, "__cuda_stream__"):
            return Stream.from_handle(int(obj.__cuda_stream__()[1]), getattr(obj, "device_index", None))
        for attr in ("cuda_stream", "ptr", "handle"):
            if hasattr(obj, attr):
                dev = getattr(obj, "device_index", None)
                if dev is None and hasattr(obj, "device") and hasattr(obj.device, "index"):
                    dev = obj.device.index
                return Stream.from_handle(int(getattr(obj, attr)), dev)
        raise TypeError(f"cannot adopt a stream from {type(obj).__name__}")

    def synchronize(self) -> None:
        if self._handle:
            check(lib.kh_stream_synchronize(self._handle))
        else:   # the NULL handle means "the CURRENT device's default stream": Stream.default(1).synchronize() from a device-0 thread
            with _device_guard(self.device):
                check(lib.kh_stream_synchronize(0))

    def set_workspace(self, buf: Optional["DeviceBuffer"]) -> None:
        """Register ``buf`` as the scratch the operators with an intermediate (separable u8 resize, wide u8 blurs, u8 warps,
        Lanczos resize) use on this stream instead of allocating — what makes them capturable (``kh_stream_set_workspace``).
        ``None`` unregisters.  The registration — not this wrapper object — keeps the buffer alive: it lives in a module-level
        table keyed by (device, handle), like the C registry, so ``Stream.default(0).set_workspace(buf)`` on a temporary wrapper
        does not free ``buf`` behind the library's back; ``buf.free()`` and destroying an owned stream unregister first."""
        key = (self.device, self._handle)
        prev = current_device()
        set_device(self.device)  # the C registry files the entry under the current device
        try:
            if buf is None:
                check(lib.kh_stream_set_workspace(self._handle, None, 0))
            else:
                check(lib.kh_stream_set_workspace(self._handle, buf.ptr, buf.nbytes))
        finally:
            set_device(prev)
        old = _WORKSPACES.pop(key, None) or getattr(self, "_workspace", None)
        if old is not None:
            old._ws_keys.discard(key)
        self._workspace = None
        if buf is not None:
            buf._ws_keys.add(key)
            if self._owned:
                # An OWNED stream is one Python object: it keeps its workspace alive itself.  stream -> buf -> buf.stream is then
                # an ordinary reference cycle the collector can reclaim; a module-global entry would pin all three for the life
                # of the process — a service creating a stream + workspace per request leaked both.
                self._workspace = buf
            else:
                _WORKSPACES[key] = buf   # default / adopted handles: wrappers are temporaries, the registration must outlive them

    @property
    def cuda_stream_ptr(self) -> int:
        return self._handle

    hip_stream_ptr = cuda_stream_ptr

    def __cuda_stream__(self) -> Tuple[int, int]:
        return (0, self._handle)

    def __del__(self):
        if getattr(self, "_owned", False) and self._handle:
            try:
                key = (self.device, self._handle)
                old = _WORKSPACES.pop(key, None) or getattr(self, "_workspace", None)  # kh_stream_destroy erases the C entry
                self._workspace = None
                if old is not None:
                    old._ws_keys.discard(key)
                    if getattr(old, "stream", None) is self:
                        old.free()   # collected together with this stream: release it while the stream it frees on still exists
                lib.kh_stream_destroy(self._handle)
            except Exception:
                pass
            self._handle = 0

    def __repr__(self) -> str:
        return f"Stream(device={self.device}, handle=0x{self._handle:x})"


# (device, stream handle) -> the DeviceBuffer registered as that stream's workspace (the keepalive of Stream.set_workspace)
_WORKSPACES: dict = {}


class Event:
    """``device``: the device the event belongs to (HIP refuses to record an event on another device's stream); default = the
    calling thread's current device."""

    def __init__(self, timing: bool = True, device: Optional[int] = None):
        h = C.c_void_p(0)
        if device is None:
            check(lib.kh_event_create(C.byref(h), 1 if timing else 0))
        else:
            with _device_guard(device):
                check(lib.kh_event_create(C.byref(h), 1 if timing else 0))
        self._handle = h.value

    def record(self, stream: Stream) -> None:
        if stream.cuda_stream_ptr:
            check(lib.kh_event_record(self._handle, stream.cuda_stream_ptr))
        else:   # NULL handle: the default stream of the STREAM's device, not of whichever device is current
            with _device_guard(stream.device):
                check(lib.kh_event_record(self._handle, 0))

    def synchronize(self) -> None:
        check(lib.kh_event_synchronize(self._handle))

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float(0)
        check(lib.kh_event_elapsed_ms(self._handle, stop._handle, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        if getattr(self, "_handle", None):
            try:
                lib.kh_event_destroy(self._handle)
            except Exception:
                pass
            self._handle = None


class Graph:
    """A captured graph: record once, replay per frame at the cost of one launch (``kornia_rs.cuda.Graph``,
    cuda.pyi:80-90, PY/cuda_ext/mod.rs:1684-1790).  Capture requires allocation-free ops — pass preallocated ``out=`` /
    ``dst=`` device images — on a non-default stream (``Stream.new()``), with the operands allocated on that same
    stream.  ``retain`` keeps the operand objects (and their device memory) alive as long as the graph."""

    def __init__(self, handle: int, stream: Stream, retain):
        self._handle, self.stream, self._retained = handle, stream, list(retain)

    @staticmethod
    def capture(f, retain, stream: Optional[Stream] = None) -> "Graph":
        if stream is None or not stream.cuda_stream_ptr:
            raise _ffi.KorniaHipError(_ffi.KH_ERR_INVALID_ARG, "Graph.capture: capture needs a non-default stream (Stream.new())")
        with _device_guard(stream.device):   # the capture, the captured launches and the instantiation all belong to the stream's device
            check(lib.kh_graph_capture_begin(stream.cuda_stream_ptr))
            error = None
            try:
                f()
            except BaseException as e:  # always end the capture so the stream is usable again, then surface the error
                error = e
            handle = C.c_void_p()
            rc = lib.kh_graph_capture_end(stream.cuda_stream_ptr, C.byref(handle))
        if error is not None:
            if rc == _ffi.KH_OK:
                lib.kh_graph_destroy(handle)
            raise error
        if rc != _ffi.KH_OK:
            msg = _ffi.last_error()
            if "nothing was captured" in msg:
                raise ValueError("Graph.capture: nothing was captured (the callable enqueued no device work on this stream)")
            check(rc)
        return Graph(handle.value, stream, retain)

    def replay(self) -> None:
        with _device_guard(self.stream.device):
            check(lib.kh_graph_launch(self._handle, self.stream.cuda_stream_ptr))

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            lib.kh_graph_destroy(h)


_STAGE_BYTES = 16 << 20          # page-locked bounce memory per copying thread, used as two halves
_PIECE = _STAGE_BYTES // 2
_stage_local = threading.local()


def _stage(device: int):
    """This thread's page-locked bounce buffer and its two half-events for pageable host <-> device copies on ``device`` (16 MiB =
    two 8 MiB halves, allocated on first use; ``release_thread_staging`` or the end of the thread frees them).  Keyed by (thread,
    device): one thread may copy for several devices — ``ShardedBatch.numpy()`` walks every shard from the calling thread, and a pool
    worker is not pinned to one shard — and HIP rejects ``hipEventRecord`` when the event and the stream belong to different devices
    (``hipErrorInvalidHandle``), so the events (and the buffer) are created with the stream's device current and used for it only."""
    slots = getattr(_stage_local, "slots", None)
    if slots is None:
        slots = _stage_local.slots = {}
    slot = slots.get(device)
    if slot is None:
        with _device_guard(device):
            slot = slots[device] = (PinnedBuffer(_STAGE_BYTES), (Event(timing=False), Event(timing=False)))
    return slot


def release_thread_staging() -> None:
    """Free the calling thread's bounce buffers (``ShardPool.close`` runs this on every worker: a pool that is shut down must not
    keep 16 MiB pinned per worker and device for the life of the process)."""
    slots = getattr(_stage_local, "slots", None)
    _stage_local.slots = None
    for buf, _events in (slots or {}).values():
        buf.free()


def d2h(out: np.ndarray, device_ptr: int, stream: Stream) -> None:
    """Device -> pageable host copy (to_host, crates/kornia-tensor/src/cuda.rs:1258-1300), staged through page-locked memory:
    every transfer the runtime sees is a true stream-ordered DMA into pinned memory, followed by a host memcpy.  Handing the runtime
    a large PAGEABLE destination instead was seen to complete `hipStreamSynchronize` with a hole in the data (round 1: a 4 - 16 MiB
    run of stale bytes in a 48 MiB copy; once more in round 3 under four concurrent processes, r03m, with a single HIP runtime
    mapped) and small pageable copies were seen overtaking kernels queued on the stream — neither can happen to a pinned DMA.
    Double-buffered (round 4): the DMA of piece k + 1 into one half runs while the host copies piece k out of the other; each
    half has its own event, host-waited before the half is read.  The stream is drained once, before the first piece."""
    if out.nbytes == 0:
        return
    flat = out.reshape(-1).view(np.uint8) if out.flags["C_CONTIGUOUS"] else None
    if flat is None:
        tmp = np.empty(out.shape, out.dtype)
        d2h(tmp, device_ptr, stream)
        out[...] = tmp
        return
    with _device_guard(stream.device):   # Stream.default(d) is handle 0: "the current device's null stream"
        _d2h_pieces(flat, device_ptr, stream)


def _d2h_pieces(flat: np.ndarray, device_ptr: int, stream: Stream) -> None:
    stage, events = _stage(stream.device)
    view = stage.view()
    stream.synchronize()   # the producer kernels
    pieces = [(off, min(_PIECE, flat.size - off)) for off in range(0, flat.size, _PIECE)]

    def issue(i):
        off, n = pieces[i]
        check(lib.kh_memcpy_d2h_async(stage.ptr + (i & 1) * _PIECE, device_ptr + off, n, stream.cuda_stream_ptr))
        events[i & 1].record(stream)

    issue(0)
    for i, (off, n) in enumerate(pieces):
        if i + 1 < len(pieces):
            issue(i + 1)           # the other half: its previous contents were copied out in iteration i - 1
        events[i & 1].synchronize()
        h = (i & 1) * _PIECE
        flat[off:off + n] = view[h:h + n]


def h2d(device_ptr: int, a: np.ndarray, stream: Stream) -> None:
    """Pageable host -> device copy through the same two-half page-locked bounce buffer (see d2h): the host fills one half while the
    DMA of the other is in flight; a half is refilled only after the event of its last DMA."""
    a = np.ascontiguousarray(a)
    if a.nbytes == 0:
        return
    flat = a.reshape(-1).view(np.uint8)
    with _device_guard(stream.device):
        stage, events = _stage(stream.device)
        view = stage.view()
        stream.synchronize()   # a queued memset / kernel on this stream must not be overtaken
        for i, off in enumerate(range(0, flat.size, _PIECE)):
            n = min(_PIECE, flat.size - off)
            h = (i & 1) * _PIECE
            if i >= 2:
                events[i & 1].synchronize()   # the DMA that last read this half
            view[h:h + n] = flat[off:off + n]
            check(lib.kh_memcpy_h2d_async(device_ptr + off, stage.ptr + h, n, stream.cuda_stream_ptr))
            events[i & 1].record(stream)
        stream.synchronize()   # the bounce buffer is reused by the next call


def _stream_handle(stream: Optional[Stream]) -> int:
    return 0 if stream is None else stream.cuda_stream_ptr


class PinnedBuffer:
    """Page-locked host memory (``PinnedAllocator`` / ``zeros_pinned``, T/cuda.rs:355-380): the source
    of true stream-ordered DMA uploads.  ``view()`` is a writable uint8 numpy view."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib.kh_host_alloc(C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value

    def view(self) -> np.ndarray:
        return np.ctypeslib.as_array((C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr))[: self.nbytes]

    def free(self) -> None:
        p, self.ptr = self.ptr, None
        if p:
            lib.kh_host_free(p)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ManagedBuffer:
    """Managed (unified) memory: ONE allocation dereferenceable from the host and from kernels on any stream of the
    process (``CudaUnifiedAllocator`` / ``Backing::Managed``, T/cuda.rs:440-511: ``cuMemAllocManaged(ATTACH_GLOBAL)``,
    zero-filled).  It carries a stream like every device-accessible buffer so residency dispatch can launch on it;
    ``view()`` is a writable uint8 numpy view of the same bytes.  The host side must not touch the bytes while a kernel
    that uses them is in flight: ``Tensor.numpy()`` / ``as_slice`` drain the carried stream first.  Freed with
    ``hipFree`` (which waits for the device) after draining the stream."""

    def __init__(self, nbytes: int, stream: Optional[Stream] = None):
        self.stream = stream if stream is not None else Stream.default(current_device())
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        prev = current_device()
        set_device(self.stream.device)
        try:
            check(lib.kh_malloc_managed(C.byref(p), max(self.nbytes, 1)))  # the driver rejects a zero-size request (T/cuda.rs:468)
        finally:
            set_device(prev)
        self.ptr = p.value or 0

    def view(self) -> np.ndarray:
        return np.ctypeslib.as_array((C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr))[: self.nbytes]

    def free(self) -> None:
        p, self.ptr = self.ptr, 0
        if p:
            with _device_guard(self.stream.device):
                lib.kh_stream_synchronize(self.stream.cuda_stream_ptr)
                lib.kh_free(p)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """An owned, stream-ordered device allocation (``CudaResource`` with ``Backing::Device`` in
    crates/kornia-tensor/src/cuda.rs:89-169): carries its stream, freed on that stream."""

    def __init__(self, nbytes: int, stream: Optional[Stream] = None, zeroed: bool = True):
        self.stream = stream if stream is not None else Stream.default(current_device())
        self.nbytes = int(nbytes)
        self._ws_keys: set = set()  # (device, handle) registrations of this buffer as a stream workspace
        p = C.c_void_p(0)
        prev = current_device()
        set_device(self.stream.device)
        try:
            check(lib.kh_malloc_async(C.byref(p), self.nbytes, 1 if zeroed else 0,
                                      self.stream.cuda_stream_ptr))
        finally:
            set_device(prev)
        self.ptr = p.value or 0

    @staticmethod
    def from_numpy(a: np.ndarray, stream: Optional[Stream] = None) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = DeviceBuffer(a.nbytes, stream, zeroed=False)
        buf.copy_from_host(a)
        return buf

    def copy_from_host(self, a: np.ndarray, offset: int = 0) -> None:
        a = np.ascontiguousarray(a)
        assert offset + a.nbytes <= self.nbytes
        if a.nbytes == 0:
            return
        h2d(self.ptr + offset, a, self.stream)   # pageable source: through the page-locked bounce buffer

    def to_numpy(self, dtype, shape, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        d2h(out, self.ptr + offset, self.stream)
        return out

    def free(self) -> None:
        if self.ptr:
            for dev, handle in list(getattr(self, "_ws_keys", ())):  # never leave the library pointing at freed memory
                prev = current_device()
                try:
                    set_device(dev)
                    lib.kh_stream_set_workspace(handle, None, 0)
                finally:
                    set_device(prev)
                _WORKSPACES.pop((dev, handle), None)
                if getattr(self.stream, "_workspace", None) is self:
                    self.stream._workspace = None
            self._ws_keys = set()
            with _device_guard(self.stream.device):   # a null stream handle means "the current device's": free where it was allocated
                lib.kh_free_async(self.ptr, self.stream.cuda_stream_ptr)
            self.ptr = 0

    @property
    def device_id(self) -> int:
        return self.stream.device

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def last_workspace_bytes() -> int:
    """Scratch the last compute call on this thread asked for (0 = none): size ``Stream.set_workspace`` with it."""
    n = C.c_size_t(0)
    check(lib.kh_last_workspace_bytes(C.byref(n)))
    return int(n.value)


def runtime_info() -> dict:
    """Which HIP runtime image this process uses and how it was chosen (``_ffi._preload_hip_runtime``)."""
    return {"choice": _ffi.RUNTIME_CHOICE, "version": _ffi.RUNTIME_VERSION_NOTE, **_ffi.mapped_hip_runtimes()}


def pointer_domain(ptr: int) -> Tuple[int, int]:
    d = C.c_int32(0)
    dev = C.c_int32(-1)
    check(lib.kh_pointer_domain(ptr, C.byref(d), C.byref(dev)))
    return int(d.value), int(dev.value)
