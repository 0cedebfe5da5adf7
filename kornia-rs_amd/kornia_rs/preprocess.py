"""Fused camera preprocess — host side of ``kh_preprocess_to_chw``.

Mirrors ``kornia_imgproc::preprocess`` (crates/kornia-imgproc/src/preprocess.rs): the
``ResizeMode`` / ``Normalize`` / ``SourceFormat`` enums (:67-251), ``PreprocessError`` (:253-340),
the shared ``Affine`` geometry (:350-369), the builder defaults (:662-672) and the ``run_raw`` /
``run_raw_batch`` / ``run_surface`` / ``run`` entry points with their validation order
(:887-1375) — and the Python class ``kornia_rs.Preprocessor``
(kornia-py/python/kornia_rs/__init__.pyi:70-111).

Device only: the kernel is the product.  A ``Preprocessor`` needs a ``Stream``; host operands are
rejected with the reference's ``NotDeviceImage`` / ``NotDeviceTensor`` errors — never silently
processed on the CPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _ffi
from ._ffi import PreprocessParams, check, lib
from .hip import IMAGENET_MEAN, IMAGENET_STD, DeviceBuffer, Stream
from .tensor import Tensor

f32 = np.float32


class PreprocessError(ValueError):
    """``kind`` names the reference enum variant (P/preprocess.rs:253-340)."""

    def __init__(self, kind: str, message: str, **fields):
        super().__init__(message)
        self.kind = kind
        self.fields = fields


class ResizeMode:
    LETTERBOX = "letterbox"
    STRETCH = "stretch"


_SAMPLING = {"nearest": _ffi.KH_SAMPLE_NEAREST, "bilinear": _ffi.KH_SAMPLE_BILINEAR,
             "lanczos": _ffi.KH_SAMPLE_LANCZOS}


@dataclass(frozen=True)
class SourceFormat:
    """P/preprocess.rs:131-250."""

    name: str
    fmt_code: int
    bpp: int
    interleaved: bool

    def pitch(self, w: int) -> int:
        return w * self.bpp

    def buffer_len(self, w: int, h: int) -> int:
        chroma = w * h // 2 if self.name == "nv12" else 0
        return self.pitch(w) * h + chroma

    def dims_ok(self, w: int, h: int) -> bool:
        if self.name == "nv12":
            return w % 2 == 0 and h % 2 == 0
        if self.name == "yuyv":
            return w % 2 == 0
        return True

    @staticmethod
    def from_name(name: str) -> Optional["SourceFormat"]:
        return _FORMATS.get(name.lower())


_FORMATS = {}
for _names, _f in (
    (("rgb", "rgb8"), SourceFormat("rgb8", _ffi.KH_FMT_RGB, 3, True)),
    (("bgr", "bgr8"), SourceFormat("bgr8", _ffi.KH_FMT_BGR, 3, True)),
    (("rgba", "rgba8"), SourceFormat("rgba8", _ffi.KH_FMT_RGB, 4, True)),
    (("bgra", "bgra8"), SourceFormat("bgra8", _ffi.KH_FMT_BGR, 4, True)),
    (("gray", "gray8"), SourceFormat("gray8", _ffi.KH_FMT_GRAY, 1, False)),
    (("nv12",), SourceFormat("nv12", _ffi.KH_FMT_NV12, 1, False)),
    (("yuyv",), SourceFormat("yuyv", _ffi.KH_FMT_YUYV, 2, False)),
):
    for _n in _names:
        _FORMATS[_n] = _f


def affine(mode: str, sw: int, sh: int, dw: int, dh: int) -> Tuple[f32, f32, f32, f32]:
    """``Affine::new`` in f32 arithmetic (P/preprocess.rs:350-369): (scale_x, scale_y, pad_x, pad_y)."""
    sw_, sh_, dw_, dh_ = f32(sw), f32(sh), f32(dw), f32(dh)
    if mode == ResizeMode.LETTERBOX:
        s = min(f32(dw_ / sw_), f32(dh_ / sh_))
        return s, s, f32(f32(dw_ - f32(sw_ * s)) * f32(0.5)), f32(f32(dh_ - f32(sh_ * s)) * f32(0.5))
    return f32(dw_ / sw_), f32(dh_ / sh_), f32(0.0), f32(0.0)


def mean_inv_std(mean, std) -> Tuple[np.ndarray, np.ndarray]:
    """``Normalize::mean_inv_std`` (P/preprocess.rs:107-121); ``None`` = UnitScale."""
    if mean is None and std is None:
        return np.zeros(3, f32), np.ones(3, f32)
    mean = np.asarray(IMAGENET_MEAN if mean is None else mean, dtype=f32)
    std = np.asarray(IMAGENET_STD if std is None else std, dtype=f32)
    if mean.shape != (3,) or std.shape != (3,):
        raise PreprocessError("InvalidNormalize", "mean/std must have 3 entries")
    if not (np.all(np.isfinite(std)) and np.all(std > 0) and np.all(np.isfinite(mean))):
        raise PreprocessError(
            "InvalidNormalize",
            f"invalid normalize: mean {mean.tolist()} must be finite, std {std.tolist()} must be finite and > 0",
            mean=mean.tolist(), std=std.tolist())
    return mean, (f32(1.0) / std).astype(f32)


RawSource = Union[DeviceBuffer, Tensor, int]


class Normalize:
    """``Normalize`` (P/preprocess.rs:78-121): ``unit_scale()`` (divide by 255, the default), ``mean_std(mean, std)`` in the
    [0, 1] domain, ``imagenet()`` = torchvision's constants."""

    def __init__(self, mean: Optional[Sequence[float]] = None, std: Optional[Sequence[float]] = None):
        self.mean, self.std = (None, None) if mean is None and std is None else (tuple(mean), tuple(std))

    @staticmethod
    def unit_scale() -> "Normalize":
        return Normalize()

    @staticmethod
    def mean_std(mean: Sequence[float], std: Sequence[float]) -> "Normalize":
        return Normalize(mean, std)

    @staticmethod
    def imagenet() -> "Normalize":
        return Normalize(IMAGENET_MEAN, IMAGENET_STD)

    def __eq__(self, other):
        return isinstance(other, Normalize) and (self.mean, self.std) == (other.mean, other.std)


class PreprocessorBuilder:
    """``PreprocessorBuilder`` (P/preprocess.rs:654-785): chainable configuration with the reference's defaults (letterbox,
    unit scale, pad 114, bilinear, RGB8).  ``build_hip(stream)`` is ``build_cuda(stream)``; ``build()`` — the reference's CPU
    preprocessor — does not exist in this backend and says so."""

    def __init__(self):
        self._mode, self._normalize, self._pad_value = ResizeMode.LETTERBOX, Normalize.unit_scale(), 114
        self._sampling, self._format = "bilinear", "rgb"

    @staticmethod
    def new() -> "PreprocessorBuilder":
        return PreprocessorBuilder()

    def source_format(self, format) -> "PreprocessorBuilder":
        self._format = format.name if isinstance(format, SourceFormat) else format
        return self

    def mode(self, mode: str) -> "PreprocessorBuilder":
        self._mode = mode
        return self

    def normalize(self, normalize: Normalize) -> "PreprocessorBuilder":
        self._normalize = normalize
        return self

    def pad_value(self, pad_value: int) -> "PreprocessorBuilder":
        if not 0 <= int(pad_value) <= 255:
            raise ValueError("pad_value is a u8")
        self._pad_value = int(pad_value)
        return self

    def sampling(self, sampling: str) -> "PreprocessorBuilder":
        self._sampling = sampling
        return self

    def build(self) -> "Preprocessor":
        raise PreprocessError("NotDeviceImage", "PreprocessorBuilder.build(): the CPU preprocessor lives in the reference crate; this backend "
                                                "builds device preprocessors only — build_hip(stream)")

    def build_hip(self, stream: Stream, f16: bool = False) -> "Preprocessor":
        return Preprocessor(self._mode, self._format, self._sampling, f16, self._normalize.mean, self._normalize.std, self._pad_value, stream)



def _ptr_len(src: Any) -> Tuple[int, Optional[int]]:
    if isinstance(src, (DeviceBuffer, _DeviceView)):
        return src.ptr, src.nbytes
    if isinstance(src, Tensor):
        if not src.is_device:
            raise PreprocessError("NotDeviceImage",
                                  "HIP preprocessor requires a device-resident source image")
        return src.data_ptr, src.nbytes
    if hasattr(src, "__cuda_array_interface__"):
        cai = src.__cuda_array_interface__
        n = int(np.prod(cai["shape"], dtype=np.int64)) * np.dtype(cai["typestr"]).itemsize
        return int(cai["data"][0]), n
    if isinstance(src, np.ndarray):
        raise PreprocessError("NotDeviceImage",
                              "HIP preprocessor requires a device-resident source image")
    return int(src), None


class _Slot:
    """One entry of the staging ring: a page-locked host buffer, a device buffer, the event after the last upload INTO the slot
    (host-waited before the pinned bytes change) and the event after the last kernel that READ the slot's device buffer (the copy
    stream waits for it before the next upload overwrites that buffer)."""

    __slots__ = ("pinned", "device", "upload_done", "consumed", "used")

    def __init__(self):
        self.pinned = self.device = self.upload_done = self.consumed = None
        self.used = False


class _Staging:
    """Persistent upload staging for host frames (``Staging``, PY/cuda_ext/mod.rs:647-745), as a TWO-DEEP RING on a copy stream.

    The reference keeps one page-locked buffer + one device buffer per preprocessor and host-waits the previous upload before
    the pinned bytes are overwritten; with one slot on one stream, call k + 1 can neither copy its frames into pinned memory nor
    start its DMA while kernel k runs.  Here call k uses slot ``k % 2``:

      1. host-wait the upload issued from this slot two calls ago (long finished), grow the slot's buffers if needed;
      2. host memcpy of the frames into the slot's pinned buffer — kernel k - 1 and upload k - 1 are still in flight;
         with ``zero_copy=True`` (opt-in) frames that ALREADY live in page-locked memory (``hip.PinnedBuffer`` capture buffers)
         skip this copy and are DMA'd from where they are — the caller must then leave them alone until ``wait_uploads()``;
      3. on the COPY stream: wait for the kernel that last read this slot's device buffer (call k - 2), H2D, record ``upload_done``;
      4. the compute stream waits for that one event and the caller launches the kernel; ``mark_consumed`` records the event step 3 of
         call k + 2 will wait for.

    Page-locking is far too expensive for a frame loop (buffers are grown on demand, never per call: ``allocations`` counts them),
    and pinned memory makes the H2D copy a stream-ordered DMA."""

    DEPTH = 2

    def __init__(self):
        self.slots = [_Slot() for _ in range(self.DEPTH)]
        self.turn = 0
        self.current: Optional[_Slot] = None
        self.copy_stream: Optional[Stream] = None
        self.allocations = 0   # grows only; exposed for the tests
        self.zero_copy_uploads = 0
        self.last_upload = None   # the newest upload's event (whatever its slot)

    @staticmethod
    def _pinned_sources(frames) -> bool:
        """True when every frame's bytes are page-locked host memory the runtime can DMA from directly."""
        from .hip import pointer_domain
        try:
            return all(pointer_domain(int(fr.ctypes.data))[0] == _ffi.KH_DOMAIN_HOST_PINNED for fr in frames)
        except Exception:
            return False

    def upload(self, stream: Stream, frames, zero_copy: bool = False):
        """``zero_copy=True`` lets frames that already live in page-locked memory be DMA'd from where they are (no host copy): the
        caller then owns the hazard — such a frame must not be rewritten before ``wait_uploads()``.  The default copies every frame
        into the ring's own pinned slot, so the caller may reuse its frames as soon as this returns, pinned or not: the reference's
        ``Staging`` contract (PY/cuda_ext/mod.rs:647-745)."""
        from .hip import _device_guard
        with _device_guard(stream.device):   # events / buffers are created for the stream's device, whatever the caller had current
            return self._upload(stream, frames, zero_copy)

    def _upload(self, stream: Stream, frames, zero_copy: bool):
        from .hip import Event, PinnedBuffer
        slot = self.slots[self.turn % self.DEPTH]
        self.turn += 1
        self.current = slot
        frame_len = int(frames[0].size)
        stride = (frame_len + 255) // 256 * 256  # keep every frame 256-byte aligned on the device
        total = stride * len(frames)
        if slot.upload_done is not None:  # wait_prev_upload: BEFORE touching (or re-allocating) the slot's pinned buffer
            slot.upload_done.synchronize()
            slot.upload_done = None
        if self.copy_stream is None:
            self.copy_stream = Stream.new(stream.device)
        cs = self.copy_stream
        zero_copy = bool(zero_copy) and self._pinned_sources(frames)
        if not zero_copy and (slot.pinned is None or slot.pinned.nbytes < total):
            slot.pinned = PinnedBuffer(total)
            self.allocations += 1
        fresh_device = slot.device is None or slot.device.nbytes < total
        if fresh_device:
            slot.device = DeviceBuffer(total, stream, zeroed=False)   # stream-ordered on the COMPUTE stream ...
            self.allocations += 1
        if not zero_copy:
            self._host_copy(slot.pinned, frames, frame_len, stride)
        # the copy stream may write the slot's device buffer once (a) a fresh allocation exists there and (b) the kernel that last
        # read the buffer has finished; both are events of the compute stream
        if fresh_device or (slot.used and slot.consumed is None):
            fence = Event(timing=False)      # ... so the copy stream waits for the allocation (and, if the caller never marked the
            fence.record(stream)             # last read, for everything queued so far: correct, merely without overlap)
            check(lib.kh_stream_wait_event(cs.cuda_stream_ptr, fence._handle))
        elif slot.consumed is not None:
            check(lib.kh_stream_wait_event(cs.cuda_stream_ptr, slot.consumed._handle))
        # One upload in the copy queue at a time.  Uploads are serial on the copy stream anyway, but a copy ENQUEUED while the previous
        # one is still running is handed to whichever copy engine is free at that moment, or to a shader copy: on MI355X every
        # second 199 MB batch then took 6.6 ms instead of 3.5 (profiles/r04zh_h2d_slots.txt).  The host waits for the previous
        # upload (not for any kernel) before it queues the next one; the gap this leaves on the link is one wake-up.
        if self.last_upload is not None:
            self.last_upload.synchronize()
        if zero_copy:
            self.zero_copy_uploads += 1
            ptrs = [int(fr.ctypes.data) for fr in frames]
            gaps = {q - p_ for p_, q in zip(ptrs, ptrs[1:])}
            if len(frames) == 1 or gaps == {stride}:   # one contiguous run at the device stride: one DMA
                check(lib.kh_memcpy_h2d_async(slot.device.ptr, ptrs[0], stride * (len(frames) - 1) + frame_len, cs.cuda_stream_ptr))
            else:
                for k, ptr in enumerate(ptrs):
                    check(lib.kh_memcpy_h2d_async(slot.device.ptr + k * stride, ptr, frame_len, cs.cuda_stream_ptr))
        else:
            check(lib.kh_memcpy_h2d_async(slot.device.ptr, slot.pinned.ptr, total, cs.cuda_stream_ptr))
        ev = Event(timing=False)
        ev.record(cs)  # mark_upload
        slot.upload_done = self.last_upload = ev
        check(lib.kh_stream_wait_event(stream.cuda_stream_ptr, ev._handle))   # the kernel of THIS call waits for THIS upload only
        slot.used, slot.consumed = True, None
        return _DeviceView(slot.device, frame_len if len(frames) == 1 else total), stride

    _copy_pool = None   # shared by every preprocessor of the process
    copy_workers = 8    # memcpy workers of that pool (set before the first pageable upload)

    @classmethod
    def _host_copy(cls, pinned, frames, frame_len: int, stride: int) -> None:
        """Pageable frames -> the slot's page-locked buffer.  One thread moves ~25 GB/s here, half of what the host link takes
        (57 GB/s measured, profiles/r04e): above 4 MiB the frames are split over eight workers (ctypes.memmove drops the GIL)."""
        import ctypes
        total = frame_len * len(frames)
        srcs = [np.ascontiguousarray(fr) for fr in frames]
        if total < (4 << 20) or len(frames) < 2:
            view = pinned.view()
            for k, fr in enumerate(srcs):
                view[k * stride: k * stride + frame_len] = fr
            return
        if cls._copy_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            cls._copy_pool = ThreadPoolExecutor(max_workers=cls.copy_workers, thread_name_prefix="kornia-stage")
        base = pinned.ptr

        def move(lo, hi):
            for k in range(lo, hi):
                ctypes.memmove(base + k * stride, srcs[k].ctypes.data, frame_len)

        n, w = len(srcs), cls.copy_workers
        futs = [cls._copy_pool.submit(move, n * i // w, n * (i + 1) // w) for i in range(w)]
        for f in futs:
            f.result()

    def mark_consumed(self, stream: Stream) -> None:
        """Call after the kernel that reads the last ``upload`` has been enqueued on ``stream``."""
        from .hip import Event, _device_guard
        if self.current is not None:
            with _device_guard(stream.device):
                ev = Event(timing=False)
                ev.record(stream)
            self.current.consumed = ev

    def wait_uploads(self) -> None:
        """Block until every upload issued so far has left host memory (zero-copy capture buffers may then be rewritten)."""
        for slot in self.slots:
            if slot.upload_done is not None:
                slot.upload_done.synchronize()


class _DeviceView:
    """A (ptr, nbytes) view of the staging device buffer, accepted by ``_ptr_len``."""

    def __init__(self, buf: DeviceBuffer, nbytes: int):
        self._buf, self.ptr, self.nbytes = buf, buf.ptr, nbytes


class Preprocessor:
    """Resize (+pad) + normalise a raw frame into ``[N, 3, H, W]`` on a HIP stream."""

    def __init__(self, mode: str = "letterbox", format: str = "rgb", sampling: str = "bilinear",
                 f16: bool = False, mean: Optional[Sequence[float]] = None,
                 std: Optional[Sequence[float]] = None, pad_value: float = 114,
                 stream: Optional[Stream] = None):
        if mode not in (ResizeMode.LETTERBOX, ResizeMode.STRETCH):
            raise ValueError(f"unknown mode {mode!r} (expected 'letterbox' or 'stretch')")
        fmt = SourceFormat.from_name(format)
        if fmt is None:
            raise ValueError(f"unknown source format {format!r}")
        if sampling not in _SAMPLING:
            raise PreprocessError(
                "UnsupportedSampling",
                f"unsupported sampling mode {sampling!r} (expected Nearest, Bilinear, or Lanczos)")
        if stream is None:
            raise PreprocessError(
                "NotDeviceImage",
                "this build has no CPU preprocessor: pass stream=Stream.default(device) "
                "(the device kernel is the product; there is no CPU fallback)")
        self.mode = mode
        self.source_format = fmt
        self.sampling = sampling
        self.f16 = bool(f16)
        self.mean, self.inv_std = mean_inv_std(mean, std)
        self.pad_value = f32(pad_value)
        self.stream = stream
        self._staging = _Staging()
        # what a per-device replica is built from (ShardedPreprocessor; kernel parameters are replicated)
        self._config = dict(mode=mode, format=format, sampling=sampling, f16=bool(f16), mean=mean, std=std, pad_value=pad_value)
        self._sharders: dict = {}

    # -- helpers ------------------------------------------------------------------------------
    def _params(self, sw: int, sh: int, pitch: int, bpp: int, fmt_code: int, dw: int, dh: int,
                nframes: int, src_stride: int, out_f16: bool, force_generic: bool) -> PreprocessParams:
        lim = 2**31 - 1
        if sw > lim or sh > lim or pitch > lim or dw * dh > lim:
            raise PreprocessError("DimensionsTooLarge",
                                  "dimensions exceed the 32-bit kernel index limit")
        sx, sy, px, py = affine(self.mode, sw, sh, dw, dh)
        p = PreprocessParams()
        p.scale_x, p.scale_y, p.pad_x, p.pad_y = sx, sy, px, py
        p.src_w, p.src_h, p.src_pitch, p.src_bpp, p.fmt = sw, sh, pitch, bpp, fmt_code
        p.dst_w, p.dst_h = dw, dh
        for c in range(3):
            p.mean[c] = self.mean[c]
            p.inv_std[c] = self.inv_std[c]
        p.pad_value = self.pad_value
        p.sampling = _SAMPLING[self.sampling]
        p.out_dtype = _ffi.KH_OUT_F16 if out_f16 else _ffi.KH_OUT_F32
        p.nframes = nframes
        p.flags = _ffi.KH_PRE_FORCE_GENERIC if force_generic else 0
        p.src_frame_stride = src_stride
        p.dst_frame_stride = 3 * dw * dh
        return p

    @staticmethod
    def _validate_dst(dst: Tensor, expected_n: int, want_f16: bool) -> None:
        """validate_dst_shape (P/preprocess.rs:805-818) + residency."""
        if not isinstance(dst, Tensor) or not dst.is_device:
            raise PreprocessError("NotDeviceTensor",
                                  "HIP preprocessor requires a device-resident destination tensor")
        shape = dst.shape
        if len(shape) != 4 or shape[1] != 3 or (expected_n == 1 and shape[0] != 1):
            raise PreprocessError("BadOutputShape",
                                  f"destination tensor must be [1, 3, H, W], got {list(shape)}",
                                  shape=list(shape))
        if shape[0] != expected_n:
            raise PreprocessError("BatchMismatch",
                                  f"destination batch dim {shape[0]} != frame count {expected_n}",
                                  dst_n=shape[0], frames=expected_n)
        want = "float16" if want_f16 else "float32"
        if dst.dtype != want:
            raise PreprocessError("BadOutputShape", f"destination dtype {dst.dtype} != {want}")

    def _validate_raw(self, got: Optional[int], w: int, h: int) -> None:
        """validate_raw (P/preprocess.rs:1287-1301)."""
        f = self.source_format
        need = f.buffer_len(w, h)
        if not f.dims_ok(w, h) or w <= 0 or h <= 0 or (got is not None and got < need):
            raise PreprocessError(
                "InvalidRawSource",
                f"invalid raw source for {f.name} at {w}x{h} (got {got} bytes, need {need})",
                format=f.name, width=w, height=h, got=got, need=need)

    def _launch(self, src_ptr: Optional[int], dst: Tensor, p: PreprocessParams, frame_ptrs: Optional[Sequence[int]] = None) -> None:
        # Launch on the preprocessor's stream.  If dst carries another stream it is fenced in before the launch and the
        # launch stream is fenced back into it afterwards (DeviceExec::for_streams + run, P/cuda/dispatch.rs:50-82), so
        # dst's own stream — its numpy(), the next op that reads it, its stream-ordered free — is ordered after the kernel.
        other = dst.stream is not None and dst.stream.cuda_stream_ptr != self.stream.cuda_stream_ptr
        if other:
            check(lib.kh_stream_fence(dst.stream.cuda_stream_ptr, self.stream.cuda_stream_ptr))
        if frame_ptrs is not None:
            check(lib.kh_preprocess_to_chw_list(self.stream.cuda_stream_ptr, _ffi.pointer_array(frame_ptrs), dst.data_ptr, C.byref(p)))
        else:
            check(lib.kh_preprocess_to_chw(self.stream.cuda_stream_ptr, src_ptr, dst.data_ptr, C.byref(p)))
        if other:
            check(lib.kh_stream_fence(self.stream.cuda_stream_ptr, dst.stream.cuda_stream_ptr))

    # -- Rust-shaped entry points ---------------------------------------------------------------
    def run_raw(self, src: RawSource, src_w: int, src_h: int, dst: Tensor, *,
                _force_generic: bool = False) -> None:
        """One raw frame -> ``[1, 3, H, W]`` (P/preprocess.rs:1184-1222)."""
        want_f16 = isinstance(dst, Tensor) and dst.dtype == "float16"
        self._validate_dst(dst, 1, want_f16)
        ptr, n = _ptr_len(src)
        self._validate_raw(n, src_w, src_h)
        f = self.source_format
        p = self._params(src_w, src_h, f.pitch(src_w), f.bpp, f.fmt_code, dst.shape[3], dst.shape[2],
                         1, 0, want_f16, _force_generic)
        self._launch(ptr, dst, p)

    def sharded(self, devices: Sequence[int]):
        """This configuration replicated over ``devices`` — one stream, one worker thread and one ``Preprocessor`` per
        ordinal (``kornia_rs.sharding.ShardedPreprocessor``; cached per device list)."""
        from .sharding import ShardedPreprocessor
        key = tuple(int(d) for d in devices)
        if key not in self._sharders:
            self._sharders[key] = ShardedPreprocessor(key, **self._config)
        return self._sharders[key]

    def run_raw_batch(self, frames: Union[Sequence[RawSource], RawSource], src_w: int, src_h: int,
                      dst: Any, *, frame_stride: Optional[int] = None, devices: Optional[Sequence[int]] = None,
                      _force_generic: bool = False) -> Any:
        """``N`` same-sized raw frames -> ``[N, 3, H, W]`` (P/preprocess.rs:1234-1282).

        ``frames`` is either a sequence of per-frame device buffers (the reference signature) or
        ONE device buffer holding ``N`` frames ``frame_stride`` bytes apart.  Equally-spaced
        frames go out as a single launch; separately allocated ones as one launch per 256 frames
        (``kh_preprocess_to_chw_list``: the frame bases travel in the kernel arguments) — the
        reference launches once per frame (:1277-1280).

        ``devices=[g0, g1, ...]`` shards the batch across GPUs in this process (contiguous slices, one
        host thread + one stream per device, no collective): ``frames`` is then a host ``[N, frame_bytes]`` uint8 array /
        list of host frames, or one ``(device_buffer, n_frames)`` pair per device; ``dst`` is a list with one
        ``[n_g, 3, H, W]`` tensor per device, or an ``(out_height, out_width)`` pair to have them allocated.  Returns a
        ``sharding.ShardedBatch``."""
        if devices is not None:
            sp = self.sharded(devices)
            if isinstance(dst, (tuple, list)) and len(dst) == 2 and all(isinstance(v, int) for v in dst):
                return sp.run_raw_batch(frames, src_w, src_h, dst[0], dst[1], frame_stride=frame_stride)
            outs = list(dst)
            if not outs or len(outs) != sp.world:
                raise PreprocessError("BatchMismatch", f"devices={list(devices)} needs one destination tensor per device, got {len(outs)}",
                                      dst_n=len(outs), frames=sp.world)
            first = next(t for t in outs if t is not None)
            return sp.run_raw_batch(frames, src_w, src_h, first.shape[2], first.shape[3], frame_stride=frame_stride, out=outs)
        f = self.source_format
        need = f.buffer_len(src_w, src_h)
        if frame_stride is not None:
            n_frames = dst.shape[0] if isinstance(dst, Tensor) and len(dst.shape) == 4 else 0
            want_f16 = isinstance(dst, Tensor) and dst.dtype == "float16"
            self._validate_dst(dst, n_frames, want_f16)
            ptr, n = _ptr_len(frames)
            if frame_stride < need:
                raise PreprocessError("InvalidRawSource",
                                      f"frame stride {frame_stride} shorter than a frame ({need})",
                                      format=f.name, width=src_w, height=src_h, got=frame_stride, need=need)
            got = n if (n is None or n_frames == 0) else n - (n_frames - 1) * frame_stride
            self._validate_raw(got, src_w, src_h)
            ptrs = [ptr + k * frame_stride for k in range(n_frames)]
        else:
            frames = list(frames)
            want_f16 = isinstance(dst, Tensor) and dst.dtype == "float16"
            self._validate_dst(dst, len(frames), want_f16)
            ptrs = []
            for fr in frames:
                ptr, n = _ptr_len(fr)
                self._validate_raw(n, src_w, src_h)
                ptrs.append(ptr)
        if not ptrs:
            return
        dw, dh = dst.shape[3], dst.shape[2]
        strides = {b - a for a, b in zip(ptrs, ptrs[1:])}
        if len(strides) <= 1 and (not strides or next(iter(strides)) >= 0):
            stride = strides.pop() if strides else 0
            p = self._params(src_w, src_h, f.pitch(src_w), f.bpp, f.fmt_code, dw, dh, len(ptrs),
                             stride, want_f16, _force_generic)
            self._launch(ptrs[0], dst, p)
        else:
            # separately allocated frame buffers — the reference's own signature (&[&CudaSlice<u8>]): the bases go to the
            # library as a host pointer array and out as ceil(N / 256) launches (kh_preprocess_to_chw_list), not N
            p = self._params(src_w, src_h, f.pitch(src_w), f.bpp, f.fmt_code, dw, dh, len(ptrs), 0, want_f16, _force_generic)
            self._launch(None, dst, p, frame_ptrs=ptrs)

    def run_surface(self, data: RawSource, width: int, height: int, row_pitch: int, channels: int,
                    dst: Tensor) -> None:
        """Pitched interleaved surface (``PitchedSurface``, P/preprocess.rs:380-391, 1112-1180)."""
        want_f16 = isinstance(dst, Tensor) and dst.dtype == "float16"
        self._validate_dst(dst, 1, want_f16)
        f = self.source_format
        if not f.interleaved:
            raise PreprocessError("FormatNeedsRawBuffer",
                                  f"source format {f.name} needs run_raw (raw device buffer), not the typed run()")
        if channels not in (3, 4):
            raise PreprocessError("UnsupportedChannels",
                                  f"unsupported source channel count {channels} (expected 3 or 4)")
        ptr, n = _ptr_len(data)
        if (row_pitch < width * channels or width == 0 or height == 0
                or (n is not None and n < row_pitch * height)):
            raise PreprocessError("InvalidSurface",
                                  "invalid pitched surface (need pitch >= width*channels and len >= pitch*height)")
        if f.bpp == 4 and channels != 4:
            raise PreprocessError("FormatNeedsRawBuffer", f"source format {f.name} needs 4 channels")
        p = self._params(width, height, row_pitch, channels, f.fmt_code, dst.shape[3], dst.shape[2],
                         1, 0, want_f16, False)
        self._launch(ptr, dst, p)

    def run_image(self, image: Any, dst: Tensor) -> None:
        """Typed ``run(&Image<u8, C>, &mut Tensor)`` (P/preprocess.rs:887-905): interleaved
        formats only; ``C`` comes from the image."""
        c = int(image.channels)
        if c not in (3, 4):
            raise PreprocessError("UnsupportedChannels",
                                  f"unsupported source channel count {c} (expected 3 or 4)")
        want_f16 = isinstance(dst, Tensor) and dst.dtype == "float16"
        self._validate_dst(dst, 1, want_f16)
        f = self.source_format
        ok = (c == 4) if f.name in ("rgba8", "bgra8") else f.interleaved
        if not ok:
            raise PreprocessError("FormatNeedsRawBuffer",
                                  f"source format {f.name} needs run_raw (raw device buffer), not the typed run()")
        if not image.is_device:
            raise PreprocessError("NotDeviceImage",
                                  "HIP preprocessor requires a device-resident source image")
        w, h = int(image.width), int(image.height)
        p = self._params(w, h, w * c, c, f.fmt_code, dst.shape[3], dst.shape[2], 1, 0, want_f16, False)
        self._launch(image.data_ptr, dst, p)

    # f16 twins (run_f16 / run_surface_f16 / run_raw_f16 / run_raw_batch_f16, P/preprocess.rs:1086-1282): the same launch into a
    # float16 destination; a float32 tensor is the reference's dtype mismatch, rejected before any device work
    @staticmethod
    def _need_f16(dst: Tensor, what: str) -> None:
        if not isinstance(dst, Tensor) or dst.dtype != "float16":
            raise PreprocessError("BadOutputShape", f"{what}: destination must be a float16 device tensor")

    def run_f16(self, image: Any, dst: Tensor) -> None:
        self._need_f16(dst, "run_f16")
        self.run_image(image, dst)

    def run_surface_f16(self, data: RawSource, width: int, height: int, row_pitch: int, channels: int, dst: Tensor) -> None:
        self._need_f16(dst, "run_surface_f16")
        self.run_surface(data, width, height, row_pitch, channels, dst)

    def run_raw_f16(self, src: RawSource, src_w: int, src_h: int, dst: Tensor) -> None:
        self._need_f16(dst, "run_raw_f16")
        self.run_raw(src, src_w, src_h, dst)

    def run_raw_batch_f16(self, frames, src_w: int, src_h: int, dst: Tensor, *, frame_stride: Optional[int] = None) -> None:
        self._need_f16(dst, "run_raw_batch_f16")
        self.run_raw_batch(frames, src_w, src_h, dst, frame_stride=frame_stride)

    # constructors of the Rust API (P/preprocess.rs:838-880)
    @staticmethod
    def builder() -> "PreprocessorBuilder":
        return PreprocessorBuilder()

    @staticmethod
    def letterbox(stream: Stream) -> "Preprocessor":
        return Preprocessor.with_mode(stream, ResizeMode.LETTERBOX)

    @staticmethod
    def stretch(stream: Stream) -> "Preprocessor":
        return Preprocessor.with_mode(stream, ResizeMode.STRETCH)

    @staticmethod
    def with_mode(stream: Stream, mode: str) -> "Preprocessor":
        return PreprocessorBuilder().mode(mode).build_hip(stream)

    # -- Python-shaped entry points (kornia_rs/__init__.pyi:88-111) ------------------------------
    def alloc_output(self, out_height: int, out_width: int, batch: int = 1) -> Tensor:
        return Tensor.zeros((batch, 3, out_height, out_width),
                            "float16" if self.f16 else "float32", stream=self.stream)

    def run_host_batch(self, frames: Sequence[np.ndarray], width: int, height: int, dst: Tensor, *, zero_copy: bool = False) -> None:
        """``N`` same-sized HOST frames (1-D uint8 arrays) -> ``dst`` ``[N, 3, H, W]``: staged through the two-deep upload ring (see
        ``_Staging``) — the host copy and the DMA of this call overlap the previous call's kernel — then ONE batched launch.
        The frames may be reused as soon as this returns (the reference's staging contract), page-locked or not.
        ``zero_copy=True`` is the capture-ring fast path: frames that live in page-locked memory (``hip.PinnedBuffer``,
        ``hipHostRegister``-ed or torch ``pin_memory`` buffers) are DMA'd in place, and the CALLER must not rewrite them before
        ``wait_uploads()``; frames that are not page-locked are staged as usual."""
        frames = [np.asarray(a).reshape(-1) for a in frames]
        if any(a.dtype != np.uint8 for a in frames):
            raise TypeError("raw frames must be uint8")
        if len({a.size for a in frames}) != 1:
            raise PreprocessError("InvalidRawSource", "batched frames must have the same length")
        dev, stride = self._staging.upload(self.stream, frames, zero_copy=zero_copy)
        self.run_raw_batch(dev, width, height, dst, frame_stride=stride)
        self._staging.mark_consumed(self.stream)

    def wait_uploads(self) -> None:
        """Block until every host -> device upload issued by ``run`` / ``run_host_batch`` has completed."""
        self._staging.wait_uploads()

    def run(self, frame: Any, width: int, height: int, out_height: int, out_width: int,
            out: Optional[Tensor] = None, consumer_stream: Any = None) -> Tensor:
        """A raw frame (1-D uint8 numpy array, uploaded on the preprocessor's stream), a list of
        them (batch), a device buffer, or an interleaved device ``Image``."""
        frames: Optional[List[Any]] = None
        if isinstance(frame, (list, tuple)):
            frames = list(frame)
            if out is not None:
                raise ValueError("out= is only valid for a single raw frame")
        n = len(frames) if frames is not None else 1
        dst = out if out is not None else self.alloc_output(out_height, out_width, n)
        if (dst.shape[2], dst.shape[3]) != (out_height, out_width):
            raise PreprocessError("BadOutputShape",
                                  f"out= is {list(dst.shape)}, expected [*, 3, {out_height}, {out_width}]")

        def host(a):
            if isinstance(a, np.ndarray):
                if a.dtype != np.uint8:
                    raise TypeError("raw frames must be uint8")
                return True
            return False

        if frames is not None and frames and all(host(a) for a in frames):
            sizes = {a.size for a in frames}
            if len(sizes) != 1:
                raise PreprocessError("InvalidRawSource", "batched frames must have the same length")
            self.run_host_batch(frames, width, height, dst)
        elif frames is not None:
            self.run_raw_batch([DeviceBuffer.from_numpy(a.reshape(-1), self.stream) if host(a) else a for a in frames],
                               width, height, dst)
        elif hasattr(frame, "channels") and hasattr(frame, "is_device"):
            self.run_image(frame, dst)
        elif host(frame):
            dev, _ = self._staging.upload(self.stream, [frame.reshape(-1)])
            self.run_raw(dev, width, height, dst)
            self._staging.mark_consumed(self.stream)
        else:
            self.run_raw(frame, width, height, dst)
        if consumer_stream is not None:
            cs = Stream.from_cuda_stream(consumer_stream)
            check(lib.kh_stream_fence(self.stream.cuda_stream_ptr, cs.cuda_stream_ptr))
        return dst


def _bind_preprocessor_methods() -> None:
    """Every launching method of ``Preprocessor`` runs with its stream's device current (``hip.on_operand_device``: the preprocessor
    carries its stream), like the reference binds its context per call."""
    from .hip import on_operand_device
    for name in ("run_raw", "run_raw_batch", "run_raw_f16", "run_raw_batch_f16", "run_surface", "run_image", "alloc_output",
                 "run_host_batch", "run"):
        setattr(Preprocessor, name, on_operand_device(getattr(Preprocessor, name)))


_bind_preprocessor_methods()
