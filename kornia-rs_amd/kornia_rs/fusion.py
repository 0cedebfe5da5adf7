"""Fused per-pixel pipelines — host mirror of ``kornia_imgproc::cuda::fusion`` (P/cuda/fusion.rs).

A pipeline is ``[source, map..., sink]`` over a destination grid: the source produces an f32 RGB value
per output pixel, maps transform it in registers, the sink writes it; nothing touches memory in
between.  Stage classes and ``FusedPipeline.{build, build_batched, launch, launch_batched,
generated_source}`` follow the reference (fusion.rs:196-690); errors are ``FusionError(kind)`` with the
reference's variants ``Pipeline`` / ``ParamsTooLarge`` / ``Hip``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

from . import _ffi
from ._ffi import lib
from .hip import Stream
from .tensor import Tensor


class FusionError(ValueError):
    def __init__(self, kind: str, message: str):
        super().__init__(message)
        self.kind = kind


def _check(rc: int) -> None:
    if rc == _ffi.KH_OK:
        return
    msg = _ffi.last_error()
    kind = {_ffi.KH_ERR_INVALID_ARG: "Pipeline", _ffi.KH_ERR_TOO_LARGE: "ParamsTooLarge",
            _ffi.KH_ERR_SLICE_TOO_SMALL: "Pipeline"}.get(rc, "Hip")
    raise FusionError(kind, msg)


class ReadU8RgbBilinear:
    """Source: bilinear-resample interleaved u8 RGB on the half-pixel grid -> f32 RGB in [0, 255]."""

    def __init__(self, src_w: int, src_h: int, dst_w: int, dst_h: int):
        self.src_w, self.src_h, self.dst_w, self.dst_h = int(src_w), int(src_h), int(dst_w), int(dst_h)

    def name(self) -> str:
        return "read_u8rgb_bilinear"

    def _stage(self) -> _ffi.FusedStage:
        s = _ffi.FusedStage()
        s.kind = _ffi.KH_FUSE_READ_U8RGB_BILINEAR
        s.u[0], s.u[1], s.u[2], s.u[3] = self.src_w, self.src_h, self.dst_w, self.dst_h
        return s


class Normalize:
    """Map: per-channel ``v * scale + bias``."""

    def __init__(self, scale: Sequence[float], bias: Sequence[float]):
        self.scale, self.bias = [float(v) for v in scale], [float(v) for v in bias]
        if len(self.scale) != 3 or len(self.bias) != 3:
            raise FusionError("Pipeline", "normalize takes 3 scales and 3 biases")

    def name(self) -> str:
        return "normalize"

    def _stage(self) -> _ffi.FusedStage:
        s = _ffi.FusedStage()
        s.kind = _ffi.KH_FUSE_NORMALIZE
        for i in range(3):
            s.f[i], s.f[3 + i] = self.scale[i], self.bias[i]
        return s


class _Plain:
    _kind = 0
    _name = ""

    def name(self) -> str:
        return self._name

    def _stage(self) -> _ffi.FusedStage:
        s = _ffi.FusedStage()
        s.kind = self._kind
        return s


class RgbToGray(_Plain):
    """Map: BT.601 gray replicated to all three lanes."""
    _kind, _name = _ffi.KH_FUSE_RGB_TO_GRAY, "rgb_to_gray"


class WriteChwF32(_Plain):
    """Sink: three f32 planes ``[3, dst_h, dst_w]``."""
    _kind, _name = _ffi.KH_FUSE_WRITE_CHW_F32, "write_chw_f32"


class WriteC1F32(_Plain):
    """Sink: one f32 plane from lane x."""
    _kind, _name = _ffi.KH_FUSE_WRITE_C1_F32, "write_c1_f32"


class FusedPipeline:
    def __init__(self, handle: int, dst_w: int, dst_h: int, batch: int, names):
        self._h, self.dst_w, self.dst_h, self.batch, self.stage_names = handle, dst_w, dst_h, batch, names

    @staticmethod
    def _build(stages, dst_w: int, dst_h: int, batch: int, out_elems_per_image: int) -> "FusedPipeline":
        arr = (_ffi.FusedStage * max(len(stages), 1))(*[s._stage() for s in stages])
        h = C.c_void_p()
        _check(lib.kh_fused_pipeline_build(C.cast(arr, C.c_void_p), len(stages), dst_w, dst_h, batch, out_elems_per_image,
                                           C.byref(h)))
        return FusedPipeline(h.value, dst_w, dst_h, batch, [s.name() for s in stages])

    @staticmethod
    def build(stages, dst_w: int, dst_h: int) -> "FusedPipeline":
        return FusedPipeline._build(stages, dst_w, dst_h, 1, 0)

    @staticmethod
    def build_batched(stages, dst_w: int, dst_h: int, batch: int, out_elems_per_image: int) -> "FusedPipeline":
        if batch == 0:
            raise FusionError("Pipeline", "batch must be >= 1")
        return FusedPipeline._build(stages, dst_w, dst_h, batch, out_elems_per_image)

    def generated_source(self) -> str:
        """The stage program the kernel walks (the reference returns its generated CUDA source)."""
        buf = C.create_string_buffer(4096)
        lib.kh_fused_pipeline_describe(self._h, buf, len(buf))
        return buf.value.decode()

    def launch_batched(self, stream: Stream, srcs: Sequence[Tensor], dst: Tensor) -> None:
        for t in list(srcs) + [dst]:
            if not t.is_device:
                raise FusionError("Pipeline", "fused pipelines take device-resident tensors")
        ptrs = (C.c_void_p * max(len(srcs), 1))(*[t.data_ptr for t in srcs])
        nbytes = min((t.nbytes for t in srcs), default=0)
        _check(lib.kh_fused_pipeline_launch(self._h, stream.cuda_stream_ptr, ptrs, len(srcs), nbytes, dst.data_ptr,
                                            dst.nbytes // 4))

    def launch(self, stream: Stream, src: Tensor, dst: Tensor) -> None:
        self.launch_batched(stream, [src], dst)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.kh_fused_pipeline_destroy(h)


def _bind_pipeline_methods() -> None:
    from .hip import on_operand_device   # launch with the stream's device current (the first device operand: the stream argument)
    for name in ("launch_batched", "launch"):
        setattr(FusedPipeline, name, on_operand_device(getattr(FusedPipeline, name)))


_bind_pipeline_methods()
