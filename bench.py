#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X imgproc backend (driver contract in the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = one pass of the hot path over one resident batch of synthetic frames.  The default
workload is BASELINE.json configs[2] — the north star: fused NV12 -> normalized CHW f32,
1920x1080, batch 1024 per GPU, Stretch at scale 1 with ImageNet mean/std.  Inputs are generated
once (LCG bytes, frame k = base pattern shifted by 31*k, like the reference's batch test
crates/kornia-imgproc/src/preprocess.rs:1869) and are resident in HBM before the timed region.

Weak scaling: each rank owns its own 1024-frame batch on its own GPU; there is no collective on
the data path (frames are independent units — SURVEY.md §8e).  torch.distributed (RCCL) is used
only for the start/stop barrier and the max-over-ranks of the elapsed time.

The JSON line additionally carries
  roofline     — algorithmic bytes per launch / mean launch duration (HIP events recorded on the
                 launch stream around every timed launch), against the 8 TB/s HBM3E peak;
  cpu_baseline — the CPU oracle (a C restatement of the reference, `kind: "port"`) timed on this
                 box's host cores over a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "kornia-rs_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def lcg_bytes(n: int) -> np.ndarray:
    """pattern_u8 of the reference's GPU tests (cuda/color/mod.rs:303-317), vectorised: the LCG
    state after k steps is A_k*s0 + C_k (mod 2^32); blocks are extended by doubling."""
    prefix = np.array([0, 255, 255, 0, 0, 0, 255, 255, 255, 1, 254, 128, 128, 128, 64], np.uint8)
    if n <= 15:
        return prefix[:n].copy()
    m = n - 15
    mask = np.uint64(0xFFFFFFFF)
    a, c = np.uint64(1664525), np.uint64(1013904223)
    states = np.empty(m, np.uint64)
    states[0] = (np.uint64(0x12345678) * a + c) & mask
    have, A, Cc = 1, a, c  # (A, Cc) = `have`-step jump
    while have < m:
        take = min(have, m - have)
        states[have:have + take] = (states[:take] * A + Cc) & mask
        Cc = (A * Cc + Cc) & mask
        A = (A * A) & mask
        have += take
    return np.concatenate([prefix, (states >> np.uint64(24)).astype(np.uint8)])


class Workload:
    name = ""
    units_per_step = 0           # Mpixels (source pixels) processed per step per GPU
    alg_bytes_per_launch = 0     # algorithmic HBM bytes of the dominant kernel per launch
    kernel = ""
    dtype = ""

    def setup(self, stream): ...
    def step(self): ...
    def describe(self) -> dict: ...
    def cpu_baseline(self) -> dict: ...


class NorthStarNV12(Workload):
    """configs[2]: fused NV12 -> normalized CHW f32, 1920x1080, batch 1024 on 1 GPU — and its secondaries: the same frames
    letterboxed to `out` x `out` (640: the reference's own preprocess_nv12_640; 608: a geometry whose taps do NOT fall on whole
    pixels, so the general four-tap kernel is in the record), and the YUYV source format (fmt="yuyv": the reference's
    published 1080p -> 640 fused configuration, docs/benchmark-cuda-color-conversions.md:104)."""

    name = "nv12_1080p_to_chw_f32_b1024"
    W, H = 1920, 1080
    kernel = "preprocess_nv12_identity"
    dtype = "f32"

    def __init__(self, batch: int = 1024, out: int = 0, sampling: str = "bilinear", fmt: str = "nv12"):
        self.N = batch
        self.sampling = sampling
        self.fmt = fmt
        self.out = out  # 0: same-size (north star); else letterbox to out x out (secondary rows)
        px = self.W * self.H
        self.frame_bytes = px * 3 // 2 if fmt == "nv12" else px * 2
        self.src_bytes_per_px = 1.5 if fmt == "nv12" else 2.0
        self.units_per_step = self.N * px / 1e6
        if out == 0:
            # 1.5 B/px read + 12 B/px written = 27 993 600 B per frame
            self.alg_bytes_per_launch = self.N * (self.frame_bytes + 12 * px)
        else:
            self.name = f"{fmt}_1080p_to_chw_f32_letterbox{out}{'' if sampling == 'bilinear' else '_' + sampling}_b{batch}"
            self.kernel = "preprocess_generic"
            self.alg_bytes_per_launch = 0  # priced in setup(): it depends on the variant the library launches for this geometry

    def _price_letterbox(self):
        """Algorithmic bytes of a letterboxed launch = every destination float written + the source taps the LAUNCHED variant
        needs for the ACTIVE destination pixels only (padding pixels read nothing): one tap when the library reports the
        on-grid form (every source coordinate a whole number: `generic_bilinear_on_grid`) or nearest sampling, four for the
        general bilinear kernel, 36 for Lanczos-3 — each tap `src_bytes_per_px` (1.5 B NV12, 2 B YUYV), SURVEY.md §8(d).
        Active pixels are counted with the kernel's own f32 coordinate expression (plan_pixel, P/preprocess.rs:437-448)."""
        from kornia_rs import _ffi
        f = self.pre.source_format
        p = self.pre._params(self.W, self.H, f.pitch(self.W), f.bpp, f.fmt_code, self.out, self.out, self.N, self.frame_bytes, False, False)
        self.variant = _ffi.lib.kh_preprocess_variant(C.byref(p)).decode()
        f32 = np.float32

        def active(n, pad, scale, length):
            s = (np.arange(n, dtype=f32) - f32(pad)) / f32(scale)
            return int(((s >= 0) & (s < f32(length))).sum())

        ax, ay = active(self.out, p.pad_x, p.scale_x, self.W), active(self.out, p.pad_y, p.scale_y, self.H)
        taps = 36 if self.sampling == "lanczos" else (1 if (self.variant.endswith("on_grid") or self.sampling == "nearest") else 4)
        self.active_px, self.taps = ax * ay, taps
        self.kernel = f"preprocess_generic ({self.variant})"
        self.alg_bytes_per_launch = int(self.N * (self.out * self.out * 12 + ax * ay * taps * self.src_bytes_per_px))

    def setup(self, stream):
        from kornia_rs import Preprocessor, Tensor
        from kornia_rs.hip import DeviceBuffer, lib, check
        self.stream = stream
        base = lcg_bytes(self.frame_bytes + 31 * self.N)
        dbase = DeviceBuffer.from_numpy(base, stream)
        self.src = DeviceBuffer(self.frame_bytes * self.N, stream, zeroed=False)
        for k in range(self.N):  # frame k = base[31k : 31k + frame_bytes], assembled on device
            check(lib.kh_memcpy_d2d_async(self.src.ptr + k * self.frame_bytes, dbase.ptr + 31 * k,
                                          self.frame_bytes, stream.cuda_stream_ptr))
        stream.synchronize()
        self.base = base
        oh, ow = (self.H, self.W) if self.out == 0 else (self.out, self.out)
        self.dst = Tensor.uninit((self.N, 3, oh, ow), "float32", stream)
        self.pre = Preprocessor(mode="stretch" if self.out == 0 else "letterbox", format=self.fmt,
                                sampling=self.sampling, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                                stream=stream)
        if self.out:
            self._price_letterbox()

    def step(self):
        self.pre.run_raw_batch(self.src, self.W, self.H, self.dst, frame_stride=self.frame_bytes)

    def describe(self):
        d = {"workload": self.name, "op": f"Preprocessor.run_raw_batch (fused {self.fmt.upper()} decode + "
             f"{self.sampling} + ImageNet normalize + HWC->CHW)", "src": f"1920x1080 {self.fmt.upper()}",
             "dst": f"[{self.N},3,{self.dst.shape[2]},{self.dst.shape[3]}] f32",
             "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}
        if self.out:
            d.update(variant=self.variant, active_dst_px=self.active_px, taps_per_active_px=self.taps,
                     pricing="dst floats written + taps x source bytes per ACTIVE dst pixel of the launched variant")
        return d

    def cpu_baseline(self):
        """Chained rgb_from_nv12 / rgb_from_yuyv (Q20) -> direct bilinear/normalise/CHW, the comparator BASELINE.md
        §3 names (the reference has no CPU fused camera-format path).  Bounded to ~10-20 s."""
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_ffi as O  # test infrastructure: used here ONLY as the timed CPU baseline
        threads = O.ko.ko_max_threads()
        oh, ow = (self.H, self.W) if self.out == 0 else (self.out, self.out)
        decode = O.rgb_from_nv12 if self.fmt == "nv12" else O.rgb_from_yuyv
        frames, t0, budget = 0, time.perf_counter(), 12.0 * CPU_BUDGET_SCALE
        while True:
            raw = self.base[31 * (frames % self.N): 31 * (frames % self.N) + self.frame_bytes]
            rgb = decode(raw, self.W, self.H)
            O.preprocess(rgb, self.W, self.H, ow, oh, fmt="rgb",
                         mode="stretch" if self.out == 0 else "letterbox", sampling=self.sampling,
                         mean=IMAGENET_MEAN, std=IMAGENET_STD)
            frames += 1
            dt = time.perf_counter() - t0
            if dt > budget or frames >= 16384:
                break
        return {"value": round(frames * self.W * self.H / 1e6 / dt, 2), "unit": "Mpixels/s",
                "cores": threads, "kind": "port",
                "sample": f"{frames} frames of the same 1080p {self.fmt.upper()} workload in {dt:.1f} s; C oracle "
                          "(faithful restatement of kornia-imgproc, not the upstream Rust binary), "
                          f"chained rgb_from_{self.fmt} -> {self.sampling}/normalize/CHW, OpenMP x{threads}"}


class NorthStarNV12List(NorthStarNV12):
    """The north star through the reference's OWN batch signature — `run_raw_batch(frames: &[&CudaSlice<u8>], ..)`
    (P/preprocess.rs:1258-1282): 1024 SEPARATELY ALLOCATED 1080p NV12 frame buffers (unequal spacing), one [1024, 3, 1080, 1920]
    destination.  The reference launches once per frame; here the frame bases travel in the kernel arguments, 256 per launch
    (kh_preprocess_to_chw_list).  Must sit within 2 % of the equally spaced headline."""

    def __init__(self, batch: int = 1024):
        super().__init__(batch, 0)
        self.cpu_twin = "nv12_chw"
        self.name = f"nv12_1080p_to_chw_f32_frame_list_b{batch}"
        self.kernel = "preprocess_nv12_identity_list"

    def setup(self, stream):
        from kornia_rs import Preprocessor, Tensor
        from kornia_rs.hip import DeviceBuffer, lib, check
        self.stream = stream
        self.base = lcg_bytes(self.frame_bytes + 31 * self.N)
        dbase = DeviceBuffer.from_numpy(self.base, stream)
        self.frames, self._spacers = [], []
        for k in range(self.N):   # every frame its own allocation; 0 / 2 / 4 MiB spacer allocations in between: the bases are not equally spaced
            if (5 * k) % 3:
                self._spacers.append(DeviceBuffer(((5 * k) % 3) << 21, stream, zeroed=False))
            self.frames.append(DeviceBuffer(self.frame_bytes, stream, zeroed=False))
        if self.N >= 3 and len({b.ptr - a.ptr for a, b in zip(self.frames, self.frames[1:])}) == 1:
            # (a pool that recycles freed blocks can still hand out an even run: exchange two buffers — equally spaced frames would
            # take the strided launch and this row would not measure the list)
            self.frames[1], self.frames[2] = self.frames[2], self.frames[1]
        for k, buf in enumerate(self.frames):
            check(lib.kh_memcpy_d2d_async(buf.ptr, dbase.ptr + 31 * k, self.frame_bytes, stream.cuda_stream_ptr))
        stream.synchronize()
        self.dst = Tensor.uninit((self.N, 3, self.H, self.W), "float32", stream)
        self.pre = Preprocessor(mode="stretch", format="nv12", sampling=self.sampling, mean=IMAGENET_MEAN, std=IMAGENET_STD, stream=stream)

    def step(self):
        self.pre.run_raw_batch(self.frames, self.W, self.H, self.dst)

    def describe(self):
        d = super().describe()
        d.update(op="Preprocessor.run_raw_batch([frame_0 .. frame_N-1]) — a LIST of separately allocated device buffers, the reference's "
                    "signature — -> kh_preprocess_to_chw_list (256 frame bases per launch)",
                 src=f"{self.N} separately allocated 1920x1080 NV12 buffers")
        return d


class NorthStarNV12F16(NorthStarNV12):
    """The north star's binary16 twin — `run_raw_batch_f16` (P/preprocess.rs:1234-1256): the same 1024 NV12 1080p frames into
    [1024, 3, 1080, 1920] f16 planes (decode and normalisation in int32 / f32, the reference's f32 -> f16 rounding at the store).  Since
    round 6 its own kernel (eight pixels per thread, three 16-byte stores of eight halves); it took the generic kernel before and ran
    slower than the f32 headline."""

    def __init__(self, batch: int = 1024):
        super().__init__(batch, 0)
        self.name = f"nv12_1080p_to_chw_f16_b{batch}"
        self.kernel = "preprocess_nv12_identity_f16"
        self.dtype = "f32 -> f16 store"
        self.alg_bytes_per_launch = self.N * (self.frame_bytes + 6 * self.W * self.H)   # 1.5 B/px read + 3 x 2 B/px written

    def setup(self, stream):
        from kornia_rs import Preprocessor, Tensor
        super().setup(stream)
        self.dst = Tensor.uninit((self.N, 3, self.H, self.W), "float16", stream)
        self.pre = Preprocessor(mode="stretch", format="nv12", sampling=self.sampling, f16=True, mean=IMAGENET_MEAN, std=IMAGENET_STD, stream=stream)

    def describe(self):
        d = super().describe()
        d.update(op="Preprocessor.run_raw_batch -> f16 planes (run_raw_batch_f16: fused NV12 decode + ImageNet normalize + HWC->CHW, f32 -> f16 at the store)",
                 dst=f"[{self.N},3,{self.H},{self.W}] f16")
        return d

    def cpu_baseline(self):
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_ffi as O  # test infrastructure: used here ONLY as the timed CPU baseline
        threads = O.ko.ko_max_threads()
        frames, t0, budget = 0, time.perf_counter(), 8.0 * CPU_BUDGET_SCALE
        while True:
            raw = self.base[31 * (frames % self.N): 31 * (frames % self.N) + self.frame_bytes]
            O.preprocess(O.rgb_from_nv12(raw, self.W, self.H), self.W, self.H, self.W, self.H, fmt="rgb", mode="stretch", sampling=self.sampling,
                         f16=True, mean=IMAGENET_MEAN, std=IMAGENET_STD)
            frames += 1
            dt = time.perf_counter() - t0
            if dt > budget or frames >= 16384:
                break
        return {"value": round(frames * self.W * self.H / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": threads, "kind": "port",
                "sample": f"{frames} frames in {dt:.1f} s; C oracle, chained rgb_from_nv12 -> normalize / CHW / f16, OpenMP x{threads}"}


class LetterboxF16(NorthStarNV12):
    """The letterbox secondaries into binary16 planes (run_raw_batch_f16; the shape a half-precision detector consumes): opt-in rows.
    Since round 6 through the flattened-quad kernel with one 8-byte store of four halves per plane (the per-pixel kernel with 2-byte
    stores before: no faster than f32 for half the bytes)."""

    def __init__(self, batch: int = 1024, out: int = 640, fmt: str = "nv12"):
        super().__init__(batch, out, "bilinear", fmt)
        self.name = f"{fmt}_1080p_to_chw_f16_letterbox{out}_b{batch}"
        self.dtype = "f32 -> f16 store"

    def setup(self, stream):
        from kornia_rs import Preprocessor, Tensor
        super().setup(stream)
        self.dst = Tensor.uninit((self.N, 3, self.out, self.out), "float16", stream)
        self.pre = Preprocessor(mode="letterbox", format=self.fmt, sampling=self.sampling, f16=True, mean=IMAGENET_MEAN, std=IMAGENET_STD, stream=stream)
        # priced like the f32 row, with 2 bytes per destination value
        self.alg_bytes_per_launch = int(self.N * (self.out * self.out * 6 + self.active_px * self.taps * self.src_bytes_per_px))

    def describe(self):
        d = super().describe()
        d.update(dst=f"[{self.N},3,{self.out},{self.out}] f16")
        return d

    def cpu_baseline(self):
        return {"value": None, "unit": "Mpixels/s", "cores": None, "kind": "port", "sample": "see the f32 row of the same geometry"}


class H2DPreprocess1080p(NorthStarNV12):
    """SURVEY.md §8(f)4, the capture side of the path: HOST NV12 frames -> page-locked capture buffers -> H2D -> fused preprocess,
    through Preprocessor.run_host_batch (the two-deep upload ring of kornia_rs/preprocess.py::_Staging on a copy stream; the
    reference's staging is kornia-py/src/cuda_ext/mod.rs:647-731, its H2D / kernel / D2H harness
    crates/kornia-imgproc/benches/bench_cuda_imgproc.rs:87-139).  One step = `batch` 1080p frames that start in host memory.
    The path is bound by the host link, not by HBM: beside the usual record, `roofline` carries the event-timed breakdown —
    the pinned H2D of one batch alone, the kernel alone on resident frames, the end-to-end step in steady state, how much of the
    kernel the ring hides behind the next upload, and end-to-end as a fraction of the pinned-H2D rate measured in this process."""

    def __init__(self, batch: int = 64, pageable: bool = True):
        super().__init__(batch, 0)
        # pageable = True (the default row): the reference's staging contract — host frames are COPIED into the upload ring's page-locked
        # slot and may be reused as soon as the call returns (kornia-py/src/cuda_ext/mod.rs:647-745).  pageable = False is the opt-in
        # `zero_copy=True` path: frames that already live in page-locked capture buffers are DMA'd in place and stay the caller's hazard
        # until wait_uploads() — a labelled extra, not the default.
        self.pageable = pageable
        self.name = f"nv12_h2d_preprocess_1080p{'' if pageable else '_zero_copy'}_b{batch}"
        self.kernel = ("host memcpy into the ring + " if pageable else "") + "hipMemcpyAsync(H2D, pinned) + preprocess_nv12_identity"

    RING = 4   # distinct host batches rotated through (>= 4; a 199 MB batch fits the 256 MB Infinity Cache)

    def setup(self, stream):
        from kornia_rs import Preprocessor, Tensor
        from kornia_rs.hip import PinnedBuffer
        self.stream = stream
        fb = self.frame_bytes
        self.base = lcg_bytes(fb + 31 * self.N * self.RING)
        self.cap = PinnedBuffer(self.RING * self.N * fb)
        view = self.cap.view()
        self.host_frames = []
        for b in range(self.RING):
            rows = []
            for k in range(self.N):
                i = b * self.N + k
                view[i * fb:(i + 1) * fb] = self.base[31 * i: 31 * i + fb]
                rows.append(view[i * fb:(i + 1) * fb] if not self.pageable else self.base[31 * i: 31 * i + fb].copy())
            self.host_frames.append(rows)
        self.dst = Tensor.uninit((self.N, 3, self.H, self.W), "float32", stream)
        self.pre = Preprocessor(mode="stretch", format="nv12", mean=IMAGENET_MEAN, std=IMAGENET_STD, stream=stream)
        self.turn = 0
        for _ in range(2 * self.RING):   # the FIRST DMA out of a page-locked region is several times slower than the following ones (r04w:
            self.step()                  # one such step inside ten timed ones put the mean at 5.6 ms against a 3.49 ms minimum): every
        stream.synchronize()             # capture buffer and both ring slots are touched once here, whatever --warmup says
        self.turn = 0

    def step(self):
        self.pre.run_host_batch(self.host_frames[self.turn % self.RING], self.W, self.H, self.dst, zero_copy=not self.pageable)
        self.turn += 1

    def roofline_extra(self, mean_step_s):
        """H2D alone / kernel alone, timed with events in this process after the headline steps."""
        from kornia_rs import hip
        from kornia_rs.hip import DeviceBuffer, lib, check
        st = self.stream
        fb, n = self.frame_bytes, self.N
        dev = DeviceBuffer(n * fb, st, zeroed=False)

        def timed(fn, reps=8):
            fn(); st.synchronize()
            e0, e1 = hip.Event(timing=True), hip.Event(timing=True)
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st); st.synchronize()
            return e0.elapsed_ms(e1) / reps

        h2d_ms = timed(lambda: check(lib.kh_memcpy_h2d_async(dev.ptr, self.cap.ptr, n * fb, st.cuda_stream_ptr)))
        ker_ms = timed(lambda: self.pre.run_raw_batch(dev, self.W, self.H, self.dst, frame_stride=fb))
        e2e_ms = mean_step_s * 1e3
        h2d_ms, ker_ms = max(h2d_ms, 1e-9), max(ker_ms, 1e-9)   # (the host simulator's events have no resolution)
        gbs = n * fb / h2d_ms / 1e6
        return {"bound_note": "host link (PCIe H2D), not HBM: frac is reported against 8 TB/s only for uniformity",
                "h2d_only_ms": round(h2d_ms, 4), "h2d_pinned_GBps": round(gbs, 2), "kernel_only_ms": round(ker_ms, 4),
                "end_to_end_ms": round(e2e_ms, 4), "serial_sum_ms": round(h2d_ms + ker_ms, 4),
                "hidden_by_overlap_ms": round(h2d_ms + ker_ms - e2e_ms, 4),
                "end_to_end_frac_of_pinned_h2d": round(h2d_ms / max(e2e_ms, 1e-9), 4),
                "source": "pageable numpy frames (host memcpy into the ring's pinned slot, then DMA)" if self.pageable
                          else "page-locked capture buffers, DMA'd in place (run_host_batch(..., zero_copy=True): opt-in)"}

    def describe(self):
        d = super().describe()
        d.update(op="Preprocessor.run_host_batch: host frames -> upload ring (copy stream) -> fused NV12 decode + normalize + CHW",
                 src="1920x1080 NV12 in HOST memory (" + ("pageable" if self.pageable else "page-locked capture buffers") + ")")
        return d


class F32Images(Workload):
    """Shared setup for the f32 HWC configs: `batch` images assembled on device from one LCG
    base pattern (image k = base shifted by 31*k floats, values u8/255)."""

    dtype = "f32"

    def _make_src(self, stream, w, h, c, batch):
        from kornia_rs.hip import DeviceBuffer, lib, check
        n = w * h * c
        base = (lcg_bytes(n + 31 * batch).astype(np.float32) / np.float32(255.0))
        dbase = DeviceBuffer.from_numpy(base, stream)
        src = DeviceBuffer(n * 4 * batch, stream, zeroed=False)
        for k in range(batch):
            check(lib.kh_memcpy_d2d_async(src.ptr + k * n * 4, dbase.ptr + 31 * k * 4, n * 4, stream.cuda_stream_ptr))
        stream.synchronize()
        self.base = base
        return src

    def _oracle(self):
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_ffi as O  # test infrastructure: used here ONLY as the timed CPU baseline
        return O

    def _make_images(self, stream, w, h, c, batch, fill=True):
        """`batch` SEPARATELY ALLOCATED device Images (the operands the reference's per-image operators are handed), image k = the
        base pattern shifted by 31 k floats like `_make_src`; a spacer allocation of varying size between consecutive images keeps
        their bases from being equally spaced."""
        from kornia_rs import Image
        from kornia_rs.hip import DeviceBuffer, lib, check
        n = w * h * c
        dbase = None
        if fill:
            self.base = (lcg_bytes(n + 31 * batch).astype(np.float32) / np.float32(255.0))
            dbase = DeviceBuffer.from_numpy(self.base, stream)
        imgs, spacers = [], []
        for k in range(batch):
            # (the stream-ordered pool rounds allocations up: only gaps of whole megabytes are sure to differ — measured in round 6,
            # where 1024 frames padded by k * 256 B came out equally spaced)
            if (5 * k) % 3:
                spacers.append(DeviceBuffer(((5 * k) % 3) << 21, stream, zeroed=False))
            im = Image.uninit(w, h, c, "float32", stream)
            if fill:
                check(lib.kh_memcpy_d2d_async(im.data_ptr, dbase.ptr + 31 * k * 4, n * 4, stream.cuda_stream_ptr))
            imgs.append(im)
        stream.synchronize()
        return ImageList(imgs, spacers)   # (imgproc.*_batch always launch through the pointer list, whatever the spacing)


class ImageList(list):
    """N separately allocated device Images standing where a packed batch buffer stands in the strided workloads."""

    def __init__(self, images, keep=()):
        super().__init__(images)
        self._keep = list(keep)

    def to_numpy(self, dtype, shape):
        return np.stack([im.numpy() for im in self]).astype(dtype, copy=False).reshape(shape)


class ResizeBilinear(F32Images):
    """configs[1]: resize bilinear 1920x1080 -> 224x224 f32x3, batch 256."""

    name, kernel = "resize_bilinear_1080p_to_224_f32_b256", "resize_rows_bilinear_kernel<3>"
    SW, SH, DW, DH, C = 1920, 1080, 224, 224, 3

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.SW * self.SH / 1e6
        # taps actually required = 224*224*(4 taps*12 B + 12 B) = 3 010 560 B / image
        self.alg_bytes_per_launch = self.N * self.DW * self.DH * (4 * 12 + 12)

    ROTATE = 4   # >= 4 rotating buffers where a step's set is small — the 154 MB output fits the 256 MB Infinity Cache

    def setup(self, stream):
        from kornia_rs.hip import DeviceBuffer
        self.stream = stream
        self.src = self._make_src(stream, self.SW, self.SH, self.C, self.N)
        self.dsts = [DeviceBuffer(self.N * self.DW * self.DH * self.C * 4, stream, zeroed=False) for _ in range(self.ROTATE)]
        self.dst, self.turn = self.dsts[0], 0

    def _next_dst(self):
        self.dst = self.dsts[self.turn % len(self.dsts)]   # `dst` = the buffer the LAST step wrote (what the parity test reads back)
        self.turn += 1
        return self.dst

    def step(self):
        from kornia_rs._ffi import lib, check
        check(lib.kh_resize_f32(self.stream.cuda_stream_ptr, self.src.ptr, self._next_dst().ptr, self.SW, self.SH, self.DW,
                                self.DH, self.C, 1, self.N, self.SW * self.SH * self.C, self.DW * self.DH * self.C))

    def floor_bytes(self):
        """What the memory system must move at its own granularity: every 128-byte line of the source that holds a byte of some tap,
        once per image (lines of consecutive taps overlap), + the destination.  Evaluated with the kernel's own f32 coordinate
        expression (half-pixel a * i + b, clamped; P/resize/mod.rs:161-176)."""
        f32 = np.float32

        def axis(src_len, dst_len):
            a = f32(src_len) / f32(dst_len)
            b = f32(0.5) * a - f32(0.5)
            s = np.clip(a * np.arange(dst_len, dtype=f32) + b, f32(0), f32(src_len - 1))
            i0 = s.astype(np.int64)
            return i0, np.minimum(i0 + 1, src_len - 1)

        x0, x1 = axis(self.SW, self.DW)
        y0, y1 = axis(self.SH, self.DH)
        px = self.C * 4
        lines = set()
        for xa, xb in zip(x0, x1):
            for x in (int(xa), int(xb)):
                lines.update(range(x * px // 128, (x * px + px - 1) // 128 + 1))
        # a source row starts at a multiple of SW * C * 4 = 23 040 = 180 * 128 bytes: the line pattern is the same for every row
        rows = len(set(y0.tolist()) | set(y1.tolist()))
        return int(self.N * (rows * len(lines) * 128 + self.DW * self.DH * px))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::resize (bilinear, half-pixel)", "src": "1920x1080x3 f32",
                "dst": "224x224x3 f32", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective",
                "source_Mpx_per_step": round(self.units_per_step, 1), "output_Mpx_per_step": round(self.N * 224 * 224 / 1e6, 2)}

    def cpu_baseline(self):
        O = self._oracle()
        threads = O.ko.ko_max_threads()
        n = self.SW * self.SH * self.C
        frames, t0 = 0, time.perf_counter()
        while True:
            O.resize(self.base[31 * (frames % self.N): 31 * (frames % self.N) + n].reshape(self.SH, self.SW, self.C), self.DW, self.DH)
            frames += 1
            dt = time.perf_counter() - t0
            if dt > 10.0 * CPU_BUDGET_SCALE or frames >= 256:
                break
        return {"value": round(frames * self.SW * self.SH / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": threads,
                "kind": "port", "sample": f"{frames} images in {dt:.1f} s; C oracle (restatement of kornia-imgproc "
                f"resize, not the upstream Rust binary), OpenMP x{threads} over output rows"}


class ResizeBilinearApi(ResizeBilinear):
    """configs[1] THROUGH THE OPERATOR API with separately allocated operands: 256 independent 1080p `Image`s
    into 256 independent 224 x 224 `Image`s —
      how = "eager": 256 calls of `imgproc.resize(&Image, &mut Image)` per step, the reference's own signature
                     (P/resize/mod.rs:114-132): one launch per image, launch-bound (a 2 us kernel behind a Python call);
      how = "graph": the same 256 calls captured once into a `hip.Graph` and replayed per step — the reference's mechanism for
                     amortising per-image launches (kornia-py/src/cuda_ext/mod.rs:1684-1790);
      how = "list":  `imgproc.resize_batch(images, outs=...)` -> `kh_resize_f32_list`: the 256 (src, dst) bases in the kernel
                     arguments, two launches of 128 images."""

    def __init__(self, batch, how):
        super().__init__(batch)
        self.how = how
        self.cpu_twin = "resize_224"
        self.name = f"resize_bilinear_1080p_to_224_f32_api_{how}_b{batch}"

    def setup(self, stream):
        from kornia_rs import hip, imgproc
        self.stream = stream
        self.src = self._make_images(stream, self.SW, self.SH, self.C, self.N)
        self.dsts = [self._make_images(stream, self.DW, self.DH, self.C, self.N, fill=False) for _ in range(self.ROTATE)]
        self.dst, self.turn = self.dsts[0], 0
        if self.how == "list":   # the host keeps the pointer arrays beside its images (imgproc.ImageBatch), as a Rust host would a Vec<*const f32>
            self.src_b, self.dst_b = imgproc.ImageBatch(self.src), [imgproc.ImageBatch(d) for d in self.dsts]
        if self.how == "graph":
            def record(outs):
                for s_, d_ in zip(self.src, outs):
                    imgproc.resize(s_, None, "bilinear", out=d_)
            self.graphs = [hip.Graph.capture(lambda o=o: record(o), retain=[self.src, o], stream=stream) for o in self.dsts]

    def step(self):
        from kornia_rs import imgproc
        dst = self._next_dst()
        if self.how == "eager":
            for s_, d_ in zip(self.src, dst):
                imgproc.resize(s_, None, "bilinear", out=d_)
        elif self.how == "graph":
            self.graphs[(self.turn - 1) % len(self.graphs)].replay()
        else:
            imgproc.resize_batch(self.src_b, None, "bilinear", outs=self.dst_b[(self.turn - 1) % len(self.dst_b)])

    def describe(self):
        d = super().describe()
        d.update(workload=self.name, operands=f"{self.N} separately allocated source Images -> {self.N} separately allocated destination Images",
                 op={"eager": "imgproc.resize(src_k, out=dst_k) x N per step (one launch per image)",
                     "graph": "hip.Graph replay of N captured imgproc.resize calls",
                     "list": "imgproc.resize_batch(srcs, outs=dsts) -> kh_resize_f32_list (128 images per launch)"}[self.how])
        return d


class ResizeNormalizeF32(ResizeBilinear):
    """The reference's own fused launcher on the configs[1] shape: bilinear resize 1920x1080 -> 224x224 f32x3 fused with
    (px - mean) * (1 / std) (launch_resize_bilinear_normalize_cuda, bench_cuda_resize.rs:456), batch 256."""

    name, kernel = "resize_bilinear_normalize_1080p_to_224_f32_b256", "resize_normalize_kernel"

    def step(self):
        import ctypes as C
        from kornia_rs._ffi import lib, check
        from kornia_rs.hip import IMAGENET_MEAN, IMAGENET_STD
        check(lib.kh_resize_bilinear_normalize_f32(self.stream.cuda_stream_ptr, self.src.ptr, self._next_dst().ptr, self.SW, self.SH,
                                                   self.DW, self.DH, (C.c_float * 3)(*IMAGENET_MEAN), (C.c_float * 3)(*IMAGENET_STD),
                                                   0, self.N, self.SW * self.SH * self.C, self.DW * self.DH * self.C))

    def describe(self):
        d = super().describe()
        d.update(workload=self.name, op="cuda::resize::launch_resize_bilinear_normalize (half-pixel, ImageNet mean/std)")
        return d

    def cpu_baseline(self):
        O = self._oracle()
        from kornia_rs.hip import IMAGENET_MEAN, IMAGENET_STD
        threads, n = O.ko.ko_max_threads(), self.SW * self.SH * self.C
        frames, t0 = 0, time.perf_counter()
        while True:
            O.resize_bilinear_normalize(self.base[31 * (frames % self.N): 31 * (frames % self.N) + n].reshape(self.SH, self.SW, self.C), self.DW, self.DH,
                                        IMAGENET_MEAN, IMAGENET_STD)
            frames += 1
            dt = time.perf_counter() - t0
            if dt > 10.0 * CPU_BUDGET_SCALE or frames >= 256:
                break
        return {"value": round(frames * self.SW * self.SH / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                "sample": f"{frames} images in {dt:.1f} s; C restatement of the fused launcher, single thread"}


class Gaussian4K(F32Images):
    """configs[3]: separable gaussian_blur 7x7 sigma 1.5 f32x3, 3840x2160, batch 256 (LDS stencil)."""

    name, kernel = "gaussian_blur_7x7_4k_f32_b256", "sep_roll_kernel<7,false>"
    W, H, C = 3840, 2160, 3

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * 2 * self.W * self.H * self.C * 4  # 1R + 1W = 199 065 600 B / image

    def setup(self, stream):
        from kornia_rs.hip import DeviceBuffer
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * self.W * self.H * self.C * 4, stream, zeroed=False)

    def step(self):
        from kornia_rs._ffi import lib, check
        n = self.W * self.H * self.C
        check(lib.kh_gaussian_blur_f32(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H, self.C,
                                       7, 7, 1.5, 1.5, self.N, n, n))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::filter::gaussian_blur (7,7) sigma (1.5,1.5), fused H+V in LDS",
                "src": "3840x2160x3 f32", "dst": "same", "batch_per_gpu": self.N,
                "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        O = self._oracle()
        n = self.W * self.H * self.C
        img = self.base[:n].reshape(self.H, self.W, self.C)
        O.ko.ko_set_threads(1)  # the reference's separable_filter is single-threaded (separable_filter.rs:87)
        t0 = time.perf_counter()
        O.gaussian_blur(img, (7, 7), (1.5, 1.5))
        dt1 = time.perf_counter() - t0
        O.ko.ko_set_threads(CPU_TEAM["threads"] if CPU_TEAM else 0)
        threads = O.ko.ko_max_threads()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 5.0 * CPU_BUDGET_SCALE and reps < 16:
            O.gaussian_blur(img, (7, 7), (1.5, 1.5))
            reps += 1
        dtn = (time.perf_counter() - t0) / max(reps, 1)
        return {"value": round(self.W * self.H / 1e6 / dt1, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                "sample": f"1 image, single thread as in the reference ({dt1:.2f} s); beyond-reference OpenMP x{threads}: "
                          f"{self.W * self.H / 1e6 / dtn:.1f} Mpixels/s; C oracle, not the upstream Rust binary"}


class Gaussian4KList(Gaussian4K):
    """configs[3] through `imgproc.gaussian_blur_batch` with 256 separately allocated 4K source and destination Images
    (kh_gaussian_blur_f32_list: two launches of 128 images)."""

    def __init__(self, batch):
        super().__init__(batch)
        self.cpu_twin = "gaussian_4k"
        self.name = f"gaussian_blur_7x7_4k_f32_api_list_b{batch}"

    def setup(self, stream):
        self.stream = stream
        from kornia_rs import imgproc
        self.src = self._make_images(stream, self.W, self.H, self.C, self.N)
        self.dst = self._make_images(stream, self.W, self.H, self.C, self.N, fill=False)
        self.src_b, self.dst_b = imgproc.ImageBatch(self.src), imgproc.ImageBatch(self.dst)

    def step(self):
        from kornia_rs import imgproc
        imgproc.gaussian_blur_batch(self.src_b, (7, 7), (1.5, 1.5), outs=self.dst_b)

    def describe(self):
        d = super().describe()
        d.update(workload=self.name, op="imgproc.gaussian_blur_batch(srcs, (7,7), (1.5,1.5), outs=dsts) -> kh_gaussian_blur_f32_list",
                 operands=f"{self.N} separately allocated Images each side")
        return d


class UndistortWarp4K(F32Images):
    """configs[4] per-GPU share: remap (Brown-Conrady maps, Oak-D parameters scaled to 4K) then
    warp_perspective (projective H), bilinear, f32x3 3840x2160, batch 256 per GPU."""

    name, kernel = "undistort_remap_then_warp_perspective_4k_f32_b256", "remap_kernel<3,bilinear>+warp_perspective_px_kernel<3,bilinear,2>"
    W, H, C = 3840, 2160, 3
    # examples/undistort_image/src/main.rs:30-50 (1280x800 calibration), intrinsics scaled by 3840/1280, 2160/800
    INTR = (577.48583984375 * 3.0, 652.8748779296875 * 3.0, 577.48583984375 * 2.7, 386.1428833007812 * 2.7)
    DIST = (1.7547749280929563, 0.0097926277667284, -0.027250492945313457, 2.1092164516448975, 0.462927520275116,
            -0.08215277642011642, -0.00005535508171073161, 0.00003768636770639569)

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        img = self.W * self.H * self.C * 4
        # SURVEY.md §8(d)'s contract figure: 464 486 400 B / image (two passes, every tap, the maps once per image).  The kernels need
        # less — out-of-bounds destination pixels read nothing, and remap reads the two maps once per FOUR images — so `frac` is priced
        # on what they need (`_price`, set in setup) and the contract figure rides along as `survey_bytes_per_launch`.
        self.survey_bytes_per_launch = self.N * (4 * img + 2 * self.W * self.H * 4)
        self.alg_bytes_per_launch = self.survey_bytes_per_launch
        w, h = float(self.W), float(self.H)
        self.hm = [1.03, 0.05, -3.0 * w / 129.0, -0.02, 0.97, 4.0 * h / 97.0, 2.0 / (h * w), 1.5 / (w * h), 1.0]

    def setup(self, stream):
        import ctypes as C
        from kornia_rs.hip import DeviceBuffer
        from kornia_rs._ffi import lib, check
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        n = self.N * self.W * self.H * self.C * 4
        self.tmp = DeviceBuffer(n, stream, zeroed=False)
        self.dst = DeviceBuffer(n, stream, zeroed=False)
        self.mx = DeviceBuffer(self.W * self.H * 4, stream, zeroed=False)
        self.my = DeviceBuffer(self.W * self.H * 4, stream, zeroed=False)
        check(lib.kh_correction_map_polynomial_f32(stream.cuda_stream_ptr, self.mx.ptr, self.my.ptr, self.W, self.H,
                                                   (C.c_double * 4)(*self.INTR), (C.c_double * 8)(*self.DIST)))
        self.hptr = (C.c_float * 9)(*self.hm)
        self._price(stream, self.mx.ptr, self.my.ptr)

    def _price(self, stream, mx_ptr, my_ptr):
        """Needed bytes per image: both destinations written (2 x 99.5 MB); per pass the DISTINCT source pixels among the four taps of
        the in-bounds destination pixels, 12 B each (an undistortion zooms in: its taps cover only part of the source; out-of-bounds
        destination pixels read nothing); the maps, 66 MB, once per kRemapNB = 4 images.  Taps: the maps read back from the device; the
        homography evaluated in f32 like the kernel.  The counted HBM traffic (profiles/pmc_traffic.json) cannot be below this at line
        granularity, so frac <= the traffic-based rate."""
        from kornia_rs import hip
        f32 = np.float32
        n = self.W * self.H
        mx, my = np.empty(n, f32), np.empty(n, f32)
        hip.d2h(mx, mx_ptr, stream)
        hip.d2h(my, my_ptr, stream)

        def touched(u, v):
            inside = (u >= 0) & (u < f32(self.W)) & (v >= 0) & (v < f32(self.H))
            iu, iv = u[inside].astype(np.int64), v[inside].astype(np.int64)
            iu1, iv1 = np.minimum(iu + 1, self.W - 1), np.minimum(iv + 1, self.H - 1)
            seen = np.zeros(n, np.bool_)
            for yy in (iv, iv1):
                for xx in (iu, iu1):
                    seen[yy * self.W + xx] = True
            return float(inside.mean()), int(seen.sum())

        in_remap, px_remap = touched(mx, my)
        inv = np.linalg.inv(np.array(self.hm, np.float64).reshape(3, 3)).astype(f32).reshape(-1)
        ys, xs = np.mgrid[0:self.H, 0:self.W].astype(f32)
        wv = inv[6] * xs + inv[7] * ys + inv[8]
        in_warp, px_warp = touched(((inv[0] * xs + inv[1] * ys + inv[2]) / wv).reshape(-1), ((inv[3] * xs + inv[4] * ys + inv[5]) / wv).reshape(-1))
        img = self.W * self.H * self.C * 4
        self.in_bounds = (round(in_remap, 4), round(in_warp, 4))
        self.src_share = (round(px_remap / n, 4), round(px_warp / n, 4))   # share of the source each pass actually reads
        self.alg_bytes_per_launch = int(self.N * (2 * img + (px_remap + px_warp) * self.C * 4 + 2 * n * 4 / 4))

    def roofline_extra(self, mean_step_s):
        return {"survey_bytes_per_launch": self.survey_bytes_per_launch,
                "frac_on_survey_bytes": round(self.survey_bytes_per_launch / mean_step_s / 1e9 / HBM_PEAK_GBS, 4),
                "in_bounds_share_remap_warp": list(self.in_bounds), "source_share_read_remap_warp": list(self.src_share),
                "pricing": "needed bytes: 2 destinations + the distinct source pixels under the taps of each pass + maps once per 4 images; "
                           "survey_bytes = SURVEY.md 8(d)'s 464 486 400 B / image"}

    def step(self):
        from kornia_rs._ffi import lib, check
        n = self.W * self.H * self.C
        s = self.stream.cuda_stream_ptr
        check(lib.kh_remap_f32(s, self.src.ptr, self.mx.ptr, self.my.ptr, self.tmp.ptr, self.W, self.H, self.W, self.H,
                               self.C, 1, self.N, n, n))
        check(lib.kh_warp_perspective_f32(s, self.tmp.ptr, self.dst.ptr, self.W, self.H, self.W, self.H, self.C,
                                          self.hptr, 1, self.N, n, n))

    def describe(self):
        return {"workload": self.name, "op": "interpolation::remap (undistort) -> warp::warp_perspective, bilinear",
                "src": "3840x2160x3 f32", "dst": "same", "batch_per_gpu": self.N,
                "parallelism": "batch-sharded (2048 images = 256 per GPU on 8 GPUs), no collective"}

    def cpu_baseline(self):
        O = self._oracle()
        threads = O.ko.ko_max_threads()
        n = self.W * self.H * self.C
        mx, my = O.correction_map(self.INTR, self.DIST, self.W, self.H)
        frames, t0 = 0, time.perf_counter()
        while True:
            img = self.base[31 * (frames % self.N): 31 * (frames % self.N) + n].reshape(self.H, self.W, self.C)
            O.warp_perspective(O.remap(img, mx, my), self.hm, self.W, self.H)
            frames += 1
            dt = time.perf_counter() - t0
            if dt > 10.0 * CPU_BUDGET_SCALE or frames >= 64:
                break
        return {"value": round(frames * self.W * self.H / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": threads,
                "kind": "port", "sample": f"{frames} images in {dt:.1f} s; C oracle remap+warp_perspective "
                f"(restatement, not the upstream Rust binary), OpenMP x{threads} over rows"}



class UndistortWarp4KList(UndistortWarp4K):
    """configs[4] per-GPU share through `imgproc.remap_batch` + `imgproc.warp_perspective_batch` with separately allocated Images."""

    def __init__(self, batch):
        super().__init__(batch)
        self.cpu_twin = "undistort_warp_4k"
        self.name = f"undistort_remap_then_warp_perspective_4k_f32_api_list_b{batch}"

    def setup(self, stream):
        from kornia_rs import imgproc
        self.stream = stream
        self.src = self._make_images(stream, self.W, self.H, self.C, self.N)
        self.tmp = self._make_images(stream, self.W, self.H, self.C, self.N, fill=False)
        self.dst = self._make_images(stream, self.W, self.H, self.C, self.N, fill=False)
        self.mx, self.my = imgproc.generate_correction_map_polynomial(self.INTR, self.DIST, (self.W, self.H), stream)
        self._price(stream, self.mx.data_ptr, self.my.data_ptr)
        self.src_b, self.tmp_b, self.dst_b = imgproc.ImageBatch(self.src), imgproc.ImageBatch(self.tmp), imgproc.ImageBatch(self.dst)

    def step(self):
        from kornia_rs import imgproc
        imgproc.remap_batch(self.src_b, self.mx, self.my, "bilinear", outs=self.tmp_b)
        imgproc.warp_perspective_batch(self.tmp_b, self.hm, None, "bilinear", outs=self.dst_b)

    def describe(self):
        d = super().describe()
        d.update(workload=self.name, op="imgproc.remap_batch -> imgproc.warp_perspective_batch (kh_remap_f32_list, kh_warp_perspective_f32_list)",
                 operands=f"{self.N} separately allocated Images per stage")
        return d


class ResizeBicubic540(ResizeBilinear):
    """resize bicubic (Keys a = -0.5) 1920x1080 -> 960x540 f32x3, batch 256 — the reference's CUDA bicubic launcher
    (P/cuda/resize.rs:245) on its published 1080p -> 540p shape (benchmarks.md:374).  At exactly 2x the 4x4 windows of the
    destination pixels cover every source pixel, so the algorithmic traffic is the whole source once + the destination."""

    name, kernel = "resize_bicubic_1080p_to_540p_f32_b256", "resize_bicubic_half_kernel<3,vh>"
    DW, DH = 960, 540

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.SW * self.SH / 1e6
        self.alg_bytes_per_launch = self.N * (self.SW * self.SH + self.DW * self.DH) * self.C * 4

    def step(self):
        from kornia_rs import _ffi
        _ffi.check(_ffi.lib.kh_resize_f32(self.stream.cuda_stream_ptr, self.src.ptr, self._next_dst().ptr, self.SW, self.SH, self.DW,
                                          self.DH, self.C, _ffi.KH_INTERP_BICUBIC, self.N, self.SW * self.SH * self.C,
                                          self.DW * self.DH * self.C))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::resize (bicubic, half-pixel)", "src": "1920x1080x3 f32", "dst": "960x540x3 f32",
                "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        O = self._oracle()
        threads, n = O.ko.ko_max_threads(), self.SW * self.SH * self.C
        frames, t0 = 0, time.perf_counter()
        while True:
            O.resize(self.base[31 * (frames % self.N): 31 * (frames % self.N) + n].reshape(self.SH, self.SW, self.C), self.DW, self.DH, "bicubic")
            frames += 1
            dt = time.perf_counter() - t0
            if dt > 8.0 * CPU_BUDGET_SCALE or frames >= 64:
                break
        return {"value": round(frames * self.SW * self.SH / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": threads, "kind": "port",
                "sample": f"{frames} images in {dt:.1f} s; C oracle of resize bicubic (not the upstream Rust binary), OpenMP x{threads} over output rows"}


class SameSizeF32(F32Images):
    """Shared body of the same-size f32x3 maps / stencils / warps: 1R + 1W of the image per launch."""

    W, H, C = 1920, 1080, 3
    op = ""

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * 2 * self.W * self.H * self.C * 4

    def setup(self, stream):
        from kornia_rs.hip import DeviceBuffer
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * self.W * self.H * self.C * 4, stream, zeroed=False)

    def describe(self):
        return {"workload": self.name, "op": self.op, "src": f"{self.W}x{self.H}x3 f32", "dst": "same", "batch_per_gpu": self.N,
                "parallelism": "batch-sharded, no collective"}

    def oracle_call(self, O, img):
        raise NotImplementedError

    def cpu_baseline(self):
        O = self._oracle()
        threads, n = O.ko.ko_max_threads(), self.W * self.H * self.C
        frames, t0 = 0, time.perf_counter()
        while True:
            self.oracle_call(O, self.base[31 * (frames % self.N): 31 * (frames % self.N) + n].reshape(self.H, self.W, self.C))
            frames += 1
            dt = time.perf_counter() - t0
            if dt > 8.0 * CPU_BUDGET_SCALE or frames >= 64:
                break
        return {"value": round(frames * self.W * self.H / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": threads, "kind": "port",
                "sample": f"{frames} images in {dt:.1f} s; C oracle of {self.op} (not the upstream Rust binary), OpenMP x{threads}"}


class WarpAffineF32_1080p(SameSizeF32):
    """warp_affine bilinear f32x3 (rotation 12 deg about the centre, scale 0.9) 1920x1080, batch 256 (P/cuda/warp_affine.rs:75,
    published at benchmarks.md:285).  Out-of-bounds destination pixels write 0 and read nothing; priced as 1R + 1W."""

    name, kernel, op = "warp_affine_f32_1080p_b256", "warp_affine_kernel<3,bilinear>", "imgproc::warp::warp_affine (rot 12 deg, scale 0.9, bilinear)"

    def setup(self, stream):
        import ctypes as C
        from kornia_rs._ffi import lib
        super().setup(stream)
        self.m = (C.c_float * 6)()
        lib.kh_get_rotation_matrix2d(self.W / 2.0, self.H / 2.0, 12.0, 0.9, self.m)

    def step(self):
        from kornia_rs import _ffi
        n = self.W * self.H * self.C
        _ffi.check(_ffi.lib.kh_warp_affine_f32(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H, self.W, self.H,
                                               self.C, self.m, _ffi.KH_INTERP_BILINEAR, self.N, n, n))

    def oracle_call(self, O, img):
        return O.warp_affine(img, list(self.m), self.W, self.H)


class Sobel4K(SameSizeF32):
    """sobel (3x3, magnitude sqrt(gx^2 + gy^2)) f32x3 3840x2160, batch 128 (P/filter/ops.rs:174, published at benchmarks.md:477)."""

    W, H = 3840, 2160
    name, kernel, op = "sobel_3x3_4k_f32_b128", "gradient_magnitude kernel (fused gx, gy, sqrt)", "imgproc::filter::sobel (kernel_size 3)"

    def step(self):
        from kornia_rs import _ffi
        n = self.W * self.H * self.C
        _ffi.check(_ffi.lib.kh_gradient_magnitude_f32(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H, self.C,
                                                      _ffi.KH_GRAD_SOBEL, 3, self.N, n, n))

    def oracle_call(self, O, img):
        return O.gradient_magnitude(img, 0, 3)


class BoxBlur4K(SameSizeF32):
    """box_blur (5, 5) f32x3 3840x2160, batch 128 (P/filter/ops.rs:39): the separable path with 1/5 taps."""

    W, H = 3840, 2160
    name, kernel, op = "box_blur_5x5_4k_f32_b128", "sep_roll4_kernel<5,3>", "imgproc::filter::box_blur (5, 5)"

    def step(self):
        from kornia_rs import _ffi
        n = self.W * self.H * self.C
        _ffi.check(_ffi.lib.kh_box_blur_f32(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H, self.C, 5, 5, self.N, n, n))

    def oracle_call(self, O, img):
        return O.separable_filter(img, O.box_kernel_1d(5), O.box_kernel_1d(5))


class NormalizeMeanStd1080p(SameSizeF32):
    """normalize_mean_std f32x3 (ImageNet mean / std, true IEEE division) 1920x1080, batch 512 (P/normalize.rs:56; the reference
    has no CUDA twin of it — its CPU path is published at 3.8 ms per 1080p frame on Orin)."""

    name, kernel, op = "normalize_mean_std_1080p_f32_b512", "normalize_mean_std kernel", "imgproc::normalize::normalize_mean_std (ImageNet)"

    def step(self):
        import ctypes as C
        from kornia_rs import _ffi
        _ffi.check(_ffi.lib.kh_normalize_mean_std_f32(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.N * self.W * self.H, self.C,
                                                      (C.c_float * 3)(*IMAGENET_MEAN), (C.c_float * 3)(*IMAGENET_STD)))

    def oracle_call(self, O, img):
        return O.normalize_mean_std(img, IMAGENET_MEAN, IMAGENET_STD)


class U8Images(Workload):
    """Shared setup for the u8 HWC workloads (SURVEY 8f.1 twins): image k = LCG base shifted by 31*k bytes."""

    dtype = "u8"
    W, H, C = 3840, 2160, 3

    def _make_src(self, stream, w, h, c, batch):
        from kornia_rs.hip import DeviceBuffer, lib, check
        n = w * h * c
        base = lcg_bytes(n + 31 * batch)
        dbase = DeviceBuffer.from_numpy(base, stream)
        src = DeviceBuffer(n * batch, stream, zeroed=False)
        for k in range(batch):
            check(lib.kh_memcpy_d2d_async(src.ptr + k * n, dbase.ptr + 31 * k, n, stream.cuda_stream_ptr))
        stream.synchronize()
        self.base = base
        return src

    def _oracle(self):
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_ffi as O  # test infrastructure: used here ONLY as the timed CPU baseline
        return O

    def _time_cpu(self, fn, what):
        O = self._oracle()
        threads = O.ko.ko_max_threads()
        reps, t0 = 0, time.perf_counter()
        while True:
            fn(O)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > 8.0 * CPU_BUDGET_SCALE or reps >= 64:
                break
        return {"value": round(reps * self.W * self.H / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": threads, "kind": "port",
                "sample": f"{reps} images in {dt:.1f} s; C oracle of {what} (not the upstream Rust binary), OpenMP x{threads}"}


class GaussianU8_4K(U8Images):
    """gaussian_blur_u8 7x7 sigma 1.5 (Q8 general path) on 3840x2160 RGB8, batch 256."""

    name, kernel = "gaussian_blur_u8_7x7_4k_b256", "blur_u8_roll_kernel<7,3>"

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * 2 * self.W * self.H * self.C  # 1R + 1W

    def setup(self, stream):
        from kornia_rs.hip import DeviceBuffer
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * self.W * self.H * self.C, stream, zeroed=False)

    def step(self):
        from kornia_rs._ffi import lib, check
        n = self.W * self.H * self.C
        check(lib.kh_gaussian_blur_u8(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H, self.C, 7, 7,
                                      1.5, 1.5, self.N, n, n))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::filter::gaussian_blur_u8 (7,7) sigma (1.5,1.5), Q8, fused H+V",
                "src": "3840x2160x3 u8", "dst": "same", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        return self._time_cpu(lambda O: O.gaussian_blur_u8(img, (7, 7), (1.5, 1.5)), "gaussian_blur_u8")


class WarpAffineU8_4K(U8Images):
    """warp_affine_u8 (rotation 12 deg about the centre, scale 0.9) on 3840x2160 RGB8, batch 256."""

    name, kernel = "warp_affine_u8_4k_b256", "gather_u8_staged_kernel<3,affine> (+ affine_rows)"

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * 2 * self.W * self.H * self.C

    def setup(self, stream):
        import ctypes as C
        from kornia_rs.hip import DeviceBuffer
        from kornia_rs._ffi import lib
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * self.W * self.H * self.C, stream, zeroed=False)
        self.m = (C.c_float * 6)()
        lib.kh_get_rotation_matrix2d(self.W / 2.0, self.H / 2.0, 12.0, 0.9, self.m)

    def step(self):
        from kornia_rs._ffi import lib, check
        n = self.W * self.H * self.C
        check(lib.kh_warp_affine_u8(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H, self.W, self.H,
                                    self.C, self.m, self.N, n, n))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::warp::warp_affine_u8 (rot 12 deg, scale 0.9), Q16 span + Q10 bilinear",
                "src": "3840x2160x3 u8", "dst": "same", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        m = list(self.m)
        return self._time_cpu(lambda O: O.warp_affine_u8(img, m, self.W, self.H), "warp_affine_u8")


class WarpPerspectiveU8_4K(U8Images):
    """warp_perspective_u8 (the projective H of the C5 workload) on 3840x2160 RGB8, batch 256 (P/cuda/warp_perspective_u8.rs:59)."""

    name, kernel = "warp_perspective_u8_4k_b256", "gather_u8_staged_kernel<3,perspective> (+ persp_rows)"

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * 2 * self.W * self.H * self.C
        w, h = float(self.W), float(self.H)
        self.hm = [1.03, 0.05, -3.0 * w / 129.0, -0.02, 0.97, 4.0 * h / 97.0, 2.0 / (h * w), 1.5 / (w * h), 1.0]

    def setup(self, stream):
        import ctypes as C
        from kornia_rs.hip import DeviceBuffer
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * self.W * self.H * self.C, stream, zeroed=False)
        self.m = (C.c_float * 9)(*self.hm)

    def step(self):
        from kornia_rs._ffi import lib, check
        n = self.W * self.H * self.C
        check(lib.kh_warp_perspective_u8(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H, self.W, self.H,
                                         self.C, self.m, self.N, n, n))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::warp::warp_perspective_u8 (projective H), row spans + Q10 bilinear",
                "src": "3840x2160x3 u8", "dst": "same", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        return self._time_cpu(lambda O: O.warp_perspective_u8(img, self.hm, self.W, self.H), "warp_perspective_u8")


class RemapU8_4K(U8Images):
    """remap_u8 bilinear with the Brown-Conrady undistortion maps of the C5 workload on 3840x2160 RGB8, batch 256
    (P/cuda/remap.rs:381,444).  Algorithmic bytes: 1R + 1W of the image + the two f32 maps ONCE per batch (they are shared)."""

    name, kernel = "remap_u8_undistort_4k_b256", "gather_u8_staged_kernel<3,remap>"

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * 2 * self.W * self.H * self.C + 2 * self.W * self.H * 4

    def setup(self, stream):
        import ctypes as C
        from kornia_rs.hip import DeviceBuffer
        from kornia_rs._ffi import lib, check
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * self.W * self.H * self.C, stream, zeroed=False)
        self.mx = DeviceBuffer(self.W * self.H * 4, stream, zeroed=False)
        self.my = DeviceBuffer(self.W * self.H * 4, stream, zeroed=False)
        check(lib.kh_correction_map_polynomial_f32(stream.cuda_stream_ptr, self.mx.ptr, self.my.ptr, self.W, self.H,
                                                   (C.c_double * 4)(*UndistortWarp4K.INTR), (C.c_double * 8)(*UndistortWarp4K.DIST)))

    def step(self):
        from kornia_rs._ffi import lib, check
        n = self.W * self.H * self.C
        check(lib.kh_remap_u8(self.stream.cuda_stream_ptr, self.src.ptr, self.mx.ptr, self.my.ptr, self.dst.ptr, self.W, self.H,
                              self.W, self.H, self.C, 1, self.N, n, n))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::interpolation::remap_u8 bilinear (Brown-Conrady undistortion maps), Q10",
                "src": "3840x2160x3 u8 + 2 f32 maps", "dst": "same", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        O = self._oracle()
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        mx, my = O.correction_map(UndistortWarp4K.INTR, UndistortWarp4K.DIST, self.W, self.H)
        return self._time_cpu(lambda O_: O_.remap_u8(img, mx, my, "bilinear"), "remap_u8")


class FusedRgb640(U8Images):
    """fused pipeline read_u8rgb_bilinear -> normalize -> write_chw_f32, 1920x1080 RGB8 -> 640x640
    (the reference's own `probe_fused_1080p` configuration, P/cuda/fusion.rs:849-885), batch 1024."""

    name, kernel = "fused_rgb8_1080p_to_chw640_f32_b1024", "fused_pipeline_kernel<chw>"
    W, H, C, D = 1920, 1080, 3, 640
    dtype = "f32"

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * (self.D * self.D * 12 + self.W * self.H * 3)  # write + whole source once

    def setup(self, stream):
        import ctypes as C
        from kornia_rs.hip import DeviceBuffer
        from kornia_rs import _ffi
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * 3 * self.D * self.D * 4, stream, zeroed=False)
        st = (_ffi.FusedStage * 3)()
        st[0].kind = _ffi.KH_FUSE_READ_U8RGB_BILINEAR
        st[0].u[0], st[0].u[1], st[0].u[2], st[0].u[3] = self.W, self.H, self.D, self.D
        st[1].kind = _ffi.KH_FUSE_NORMALIZE
        for i in range(3):
            st[1].f[i], st[1].f[3 + i] = 1.0 / 255.0, 0.0
        st[2].kind = _ffi.KH_FUSE_WRITE_CHW_F32
        self.h = C.c_void_p()
        _ffi.check(_ffi.lib.kh_fused_pipeline_build(C.cast(st, C.c_void_p), 3, self.D, self.D, self.N, 3 * self.D * self.D,
                                                     C.byref(self.h)))
        n = self.W * self.H * 3
        self.ptrs = (C.c_void_p * self.N)(*[self.src.ptr + k * n for k in range(self.N)])

    def step(self):
        from kornia_rs._ffi import lib, check
        check(lib.kh_fused_pipeline_launch(self.h, self.stream.cuda_stream_ptr, self.ptrs, self.N, self.W * self.H * 3,
                                           self.dst.ptr, self.N * 3 * self.D * self.D))

    def describe(self):
        return {"workload": self.name, "op": "cuda::fusion FusedPipeline [read_u8rgb_bilinear, normalize, write_chw_f32]",
                "src": "1920x1080x3 u8", "dst": "3x640x640 f32", "batch_per_gpu": self.N,
                "parallelism": "batch-sharded, no collective", "launches_per_step": (self.N + 31) // 32}

    def cpu_baseline(self):
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        return self._time_cpu(lambda O: O.fused_pipeline(img, self.D, self.D, [("normalize", [1 / 255.0] * 3, [0.0] * 3)], "chw"),
                              "the fused pipeline")


class ResizeU8_224(U8Images):
    """resize_fast_u8_aa Lanczos-3 (antialiased Q14 separable cascade) 1920x1080 RGB8 -> 224x224, batch 256."""

    name, kernel = "resize_fast_u8_lanczos_1080p_to_224_b256", "sep_h_u8_tile_kernel<3> + sep_v_u8_kernel"
    W, H, C, D = 1920, 1080, 3, 224

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * (self.W * self.H * self.C + self.D * self.D * self.C)  # source once + dst

    def setup(self, stream):
        from kornia_rs.hip import DeviceBuffer
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * self.D * self.D * self.C, stream, zeroed=False)

    def step(self):
        from kornia_rs import _ffi
        _ffi.check(_ffi.lib.kh_resize_fast_u8(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H, self.D, self.D,
                                              self.C, _ffi.KH_INTERP_LANCZOS, 1, self.N, self.W * self.H * self.C,
                                              self.D * self.D * self.C))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::resize::resize_fast_u8_aa(Lanczos, antialias) — two passes, i16 scratch",
                "src": "1920x1080x3 u8", "dst": "224x224x3 u8", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        return self._time_cpu(lambda O: O.resize_fast_u8(img, self.D, self.D, "lanczos", True), "resize_fast_u8_aa")


class ResizeNormChw224(U8Images):
    """resize_normalize_to_tensor_u8_to_f32 (bilinear) 1920x1080 RGB8 -> [3,224,224] f32, batch 256 — the reference's
    CPU timing twin of the camera preprocess (P/resize/fused.rs:57)."""

    name, kernel = "resize_normalize_chw_1080p_to_224_b256", "fused_rgb_chw_kernel<bilinear>"
    W, H, C, D = 1920, 1080, 3, 224
    dtype = "f32"

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        # 4 source taps x 3 B per destination pixel + 12 B written (the source is sub-sampled 8.6x / 4.8x: most of it is never read)
        self.alg_bytes_per_launch = self.N * self.D * self.D * (4 * 3 + 12)

    def setup(self, stream):
        import ctypes as C
        from kornia_rs.hip import DeviceBuffer
        from kornia_rs.hip import IMAGENET_MEAN, IMAGENET_STD
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * 3 * self.D * self.D * 4, stream, zeroed=False)
        f = np.float32
        self.scale_np = (f(1.0) / (np.asarray(IMAGENET_STD, f) * f(255.0))).astype(f)
        self.bias_np = (-np.asarray(IMAGENET_MEAN, f) / np.asarray(IMAGENET_STD, f)).astype(f)
        self.scale, self.bias = (C.c_float * 3)(*self.scale_np), (C.c_float * 3)(*self.bias_np)

    def step(self):
        from kornia_rs import _ffi
        _ffi.check(_ffi.lib.kh_resize_normalize_to_chw_u8_f32(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H,
                                                              self.D, self.D, self.scale, self.bias, _ffi.KH_INTERP_BILINEAR, 1,
                                                              self.N, self.W * self.H * 3, 3 * self.D * self.D))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::resize::resize_normalize_to_tensor_u8_to_f32 (bilinear, ImageNet mean/std)",
                "src": "1920x1080x3 u8", "dst": "3x224x224 f32", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        return self._time_cpu(lambda O: O.resize_normalize_to_chw(img, self.D, self.D, self.scale_np, self.bias_np, "bilinear", True),
                              "resize_normalize_to_tensor_u8_to_f32")


class PyrDownU8_4K(U8Images):
    """pyrdown_u8 (5x5 Gaussian + 2x decimation, one launch) on 3840x2160 RGB8, batch 256."""

    name, kernel = "pyrdown_u8_4k_b256", "pyrdown_u8_tile_kernel<3>"

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * (self.W * self.H * self.C + (self.W // 2) * (self.H // 2) * self.C)

    def setup(self, stream):
        from kornia_rs.hip import DeviceBuffer
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * (self.W // 2) * (self.H // 2) * self.C, stream, zeroed=False)

    def step(self):
        from kornia_rs._ffi import lib, check
        check(lib.kh_pyrdown_u8(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H, self.C, self.N,
                                self.W * self.H * self.C, (self.W // 2) * (self.H // 2) * self.C))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::pyramid::pyrdown_u8 (fused 5x5 Gaussian, reflect-101, 2x decimation)",
                "src": "3840x2160x3 u8", "dst": "1920x1080x3 u8", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        return self._time_cpu(lambda O: O.pyrdown(img), "pyrdown_u8")


class PyramidLevel(Workload):
    """One pyramid step between 3840x2160 and 1920x1080 RGB, u8 or f32: `kind` in pyrup_u8 / pyrdown_f32 / pyrup_f32 (pyrdown_u8 has
    its own class above).  Units = pixels of the LARGER image."""

    def __init__(self, kind, batch):
        self.kind, self.N = kind, batch
        self.up, self.f32 = kind.startswith("pyrup"), kind.endswith("f32")
        self.dtype = "f32" if self.f32 else "u8"
        self.es = 4 if self.f32 else 1
        self.C = 3
        self.sw, self.sh = (1920, 1080) if self.up else (3840, 2160)
        self.dw, self.dh = (3840, 2160) if self.up else (1920, 1080)
        self.name = f"{kind}_4k_b{batch}"
        self.kernel = f"{kind}_kernel<3>"
        self.units_per_step = self.N * 3840 * 2160 / 1e6
        self.alg_bytes_per_launch = self.N * (self.sw * self.sh + self.dw * self.dh) * self.C * self.es

    def setup(self, stream):
        from kornia_rs.hip import DeviceBuffer
        self.stream = stream
        helper = F32Images() if self.f32 else U8Images()
        self.src = helper._make_src(stream, self.sw, self.sh, self.C, self.N)
        self.base = helper.base
        self.dst = DeviceBuffer(self.N * self.dw * self.dh * self.C * self.es, stream, zeroed=False)

    def step(self):
        from kornia_rs._ffi import lib, check
        fn = getattr(lib, f"kh_{self.kind}")
        check(fn(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.sw, self.sh, self.C, self.N,
                 self.sw * self.sh * self.C, self.dw * self.dh * self.C))

    def describe(self):
        return {"workload": self.name, "op": f"imgproc::pyramid::{self.kind} (one launch, both passes)",
                "src": f"{self.sw}x{self.sh}x3 {self.dtype}", "dst": f"{self.dw}x{self.dh}x3 {self.dtype}", "batch_per_gpu": self.N,
                "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_ffi as O  # test infrastructure: used here ONLY as the timed CPU baseline
        img = self.base[: self.sw * self.sh * self.C].reshape(self.sh, self.sw, self.C)
        fn = O.pyrup if self.up else O.pyrdown
        reps, t0 = 0, time.perf_counter()
        while True:
            fn(img)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > 8.0 * CPU_BUDGET_SCALE or reps >= 64:
                break
        return {"value": round(reps * 3840 * 2160 / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": O.ko.ko_max_threads(), "kind": "port",
                "sample": f"{reps} images in {dt:.1f} s; C oracle of {self.kind} (not the upstream Rust binary)"}


class DilateU8_4K(U8Images):
    """u8 dilate, 5x5 box structuring element, constant border, on 3840x2160 RGB8, batch 256."""

    name, kernel = "dilate_u8_box5_4k_b256", "morph_u8_rgb_roll_kernel<5, dilate>"

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * 2 * self.W * self.H * self.C

    def setup(self, stream):
        import ctypes as C
        from kornia_rs.hip import DeviceBuffer
        from kornia_rs import _ffi
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * self.W * self.H * self.C, stream, zeroed=False)
        self.mask = (C.c_uint8 * 25)()
        _ffi.check(_ffi.lib.kh_morph_kernel(_ffi.KH_MORPH_SHAPE["box"], 5, 5, self.mask))
        self.cval = (C.c_uint8 * 4)(0, 0, 0, 0)

    def step(self):
        from kornia_rs import _ffi
        n = self.W * self.H * self.C
        _ffi.check(_ffi.lib.kh_morphology_u8(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W, self.H, self.C,
                                             _ffi.KH_MORPH_DILATE, self.mask, 5, 5, _ffi.KH_BORDER["constant"], self.cval, self.N, n, n))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::morphology::dilate (5x5 box, constant border)", "src": "3840x2160x3 u8",
                "dst": "same", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        return self._time_cpu(lambda O: O.morphology_u8(img, "dilate", O.morph_kernel("box", 5)), "dilate")


class SpatialGradient1080p(F32Images):
    """spatial_gradient_float (normalised 3x3 Sobel, dx + dy) on 1920x1080 f32x3, batch 256."""

    name, kernel = "spatial_gradient_sobel_1080p_f32_b256", "spatial_gradient_x4_kernel<3>"
    W, H, C = 1920, 1080, 3

    def __init__(self, batch):
        self.N = batch
        self.units_per_step = self.N * self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.N * 3 * self.W * self.H * self.C * 4  # 1R + 2W

    def setup(self, stream):
        from kornia_rs.hip import DeviceBuffer
        self.stream = stream
        self.src = self._make_src(stream, self.W, self.H, self.C, self.N)
        self.dst = DeviceBuffer(self.N * self.W * self.H * self.C * 4, stream, zeroed=False)
        self.dst_y = DeviceBuffer(self.N * self.W * self.H * self.C * 4, stream, zeroed=False)

    def step(self):
        from kornia_rs._ffi import lib, check
        n = self.W * self.H * self.C
        check(lib.kh_spatial_gradient_f32(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.dst_y.ptr, self.W, self.H, self.C, 0,
                                          self.N, n, n))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::filter::spatial_gradient_float (3x3 Sobel, dx and dy)", "src": "1920x1080x3 f32",
                "dst": "2 x same", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        O = self._oracle()
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        threads = O.ko.ko_max_threads()
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 8.0 * CPU_BUDGET_SCALE and reps < 64:
            O.spatial_gradient(img, "sobel")
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": round(reps * self.W * self.H / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": threads, "kind": "port",
                "sample": f"{reps} images in {dt:.1f} s; C oracle of spatial_gradient_float_parallel_row (not the upstream Rust binary), OpenMP x{threads}"}


class BoxBlurFast1080p(SpatialGradient1080p):
    """box_blur_fast sigma (2, 2): six running-sum passes through a transposed scratch, 1920x1080 f32x3, batch 64."""

    name, kernel = "box_blur_fast_sigma2_1080p_f32_b64", "fast_hfilter_lds_kernel x6"

    def __init__(self, batch):
        super().__init__(batch)
        self.alg_bytes_per_launch = 6 * self.N * 2 * self.W * self.H * self.C * 4  # six passes per step (the events bracket the step), each 1R + 1W

    def step(self):
        from kornia_rs._ffi import lib, check
        n = self.W * self.H * self.C
        check(lib.kh_box_blur_fast_f32(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.dst_y.ptr, self.W, self.H, self.C, 2.0, 2.0,
                                       self.N, n, n))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::filter::box_blur_fast sigma (2,2): 3 x (row pass -> transposed scratch -> row pass)",
                "src": "1920x1080x3 f32", "dst": "same", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        O = self._oracle()
        img = self.base[: self.W * self.H * self.C].reshape(self.H, self.W, self.C)
        threads = O.ko.ko_max_threads()
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 8.0 * CPU_BUDGET_SCALE and reps < 64:
            O.box_blur_fast(img, (2.0, 2.0))
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": round(reps * self.W * self.H / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": threads, "kind": "port",
                "sample": f"{reps} images in {dt:.1f} s; C oracle of box_blur_fast (single-threaded in the reference), OpenMP x{threads} over rows"}


class ColorMap1080p(Workload):
    """Pointwise colour maps on 1920x1080 images, batch 1024 (one launch over N*W*H pixels): the reference's own headline
    GPU claim is gray_from_rgb f32 at 87 % of its part's peak (crates/kornia-imgproc/benchmarks.md:72)."""

    W, H = 1920, 1080
    SPECS = {  # name -> (entry, dtype, src channels, dst channels, oracle call)
        "gray_u8": ("kh_gray_from_rgb_u8", "u8", 3, 1, "map_u8_quads<Gray>"),
        "gray_f32": ("kh_gray_from_rgb_f32", "f32", 3, 1, "map_f32<Gray>"),
        "hsv_f32": ("kh_hsv_from_rgb_f32", "f32", 3, 3, "map_f32<Hsv>"),
        "bgr_u8": ("kh_bgr_from_rgb_u8", "u8", 3, 3, "map_u8_quads<Swizzle>"),
        # ycbcr_from_rgb (Family A, full-range Q14 / f32; P/cuda/color/yuv.rs:110): the entry takes the channel order as a 5th argument
        "ycbcr_u8": ("kh_ycc_from_rgb_u8", "u8", 3, 3, "map_u8_quads<Ycc>"),
        "ycbcr_f32": ("kh_ycc_from_rgb_f32", "f32", 3, 3, "map_f32<Ycc>"),
    }
    EXTRA = {"kh_ycc_from_rgb_u8": (0,), "kh_ycc_from_rgb_f32": (0,)}  # KH_YCC_YCRCB

    def __init__(self, which, batch):
        self.which, self.N = which, batch
        self.entry, self.dtype, self.cin, self.cout, self.kernel = self.SPECS[which]
        self.item = 1 if self.dtype == "u8" else 4
        self.name = f"{self.entry[3:]}_1080p_b{batch}"
        px = self.W * self.H
        self.units_per_step = self.N * px / 1e6
        self.alg_bytes_per_launch = self.N * px * (self.cin + self.cout) * self.item

    def setup(self, stream):
        from kornia_rs.hip import DeviceBuffer, lib, check
        self.stream = stream
        n = self.W * self.H * self.cin
        raw = lcg_bytes(n + 31 * self.N)
        base = raw if self.dtype == "u8" else (raw.astype(np.float32) / np.float32(255.0))
        dbase = DeviceBuffer.from_numpy(base, stream)
        self.src = DeviceBuffer(n * self.item * self.N, stream, zeroed=False)
        for k in range(self.N):
            check(lib.kh_memcpy_d2d_async(self.src.ptr + k * n * self.item, dbase.ptr + 31 * k * self.item, n * self.item,
                                          stream.cuda_stream_ptr))
        stream.synchronize()
        self.base = base
        self.dst = DeviceBuffer(self.N * self.W * self.H * self.cout * self.item, stream, zeroed=False)

    def step(self):
        from kornia_rs._ffi import lib, check
        check(getattr(lib, self.entry)(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.N * self.W * self.H,
                                       *self.EXTRA.get(self.entry, ())))

    def describe(self):
        return {"workload": self.name, "op": f"imgproc::color::{self.entry[3:]}", "src": f"1920x1080x{self.cin} {self.dtype}",
                "dst": f"1920x1080x{self.cout} {self.dtype}", "batch_per_gpu": self.N, "parallelism": "batch-sharded, no collective"}

    def cpu_baseline(self):
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_ffi as O  # test infrastructure: used here ONLY as the timed CPU baseline
        img = self.base[: self.W * self.H * self.cin].reshape(self.H, self.W, self.cin)
        threads = O.ko.ko_max_threads()
        name = self.entry[3:]
        reps, t0 = 0, time.perf_counter()
        while True:
            O.color_map(name, img, self.cout, *self.EXTRA.get(self.entry, ()))
            reps += 1
            dt = time.perf_counter() - t0
            if dt > 6.0 * CPU_BUDGET_SCALE or reps >= 64:
                break
        return {"value": round(reps * self.W * self.H / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": threads, "kind": "port",
                "sample": f"{reps} images in {dt:.1f} s; C oracle of {name} (not the upstream Rust binary), OpenMP x{threads}, "
                          "row-parallel above 1 Mpx like par_strip_dispatch"}


class GrayPlumbing258x195(Workload):
    """BASELINE configs[0]: color::gray_from_rgb on ONE 258x195 RGB8 image — the reference's CPU Rayon plumbing case
    (tests/data/dog-rgb8.png's size, P/color/gray/mod.rs:254-269).  Not a GPU workload: 50 310 pixels is far below the
    reference's PAR_THRESHOLD of 1 Mpx (P/color/kernel_common.rs:40), so its CPU path runs SERIALLY — and that is what
    is timed here beside the device path (one launch per image, launch-latency bound).  Reported for completeness of the
    five configs; 201 240 algorithmic bytes per image say nothing about HBM."""

    name, kernel, dtype = "gray_from_rgb_u8_258x195_plumbing", "map_u8_quads<Gray>", "u8"
    W, H = 258, 195

    def __init__(self, batch=1):
        self.N = 1
        self.units_per_step = self.W * self.H / 1e6
        self.alg_bytes_per_launch = self.W * self.H * 4

    def setup(self, stream):
        from kornia_rs.hip import DeviceBuffer
        self.stream = stream
        self.base = lcg_bytes(self.W * self.H * 3)
        # eight (source, destination) pairs rotated through, as the reference's harness rotates 8 source buffers (benchmarks.md:29)
        self.pairs = [(DeviceBuffer.from_numpy(self.base, stream), DeviceBuffer(self.W * self.H, stream, zeroed=False)) for _ in range(8)]
        self.src, self.dst = self.pairs[0]
        self.turn = 0

    def step(self):
        from kornia_rs._ffi import lib, check
        self.src, self.dst = self.pairs[self.turn % len(self.pairs)]
        self.turn += 1
        check(lib.kh_gray_from_rgb_u8(self.stream.cuda_stream_ptr, self.src.ptr, self.dst.ptr, self.W * self.H))

    def describe(self):
        return {"workload": self.name, "op": "imgproc::color::gray_from_rgb (u8), one image", "src": "258x195x3 u8", "dst": "258x195x1 u8",
                "batch_per_gpu": 1, "parallelism": "none (BASELINE configs[0]: CPU plumbing; the device line is launch-latency bound)"}

    def cpu_baseline(self):
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_ffi as O  # test infrastructure: used here ONLY as the timed CPU baseline
        img = self.base.reshape(self.H, self.W, 3)
        O.ko.ko_set_threads(1)  # below PAR_THRESHOLD (1 Mpx) the reference's par_strip_dispatch runs the kernel serially
        try:
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 1.0:
                O.gray_from_rgb_u8(img)
                reps += 1
            dt = time.perf_counter() - t0
        finally:
            O.ko.ko_set_threads(0)
        return {"value": round(reps * self.W * self.H / 1e6 / dt, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                "sample": f"{reps} calls in {dt:.1f} s incl. the Python/ctypes call overhead (~{dt / reps * 1e6:.0f} us per call); C oracle of "
                          "gray_from_rgb_u8, serial as the reference's par_strip_dispatch is below 1 Mpx"}


WORKLOADS = {
    "nv12_chw": lambda a: NorthStarNV12(a.batch or 1024, 0),
    "nv12_chw_640": lambda a: NorthStarNV12(a.batch or 1024, 640),
    "nv12_chw_608": lambda a: NorthStarNV12(a.batch or 1024, 608),
    "nv12_chw_640_lanczos": lambda a: NorthStarNV12(a.batch or 256, 640, "lanczos"),
    "yuyv_chw_640": lambda a: NorthStarNV12(a.batch or 1024, 640, "bilinear", "yuyv"),
    "nv12_chw_list": lambda a: NorthStarNV12List(a.batch or 1024),
    "nv12_chw_f16": lambda a: NorthStarNV12F16(a.batch or 1024),
    "nv12_chw_640_f16": lambda a: LetterboxF16(a.batch or 1024, 640, "nv12"),
    "nv12_chw_608_f16": lambda a: LetterboxF16(a.batch or 1024, 608, "nv12"),
    "yuyv_chw_640_f16": lambda a: LetterboxF16(a.batch or 1024, 640, "yuyv"),
    "nv12_h2d_preprocess": lambda a: H2DPreprocess1080p(a.batch or 64),
    "nv12_h2d_preprocess_zero_copy": lambda a: H2DPreprocess1080p(a.batch or 64, pageable=False),
    "resize_224": lambda a: ResizeBilinear(a.batch or 256),
    "resize_224_api_eager": lambda a: ResizeBilinearApi(a.batch or 256, "eager"),
    "resize_224_api_graph": lambda a: ResizeBilinearApi(a.batch or 256, "graph"),
    "resize_224_api_list": lambda a: ResizeBilinearApi(a.batch or 256, "list"),
    "gaussian_4k_api_list": lambda a: Gaussian4KList(a.batch or 256),
    "undistort_warp_4k_api_list": lambda a: UndistortWarp4KList(a.batch or 256),
    "resize_bicubic_540": lambda a: ResizeBicubic540(a.batch or 256),
    "resize_normalize_f32_224": lambda a: ResizeNormalizeF32(a.batch or 256),
    "gaussian_4k": lambda a: Gaussian4K(a.batch or 256),
    "sobel_4k": lambda a: Sobel4K(a.batch or 128),
    "box_blur_4k": lambda a: BoxBlur4K(a.batch or 128),
    "undistort_warp_4k": lambda a: UndistortWarp4K(a.batch or 256),
    "warp_affine_f32_1080p": lambda a: WarpAffineF32_1080p(a.batch or 256),
    "normalize_1080p": lambda a: NormalizeMeanStd1080p(a.batch or 512),
    "gaussian_u8_4k": lambda a: GaussianU8_4K(a.batch or 256),
    "warp_affine_u8_4k": lambda a: WarpAffineU8_4K(a.batch or 256),
    "warp_perspective_u8_4k": lambda a: WarpPerspectiveU8_4K(a.batch or 256),
    "remap_u8_4k": lambda a: RemapU8_4K(a.batch or 256),
    "fused_rgb_640": lambda a: FusedRgb640(a.batch or 1024),
    "resize_u8_224": lambda a: ResizeU8_224(a.batch or 256),
    "resize_norm_chw_224": lambda a: ResizeNormChw224(a.batch or 256),
    "pyrdown_u8_4k": lambda a: PyrDownU8_4K(a.batch or 256),
    "dilate_u8_4k": lambda a: DilateU8_4K(a.batch or 256),
    "pyrup_u8_4k": lambda a: PyramidLevel("pyrup_u8", a.batch or 256),
    "pyrdown_f32_4k": lambda a: PyramidLevel("pyrdown_f32", a.batch or 64),
    "pyrup_f32_4k": lambda a: PyramidLevel("pyrup_f32", a.batch or 64),
    "spatial_gradient_1080p": lambda a: SpatialGradient1080p(a.batch or 256),
    "box_blur_fast_1080p": lambda a: BoxBlurFast1080p(a.batch or 64),
    "gray_u8_1080p": lambda a: ColorMap1080p("gray_u8", a.batch or 1024),
    "gray_f32_1080p": lambda a: ColorMap1080p("gray_f32", a.batch or 1024),
    "hsv_f32_1080p": lambda a: ColorMap1080p("hsv_f32", a.batch or 512),
    "ycbcr_u8_1080p": lambda a: ColorMap1080p("ycbcr_u8", a.batch or 1024),
    "ycbcr_f32_1080p": lambda a: ColorMap1080p("ycbcr_f32", a.batch or 512),
    "bgr_u8_1080p": lambda a: ColorMap1080p("bgr_u8", a.batch or 1024),
    "gray_258x195": lambda a: GrayPlumbing258x195(),
}

# What the default run measures after the headline workload, in the same process: the other BASELINE configs, the north star's
# letterbox secondaries (on-grid 640, off-grid 608, the YUYV source format) and one line for every operator `north_star` names
# (resize bilinear / bicubic, gray + YCbCr + HSV converts, gaussian / box / sobel, warp_affine / warp_perspective + undistort,
# normalize), then the u8 twins.  Each entry is a full roofline record with its own cpu_baseline.  (Median / bilateral / Lab
# are out of SURVEY.md §8 and have no bench line; their kernels are covered by the parity tests only.)
ALSO_DEFAULT = ["nv12_chw_list", "nv12_chw_f16", "nv12_chw_640", "nv12_chw_608", "yuyv_chw_640", "nv12_h2d_preprocess", "nv12_h2d_preprocess_zero_copy", "resize_224",
                "resize_224_api_list", "resize_224_api_graph", "resize_224_api_eager", "gaussian_4k_api_list", "undistort_warp_4k_api_list",
                "resize_bicubic_540", "gaussian_4k", "box_blur_4k", "sobel_4k",
                "undistort_warp_4k", "warp_affine_f32_1080p", "normalize_1080p", "gray_258x195", "gray_u8_1080p", "gray_f32_1080p",
                "ycbcr_u8_1080p", "ycbcr_f32_1080p", "hsv_f32_1080p", "warp_affine_u8_4k", "warp_perspective_u8_4k", "remap_u8_4k",
                "gaussian_u8_4k"]
CPU_TEAM: dict = {}     # cpu_team(), set once in main()
CPU_BUDGET_SCALE = 1.0  # lowered for the `also` entries so the default run stays within a few minutes


def cpu_team() -> dict:
    """How many threads the CPU baseline gets, and why (the reference sizes its Rayon pool to the cores it may
    use, P/parallel.rs:8-10): threads = min(physical cores, CPUs in the affinity mask, ceil(cgroup CPU quota)).  On the pool's boxes
    that is 16 (2 x 64-core EPYC, all 256 hardware threads in the mask, cpu.max = 16 CPUs): round 5 ran 128 spinning OpenMP threads on
    that quota and under-reported the CPU."""
    import math
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    physical = None
    try:
        cores = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        physical = len(cores) or None
    except OSError:
        pass
    if physical is None:
        physical = max(1, (os.cpu_count() or 2) // 2)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    threads = max(1, min(physical, affinity, math.ceil(quota) if quota else physical))
    return {"threads": threads, "physical_cores": physical, "affinity_cpus": affinity, "cgroup_cpus": quota, "logical_cpus": os.cpu_count()}


def run_cpu_baseline(wl, team: dict) -> dict:
    """One workload's CPU leg on the sized team.  `cores` stays what the contract names (the threads that ran: 1 where the reference's
    CPU path is serial); `threads` repeats it, `team_threads` is what the threaded legs were given, and the three host figures it was
    derived from ride INSIDE the object."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_ffi as O  # test infrastructure: used here ONLY as the timed CPU baseline
    if getattr(wl, "cpu_twin", None):   # an API-form row of a workload whose CPU leg is already in the line: name it, do not time it again
        return {"value": None, "unit": "Mpixels/s", "cores": None, "kind": "port", "sample": f"see the row of {wl.cpu_twin}: same arithmetic, same CPU leg"}
    O.ko.ko_set_threads(team["threads"])
    cb = wl.cpu_baseline()
    O.ko.ko_set_threads(team["threads"])   # (a leg that pinned 1 thread restores the OpenMP default, not the team)
    cb.update(threads=cb.get("cores"), team_threads=team["threads"], cgroup_cpus=team["cgroup_cpus"], physical_cores=team["physical_cores"],
              affinity_cpus=team["affinity_cpus"])
    return cb


def store_ceilings(hip, stream, dst_ptr: int, w: int, h: int, nframes: int, reps: int = 5):
    """The ceilings of the north star, timed in THIS process on the headline's own output buffer with the same HIP events
    (VERDICT r02 1c): a flat fill of the output bytes, the production store shape with no loads / decode, and (round 5) a pure READ
    stream of the same bytes (kornia-rs_amd/diag/kh_diag.hip -> lib/libkornia_hip_diag.so, a measurement-only library the product
    never loads)."""
    try:
        d = C.CDLL(str(ROOT / "kornia-rs_amd" / "lib" / "libkornia_hip_diag.so"))
        d.khd_flat_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
        d.khd_three_plane_store.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong]
        d.khd_read_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
        d.khd_stream_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        from kornia_rs.hip import DeviceBuffer
        nbytes = 12 * w * h * nframes
        sink = DeviceBuffer(16, stream)
        # read_stream LAST: it reads what the two fills wrote (never the kernel's output pattern the compare could match)
        runs = {"flat_fill_ms": lambda: d.khd_flat_fill(stream.cuda_stream_ptr, dst_ptr, nbytes),
                "three_plane_store_only_ms": lambda: d.khd_three_plane_store(stream.cuda_stream_ptr, dst_ptr, w, h, nframes, 3 * w * h),
                # the first half of the buffer copied onto the second: R + W = the same bytes (the stream-copy ceiling)
                "stream_copy_ms": lambda: d.khd_stream_copy(stream.cuda_stream_ptr, dst_ptr, dst_ptr + (nbytes // 32) * 16, (nbytes // 32) * 16),
                "read_stream_ms": lambda: d.khd_read_stream(stream.cuda_stream_ptr, dst_ptr, nbytes, sink.ptr)}
        out = {}
        for key, fn in runs.items():
            if fn() != 0:
                return {"store_ceilings_error": f"{key}: launch failed"}
            best = []
            for _ in range(reps):
                e0, e1 = hip.Event(timing=True), hip.Event(timing=True)
                e0.record(stream); fn(); e1.record(stream)
                stream.synchronize()
                best.append(e0.elapsed_ms(e1))
            out[key] = round(float(np.median(best)), 4)
        out["store_bytes"] = nbytes
        return out
    except Exception as e:  # never let the side measurement break the bench line
        return {"store_ceilings_error": str(e)[:120]}


SUMMARY_COLUMNS = ["workload", "Mpx_s", "ms_per_step", "dtype", "roofline_GBps", "roofline_frac", "traffic_over_alg", "cpu_Mpx_s", "cpu_cores"]


def summary_row(rec: dict) -> list:
    """One row per workload of the default run — the headline first — emitted as the LAST key of the line, so that it survives any
    tail truncation.  Round 3 carried every `also` record twice (a compact object AND a summary row); with 22 workloads that no longer
    fits the driver's 8 KB stdout tail, so the table is the record: [workload, Mpixels/s, ms per step, dtype, algorithmic GB/s,
    roofline frac = algorithmic GB/s / 8000, counted HBM bytes / algorithmic bytes | null (a byte ratio: the counters are replayed from
    a separate --pmc run, so no rate is built from them), CPU baseline Mpixels/s | null, CPU threads | null].  The full
    records (config, the whole roofline object, the cpu_baseline sample text) go to gpurun_out/bench_full.json and, for the round's
    reference run, to profiles/."""
    r, c = rec["roofline"], rec.get("cpu_baseline") or {}
    # (the row is labelled with the short --workload key + batch: the long names of 30 rows no longer fit the driver's 8 KB tail)
    return [rec.get("key") or rec["config"]["workload"], rec["value"], rec["ms_per_step"], rec["dtype"], r["achieved"], r["frac"], r.get("traffic_over_alg"),
            c.get("value"), c.get("cores")]


def make_workload(name: str, args) -> Workload:
    """Instantiate a workload; its name always carries the batch it actually runs (`..._b<N>`), which is also what keys the
    replayed PMC traffic — a non-default batch never inherits the default batch's counters."""
    import re
    wl = WORKLOADS[name](args)
    n = getattr(wl, "N", None)
    if n is not None and re.search(r"_b\d+$", wl.name):
        wl.name = re.sub(r"_b\d+$", f"_b{n}", wl.name)
    return wl


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def respawn_under_torchrun(args) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves — one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 — and pass their single JSON line through."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.call(cmd, env=env)


class Runner:
    """Times workloads on this rank's device: K steps bracketed by barrier + sync, per-step HIP events on the launch stream."""

    def __init__(self, hip, torch, dist, stream, rank, local_rank, world, on_gpu=True, use_dist=None, dist_on_gpu=None):
        self.hip, self.torch, self.dist, self.stream = hip, torch, dist, stream
        self.rank, self.local_rank, self.world = rank, local_rank, world
        self.on_gpu = on_gpu   # False only under the host simulator (tests/hostsim): gloo ranks, no torch device
        self.use_dist = world > 1 if use_dist is None else use_dist          # a process group exists: barrier + max-over-ranks through it
        self.dist_on_gpu = on_gpu if dist_on_gpu is None else dist_on_gpu    # RCCL (device tensors) or gloo (host tensors)
        self.traffic_source = None

    def barrier(self):
        self.stream.synchronize()
        if self.on_gpu:
            self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier(device_ids=[self.local_rank]) if self.dist_on_gpu else self.dist.barrier()

    def time(self, wl, steps, warmup, spinup_s=0.0):
        hip, stream = self.hip, self.stream
        # Secondary rows only (spinup_s > 0): the previous row's CPU-baseline leg leaves the GPU idle for 10-30 s and its clocks down,
        # and even between back-to-back rows the host-side setup does: two warm-up steps of a 1-4 ms kernel do not bring them back
        # (box blur 5.2 ms in the line, 4.4 alone, profiles/r04zj_steps.txt; the ALU-heavy 608 letterbox 1.50 ms after 2 warm-up steps,
        # 1.31 after 40, 1.305 after 200, profiles/r04zu_warmup.txt).  Untimed steps of the row's own work for `spinup_s` of wall clock
        # first: the rows are steady-state numbers, like the headline's after its --warmup.
        t_end = time.perf_counter() + spinup_s
        while spinup_s > 0.0 and self.on_gpu and time.perf_counter() < t_end:
            wl.step()
            stream.synchronize()
        for _ in range(warmup):
            wl.step()
        starts = [hip.Event(timing=True) for _ in range(steps)]
        stops = [hip.Event(timing=True) for _ in range(steps)]
        self.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            starts[k].record(stream)   # HIP events on the stream the kernel is launched on
            wl.step()
            stops[k].record(stream)
        stream.synchronize()
        if self.on_gpu:
            self.torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if self.use_dist:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device="cuda" if self.dist_on_gpu else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
            self.dist.barrier(device_ids=[self.local_rank]) if self.dist_on_gpu else self.dist.barrier()
        kernel_ms = [starts[k].elapsed_ms(stops[k]) for k in range(steps)]
        return elapsed, kernel_ms

    def record(self, wl, steps, warmup, elapsed, kernel_ms, key):
        """The measured part of a JSON record for `wl` (rank 0)."""
        mean_kernel_s = float(np.mean(kernel_ms)) / 1e3 if kernel_ms else 0.0
        if not mean_kernel_s > 0.0:  # events without a resolution (host simulator): fall back to the wall clock per step
            mean_kernel_s = elapsed / steps
        achieved = wl.alg_bytes_per_launch / mean_kernel_s / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "pmc_traffic.json"  # per-launch HBM bytes from separate rocprofv3 --pmc passes
        if tfile.exists():
            tj = json.loads(tfile.read_text())
            traffic = tj.get(wl.name)
            if traffic is not None:  # ONE top-level string for the whole line (r02: repeated per record it pushed C2 / C4 out of the driver's tail)
                self.traffic_source = (f"profiles/pmc_traffic.json ({str(tj.get('_source', 'rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE'))[:160]}); "
                                       "replayed from that profile, not counted in this run")
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": wl.kernel,
                "alg_bytes_per_launch": wl.alg_bytes_per_launch, "mean_launch_ms": round(mean_kernel_s * 1e3, 4),
                "min_launch_ms": round(float(np.min(kernel_ms)), 4),
                "launch_ms": [round(float(v), 3) for v in kernel_ms[:32]]}
        # Counted HBM bytes are REPLAYED from a separate --pmc run on another box (profiles/pmc_traffic.json): a byte count is a
        # property of the kernel and the input, so its ratio to the algorithmic bytes is reported (wasted re-reads show there); a RATE
        # built from that run's bytes and THIS run's time would mix two boxes, so traffic_frac stays null unless the counters were
        # taken in this very run.
        roof["traffic_frac"] = None
        if traffic:
            roof["traffic_over_alg"] = round(traffic / wl.alg_bytes_per_launch, 3)
        floor = getattr(wl, "floor_bytes", None)
        if floor:
            # where the algorithmic figure counts only the bytes of the taps (C2) the memory system still moves whole 128-byte lines:
            # the line-granular floor of the access pattern, and the rate at which the kernel moves THAT
            fb = floor()
            roof["floor_bytes"] = fb
            roof["floor_GBps"] = round(fb / mean_kernel_s / 1e9, 1)
            roof["floor_frac"] = round(fb / mean_kernel_s / 1e9 / HBM_PEAK_GBS, 4)
        extra = getattr(wl, "roofline_extra", None)
        if extra:
            roof.update(extra(mean_kernel_s))
        n = getattr(wl, "N", None)
        return {"key": f"{key}_b{n}" if n else key,
                "value": round(self.world * wl.units_per_step * steps / elapsed, 1), "unit": "Mpixels/s", "steps": steps, "warmup": warmup,
                "ms_per_step": round(elapsed / steps * 1e3, 4), "dtype": wl.dtype, "config": wl.describe(), "roofline": roof}


def in_process_sharded(args) -> int:
    """`--in-process`: the north star sharded across `--gpus` devices INSIDE one process (one host thread +
    one non-default stream per device, contiguous batch slices, no collective) through
    kornia_rs.sharding.ShardedPreprocessor — the deployment shape of a batch server.  The driver's scaling runs use the
    process-per-GPU path; this mode exists to measure the thread-per-GPU one beside it."""
    import torch  # noqa: F401  one HIP runtime for the process
    from kornia_rs import hip
    from kornia_rs.sharding import ShardedPreprocessor, ShardPool
    n_dev = hip.device_count()
    devices = [g % max(n_dev, 1) for g in range(args.gpus)]
    if n_dev == 0:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    if args.workload != "nv12_chw":
        # Any other workload (e.g. BASELINE configs[4], --workload undistort_warp_4k): one instance per device, each set up and
        # stepped on its shard's own thread + stream (kornia_rs.sharding.ShardPool — the pool ShardedImgproc runs on), common
        # start barrier, slowest shard's end.  Weak scaling: every device owns a full per-GPU batch.
        pool = ShardPool(devices)
        wls = [make_workload(args.workload, args) for _ in devices]
        pool._each(lambda g: wls[g].setup(pool.streams[g]))
        elapsed = pool.timed_steps(lambda g: wls[g].step(), args.steps, args.warmup)
        per_dev_s = elapsed / args.steps
        name, cus, mem = hip.device_info(0)
        w0 = wls[0]
        print(json.dumps({
            "metric": f"Mpixels/s (source pixels), {w0.name}", "value": round(len(devices) * w0.units_per_step * args.steps / elapsed, 1),
            "unit": "Mpixels/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(per_dev_s * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": w0.dtype,
            "data": "synthetic (LCG bytes, reference pattern_u8; image k shifted by 31k)",
            "config": {**w0.describe(), "launcher": "in-process: one host thread + one stream per device (kornia_rs.sharding.ShardPool)", "devices": devices},
            "roofline": {"bound": "hbm", "achieved": round(w0.alg_bytes_per_launch / per_dev_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(w0.alg_bytes_per_launch / per_dev_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "kernel": w0.kernel,
                         "note": "per device, from the wall clock of the slowest shard thread (no per-launch events in this mode)"},
            "device": {"name": name, "cus": cus, "hbm_bytes": mem, "visible_devices": n_dev}}, separators=(",", ":")), flush=True)
        return 0
    per = args.batch or 1024
    W, H = 1920, 1080
    fb = W * H * 3 // 2
    sp = ShardedPreprocessor(devices, mode="stretch", format="nv12", sampling="bilinear", mean=IMAGENET_MEAN, std=IMAGENET_STD)
    base = lcg_bytes(fb + 31 * per)
    srcs, dsts = [None] * len(devices), [None] * len(devices)

    def setup(g):
        from kornia_rs import Tensor
        from kornia_rs.hip import DeviceBuffer, check, lib
        st = sp.streams[g]
        dbase = DeviceBuffer.from_numpy(base, st)
        src = DeviceBuffer(fb * per, st, zeroed=False)
        for k in range(per):
            check(lib.kh_memcpy_d2d_async(src.ptr + k * fb, dbase.ptr + 31 * k, fb, st.cuda_stream_ptr))
        st.synchronize()
        srcs[g], dsts[g] = src, Tensor.uninit((per, 3, H, W), "float32", st)

    sp._each(setup)
    elapsed = sp.timed_steps(lambda g: sp.shards[g].run_raw_batch(srcs[g], W, H, dsts[g], frame_stride=fb), args.steps, args.warmup)
    mpx = len(devices) * per * W * H / 1e6
    alg = per * (fb + 12 * W * H)
    per_dev_s = elapsed / args.steps
    name, cus, mem = hip.device_info(0)
    print(json.dumps({
        "metric": "Mpixels/s, fused 1080p NV12->normalized CHW f32 (achieved HBM GB/s in roofline)", "value": round(mpx * args.steps / elapsed, 1),
        "unit": "Mpixels/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(per_dev_s * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (LCG bytes, reference pattern_u8; frame k shifted by 31k)",
        "config": {"workload": f"nv12_1080p_to_chw_f32_b{per}", "launcher": "in-process: one host thread + one stream per device (kornia_rs.sharding)",
                   "devices": devices, "batch_per_gpu": per, "parallelism": "batch-sharded, no collective"},
        "roofline": {"bound": "hbm", "achieved": round(alg / per_dev_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / per_dev_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "preprocess_nv12_identity",
                     "note": "per device, from the wall clock of the slowest shard thread (no per-launch events in this mode)"},
        "device": {"name": name, "cus": cus, "hbm_bytes": mem, "visible_devices": n_dev}}), flush=True)
    return 0


def main():
    global CPU_BUDGET_SCALE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=20)   # ~90 ms of the headline kernel: 4.39 ms/step after 2 warm-up steps, 4.35 after 20 or 60 (r04zu)
    ap.add_argument("--workload", default="nv12_chw", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--also", default=None, help="comma-separated workloads measured after the headline one and reported under "
                                                 "\"also\" (default: the other BASELINE configs etc. when the headline is the north star; 'none' disables)")
    ap.add_argument("--also-batch", type=int, default=0, help="batch override for the --also workloads (quick checks)")
    ap.add_argument("--in-process", action="store_true", help="shard across --gpus devices inside ONE process (a thread + stream per device)")
    ap.add_argument("--dev-option", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B runs only: force a library test option (kh_debug_set_option) for this process, e.g. pre_quads=0")
    args = ap.parse_args()

    if args.in_process:
        return in_process_sharded(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return respawn_under_torchrun(args)  # bare `python bench.py --gpus N`: start the ranks ourselves

    import torch  # first: one HIP runtime (torch's bundled libamdhip64) for the whole process
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # KH_HOSTSIM=1 + KORNIA_HIP_LIB=<tests/hostsim build>: the launcher exercised end to end on a GPU-less box — the product's kernels
    # compiled for x86 (TEST INFRASTRUCTURE, tests/test_sharding_gloo.py), one simulated device per rank, gloo instead of RCCL.  The
    # numbers of such a line mean nothing; its shape (n_gpus, aggregate over ranks, max-over-ranks time) is what is checked.
    hostsim = os.environ.get("KH_HOSTSIM") == "1"
    on_gpu = torch.cuda.is_available()
    if not on_gpu and not hostsim:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    # KORNIA_BENCH_DIST_TEST (tests/test_dist_launcher_gpu.py — what a ONE-GPU box can check of the multi-process launcher):
    #   "force": create the RCCL process group at world size 1 too, so that init / barrier(device_ids) / all_reduce(MAX) on a device
    #            tensor / destroy run on real hardware;
    #   "share": every rank uses GPU 0 and the group is gloo (RCCL refuses two ranks on one device): two bench processes launched by
    #            torch.distributed.run drive real kernels at the same time, barrier and aggregate as on a node.
    dist_test = os.environ.get("KORNIA_BENCH_DIST_TEST", "")
    share_gpu = on_gpu and dist_test == "share"
    use_dist = world > 1 or dist_test == "force"
    dist_on_gpu = on_gpu and not share_gpu
    gpu_ordinal = 0 if share_gpu else local_rank
    if on_gpu:
        torch.cuda.set_device(gpu_ordinal)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if dist_on_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu_ordinal), rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from kornia_rs import hip
    if not on_gpu:   # the simulator shows KH_HOSTSIM_DEVICES devices (default 1): rank r owns device r when it exists, as on a node
        local_dev = local_rank if local_rank < hip.device_count() else 0
    else:
        local_dev = gpu_ordinal
    hip.set_device(local_dev)
    for opt in args.dev_option:
        from kornia_rs import _ffi
        name, _, value = opt.partition("=")
        _ffi.check(_ffi.lib.kh_debug_set_option(name.encode(), int(value)))
    stream = hip.Stream.new(local_dev)
    run = Runner(hip, torch, dist, stream, rank, gpu_ordinal, world, on_gpu, use_dist, dist_on_gpu)

    global CPU_TEAM
    CPU_TEAM = cpu_team()
    ranks_seen = world
    if use_dist:   # every rank adds 1 through the process group: the line shows that RCCL (gloo under the simulator) really saw N ranks
        one = torch.ones(1, dtype=torch.float64, device="cuda" if dist_on_gpu else "cpu")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(one.item())))

    wl = make_workload(args.workload, args)
    wl.setup(stream)
    elapsed, kernel_ms = run.time(wl, args.steps, args.warmup)
    line, ceilings, full = None, None, {}
    if rank == 0 and type(wl) is NorthStarNV12 and wl.out == 0:
        ceilings = store_ceilings(hip, stream, wl.dst.data_ptr, wl.W, wl.H, wl.N)
    if rank == 0:
        rec = run.record(wl, args.steps, args.warmup, elapsed, kernel_ms, args.workload)
        name, cus, mem = hip.device_info(local_dev)
        line = {"metric": ("Mpixels/s, fused 1080p NV12->normalized CHW f32 (achieved HBM GB/s in roofline)"
                           if args.workload.startswith("nv12") else f"Mpixels/s (source pixels), {wl.name}"),
                "value": rec["value"], "unit": "Mpixels/s", "n_gpus": world, "n_ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype,
                "data": "synthetic (LCG bytes, reference pattern_u8; frame k shifted by 31k)", "config": rec["config"],
                "roofline": rec["roofline"]}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = run_cpu_baseline(wl, CPU_TEAM)
        rows = [summary_row({**rec, "cpu_baseline": line.get("cpu_baseline")})]

    also = args.also
    if also is None:
        # the secondary rows belong to the one-GPU record; an N-GPU line is the headline only (seconds per rank) unless --also asks
        also = ",".join(ALSO_DEFAULT) if args.workload == "nv12_chw" and not args.batch and world == 1 else "none"
    names = [n for n in also.split(",") if n and n != "none"]
    if names:
        del wl  # frees the headline batch (28.7 GB) before the 4K configs allocate theirs
        import gc
        gc.collect()
        stream.synchronize()
        records = []
        CPU_BUDGET_SCALE = 0.25
        a_steps, a_warm = max(1, min(args.steps, 10)), max(1, min(args.warmup, 2))
        for n in names:
            if n not in WORKLOADS:
                raise SystemExit(f"--also: unknown workload {n!r}")
            w2 = make_workload(n, argparse.Namespace(batch=args.also_batch))
            w2.setup(stream)
            steps2 = 200 if n == "gray_258x195" else a_steps  # a 7 us launch needs more samples than a 10 ms one
            e2, k2 = run.time(w2, steps2, a_warm, spinup_s=0.25)
            if rank == 0:
                r2 = run.record(w2, steps2, a_warm, e2, k2, n)
                r2["n_gpus"] = world
                if world == 1 and not args.no_cpu_baseline:
                    r2["cpu_baseline"] = run_cpu_baseline(w2, CPU_TEAM)
                records.append(r2)
            del w2
            gc.collect()
            stream.synchronize()
        if rank == 0:
            full["also"] = records
            rows += [summary_row(r) for r in records]

    if rank == 0:
        name, cus, mem = hip.device_info(local_dev)
        try:
            affinity = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            affinity = None
        dev = {"name": name, "cus": cus, "hbm_bytes": mem, "host_cpus": os.cpu_count(), "host_affinity_cpus": affinity,
               "hip_runtime": str(hip.runtime_info().get("choice"))[:80]}
        dev["cpu_team"] = CPU_TEAM   # what the CPU baseline's thread count was derived from (the same figures ride inside cpu_baseline)
        if ceilings:
            dev.update(ceilings)
            ms = line["roofline"]["mean_launch_ms"]
            if ceilings.get("flat_fill_ms") and ms:
                dev["frac_of_flat_fill"] = round(ceilings["flat_fill_ms"] / ms, 4)            # kernel time vs a pure write of its output
                dev["frac_of_three_plane_store"] = round(ceilings["three_plane_store_only_ms"] / ms, 4)
                dev["note"] = ("flat_fill / three_plane_store_only: the output bytes written with the production store policy and no loads or "
                               "decode, same process, same buffer, same events (kornia-rs_amd/diag/kh_diag.hip)")
            if ceilings.get("stream_copy_ms"):
                # a flat 1R + 1W copy moving the same number of bytes (half read, half written): the measured ceiling of the same-size maps,
                # filters and warps of the summary, next to the datasheet peak
                dev["stream_copy_GBps"] = round(2 * ((ceilings["store_bytes"] // 32) * 16) / ceilings["stream_copy_ms"] / 1e6, 1)
            if ceilings.get("read_stream_ms") and ms:
                # the measured pure-read rate of this part (16 B / lane over the same 25.5 GB) and the kernel's total R + W rate against it:
                # north_star's "HBM-read roofline" taken as what a read stream actually reaches, beside roofline.frac (datasheet 8 TB/s)
                rd = ceilings["store_bytes"] / ceilings["read_stream_ms"] / 1e6
                dev["read_stream_GBps"] = round(rd, 1)
                dev["kernel_rate_over_read_stream_rate"] = round(line["roofline"]["achieved"] / rd, 4)
        line["device"] = dev
        if args.dev_option:
            line["dev_options"] = args.dev_option   # a line produced with forced code paths says so
        if not on_gpu:
            line["data"] += "; HOST SIMULATOR run (KH_HOSTSIM=1): launcher check only, the numbers are meaningless"
        if run.traffic_source:
            line["traffic_source"] = run.traffic_source
        # LAST key: one row per workload (SUMMARY_COLUMNS)
        line["summary_columns"] = SUMMARY_COLUMNS
        line["summary"] = rows
        try:  # the uncompacted records, for profiles/ (scratch on the driver's box)
            outdir = ROOT / "gpurun_out"
            outdir.mkdir(exist_ok=True)
            (outdir / "bench_full.json").write_text(json.dumps({**line, **full}, indent=1))
        except OSError:
            pass
        print(json.dumps(line, separators=(",", ":")), flush=True)

    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
