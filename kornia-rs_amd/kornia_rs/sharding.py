"""Batch sharding across the GPUs of one node — the only multi-GPU logic the path needs.

Frames / images are independent units: shard ``g`` of ``G`` owns the contiguous slice ``[g*N/G, (g+1)*N/G)`` on its own
device, with its own non-default stream and its own host thread; kernel parameters are replicated; there is NO data-path
collective.  The reference has no multi-device logic beyond a same-device check (P/cuda/dispatch.rs:51-53) and
per-ordinal streams (kornia-py/src/cuda_ext/mod.rs:61-84) — this module is the piece a batch server adds on top:

* ``shard_range``            the contiguous, balanced partition (also used by ``bench.py`` ranks);
* ``ShardedPreprocessor``    in-process sharder for the fused camera preprocess: one ``Preprocessor`` + ``Stream`` per
                             device, one worker thread per device (``hipSetDevice`` is per-thread state; ctypes calls drop
                             the GIL, so uploads and launches of different devices overlap);
* ``ShardedImgproc``         the same sharder for ANY ``imgproc`` operator (round 3): ``scatter`` a host batch into per-device
                             images, ``replicate`` shared operands (undistortion maps, homographies, LUTs), ``map`` an operator
                             over every image on its owner's stream, ``gather``; ``undistort_warp`` is BASELINE configs[4]
                             (remap with Brown-Conrady maps, then warp_perspective) on top of it; ``plan`` is the pure
                             partition both launchers (threads here, ranks under torchrun) use;
* ``aggregate_throughput``   the process-per-GPU reduction used under ``torch.distributed`` (RCCL / gloo): slowest rank's
                             time, sum of units — the only collective anywhere, and it is not on the data path.

``devices`` may repeat an ordinal (``[0, 0]``): two shards with their own streams and threads on one GPU — how the sharder
is exercised on a single-GPU box.
"""
from __future__ import annotations

import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Callable, List, Optional, Sequence, Tuple, Union

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first ``n_items % world`` ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    if n_items < 0:
        raise ValueError("negative item count")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def aggregate_throughput(units_this_rank: float, elapsed_s: float, dist=None, device=None) -> Tuple[float, float]:
    """(total units over all ranks, slowest rank's elapsed seconds).  ``dist`` is an initialised
    ``torch.distributed`` module or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return units_this_rank, elapsed_s
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units_this_rank], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item())


class ShardedBatch:
    """The result of a sharded call: ``shards[g]`` is the ``[n_g, 3, H, W]`` device tensor of slice ``ranges[g]`` on
    ``devices[g]``.  Nothing has been synchronised: each shard is ordered on its own stream."""

    def __init__(self, shards: List[Any], ranges: List[Tuple[int, int]], devices: List[int]):
        self.shards, self.ranges, self.devices = shards, ranges, devices

    def synchronize(self) -> None:
        for t in self.shards:
            if t is not None and t.stream is not None:
                t.stream.synchronize()

    def numpy(self) -> np.ndarray:
        """Host copy of the whole batch, shards concatenated in batch order (a D2H copy per shard)."""
        parts = [t.numpy_raw() for t in self.shards if t is not None and t.shape[0] > 0]
        if not parts:
            raise ValueError("empty batch")
        return np.concatenate(parts, axis=0)

    def __len__(self) -> int:
        return sum(hi - lo for lo, hi in self.ranges)


class ShardPool:
    """Devices, one non-default stream and one worker thread per shard: what every in-process sharder shares."""

    def __init__(self, devices: Sequence[int]):
        from . import hip
        if not devices:
            raise ValueError(f"{type(self).__name__} needs at least one device ordinal")
        n_dev = hip.device_count()
        for d in devices:
            if not (0 <= int(d) < n_dev):
                raise ValueError(f"device ordinal {d} out of range (this process sees {n_dev} HIP device(s))")
        self.devices = [int(d) for d in devices]
        self.streams = [hip.Stream.new(d) for d in self.devices]
        self._pool = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix="kornia-shard")

    @property
    def world(self) -> int:
        return len(self.devices)

    def close(self) -> None:
        """Every worker lets go of its page-locked bounce buffers (hip.d2h / hip.h2d pin 16 MiB per copying thread and device),
        then the pool shuts down.  The ``world`` release tasks meet at a barrier before they free anything, so each of the pool's
        workers runs exactly one of them — an idle worker cannot take two and leave another's buffer pinned until thread exit.
        Errors raised by the release itself are re-raised, not swallowed."""
        import threading
        from . import hip
        if getattr(self, "_closed", False):
            return
        self._closed = True
        if threading.current_thread().name.startswith("kornia-shard"):
            # close() running ON one of the pool's own workers (e.g. a finaliser the collector ran there): the barrier below could only
            # ever see world - 1 parties and shutdown(wait=True) would join the calling thread.  Let go of what this thread holds and
            # shut down without waiting; the other workers release their buffers when their threads end.
            hip.release_thread_staging()
            self._pool.shutdown(wait=False)
            return
        barrier = threading.Barrier(self.world)

        def release(_g: int):
            try:
                barrier.wait(timeout=30.0)   # all `world` tasks are running, i.e. one per worker thread
            except threading.BrokenBarrierError:
                pass                         # a worker died or the pool is wedged: still free what this thread holds
            hip.release_thread_staging()

        try:
            futures = [self._pool.submit(release, g) for g in range(self.world)]
        except RuntimeError:                 # "cannot schedule new futures after shutdown": nothing left to release through the pool
            futures = []
            barrier.abort()
        errors = []
        for f in futures:
            try:
                f.result()
            except Exception as e:           # noqa: BLE001 — collected, reported after the pool is down
                errors.append(e)
        self._pool.shutdown(wait=True)
        if errors:
            raise errors[0]

    def __del__(self):
        try:
            self._pool.shutdown(wait=False)
        except Exception:
            pass

    def _each(self, fn: Callable[[int], Any]) -> List[Any]:
        """Run ``fn(g)`` for every shard on the worker pool with shard ``g``'s device current; re-raises the first error."""
        from . import hip

        def task(g: int):
            hip.set_device(self.devices[g])
            return fn(g)

        futures = [self._pool.submit(task, g) for g in range(self.world)]
        return [f.result() for f in futures]

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()

    def timed_steps(self, step: Callable[[int], None], steps: int, warmup: int = 0) -> float:
        """Run ``step(g)`` ``warmup + steps`` times on every shard thread; the timed part starts behind a common barrier
        (every shard's stream drained) and ends when the slowest shard has drained its stream.  Returns seconds."""
        barrier = threading.Barrier(self.world)
        spans: List[Tuple[float, float]] = [(0.0, 0.0)] * self.world

        def body(g: int):
            try:
                for _ in range(warmup):
                    step(g)
                self.streams[g].synchronize()
            except BaseException:
                barrier.abort()  # a shard that failed in warm-up must not leave the others waiting for it forever
                raise
            try:
                barrier.wait(timeout=600.0)
            except threading.BrokenBarrierError:
                raise RuntimeError(f"shard {g}: another shard failed (or hung for 10 min) before the timed region; see its error") from None
            t0 = time.perf_counter()
            for _ in range(steps):
                step(g)
            self.streams[g].synchronize()
            spans[g] = (t0, time.perf_counter())

        self._each(body)
        return max(t1 for _, t1 in spans) - min(t0 for t0, _ in spans)


class ShardedPreprocessor(ShardPool):
    """One ``Preprocessor`` per device behind the single-device call shape.

    ``run_raw_batch(frames, src_w, src_h, out_h, out_w)`` takes either
      * a host array ``[N, frame_bytes]`` uint8 (or a list of N 1-D uint8 arrays): slice ``g`` is staged through shard
        ``g``'s persistent page-locked buffer and uploaded on shard ``g``'s stream, or
      * a list with ONE entry per shard, each a device buffer on that shard's device holding its slice back to back
        (``frame_stride`` bytes apart) together with the slice length: ``[(buf0, n0), (buf1, n1), ...]``
    and returns a ``ShardedBatch``.  Every shard runs on its own worker thread: device selection, upload, launch.
    """

    def __init__(self, devices: Sequence[int], **preprocessor_kwargs: Any):
        from .preprocess import Preprocessor
        if "stream" in preprocessor_kwargs:
            raise ValueError("ShardedPreprocessor creates one stream per device; do not pass stream=")
        super().__init__(devices)
        self.shards = [Preprocessor(stream=s, **preprocessor_kwargs) for s in self.streams]

    def alloc_output(self, n_frames: int, out_height: int, out_width: int, zeroed: bool = False) -> List[Any]:
        """Per-shard destination tensors for a batch of ``n_frames`` (uninitialised by default: the kernel writes every element)."""
        from .tensor import Tensor

        def alloc(g: int):
            lo, hi = shard_range(n_frames, g, self.world)
            dt = "float16" if self.shards[g].f16 else "float32"
            make = Tensor.zeros if zeroed else Tensor.uninit
            return make((hi - lo, 3, out_height, out_width), dt, self.streams[g])

        return self._each(alloc)

    def upload(self, frames: Union[np.ndarray, Sequence[np.ndarray]]) -> Tuple[List[Tuple[Any, int]], int]:
        """Stage + upload the host batch: slice ``g`` to device ``g`` (async on its stream through the shard's persistent
        pinned buffer).  Returns ``([(device_view, n_g), ...], frame_stride)``."""
        rows = self._host_rows(frames)
        n = len(rows)
        strides: List[int] = [0] * self.world

        def up(g: int):
            lo, hi = shard_range(n, g, self.world)
            if hi == lo:
                return (None, 0)
            view, stride = self.shards[g]._staging.upload(self.streams[g], rows[lo:hi])
            strides[g] = stride
            return (view, hi - lo)

        parts = self._each(up)
        used = {s for s, (_, k) in zip(strides, parts) if k}
        return parts, (used.pop() if used else 0)

    @staticmethod
    def _host_rows(frames) -> List[np.ndarray]:
        if isinstance(frames, np.ndarray):
            if frames.dtype != np.uint8 or frames.ndim != 2:
                raise TypeError("host batch must be a uint8 array of shape [N, frame_bytes]")
            return [frames[k] for k in range(frames.shape[0])]
        rows = [np.asarray(f) for f in frames]
        if any(r.dtype != np.uint8 for r in rows):
            raise TypeError("raw frames must be uint8")
        if len({r.size for r in rows}) > 1:
            from .preprocess import PreprocessError
            raise PreprocessError("InvalidRawSource", "batched frames must have the same length")
        return [r.reshape(-1) for r in rows]

    def run_raw_batch(self, frames: Any, src_w: int, src_h: int, out_height: int, out_width: int, *,
                      frame_stride: Optional[int] = None, out: Optional[List[Any]] = None) -> ShardedBatch:
        device_parts = (isinstance(frames, (list, tuple)) and len(frames) == self.world
                        and all(isinstance(p, tuple) and len(p) == 2 for p in frames))
        if device_parts:
            parts = list(frames)
            if frame_stride is None:
                frame_stride = self.shards[0].source_format.buffer_len(src_w, src_h)
        else:
            parts, frame_stride = self.upload(frames)
        counts = [k for _, k in parts]
        n = sum(counts)
        ranges, lo = [], 0
        for k in counts:
            ranges.append((lo, lo + k))
            lo += k
        if out is not None:
            if len(out) != self.world:
                raise ValueError(f"out= must hold one tensor per shard ({self.world})")
            dsts = list(out)
        else:
            from .tensor import Tensor
            dt = "float16" if self.shards[0].f16 else "float32"
            dsts = self._each(lambda g: Tensor.uninit((counts[g], 3, out_height, out_width), dt, self.streams[g]))

        for g, t in enumerate(dsts):
            if t is not None and (len(t.shape) != 4 or t.shape[0] != counts[g]):
                from .preprocess import PreprocessError
                raise PreprocessError("BatchMismatch", f"shard {g}: destination batch dim {t.shape[0] if len(t.shape) == 4 else t.shape} "
                                                       f"!= frame count {counts[g]}", dst_n=t.shape[0] if t.shape else 0, frames=counts[g])

        for g, t in enumerate(dsts):  # a shard that has frames needs somewhere to put them (typed, not an AttributeError in a worker)
            if t is None and counts[g] > 0:
                from .preprocess import PreprocessError
                raise PreprocessError("BatchMismatch", f"shard {g}: out[{g}] is None but the shard holds {counts[g]} frames",
                                      dst_n=0, frames=counts[g])

        def run(g: int):
            buf, k = parts[g]
            if k == 0:
                return None
            if dsts[g].device_id != self.devices[g]:
                from .image import ImageError
                raise ImageError("DeviceMismatch", f"shard {g} runs on device {self.devices[g]} but its destination lives on "
                                                   f"device {dsts[g].device_id}")
            self.shards[g].run_raw_batch(buf, src_w, src_h, dsts[g], frame_stride=frame_stride)
            if not device_parts:
                self.shards[g]._staging.mark_consumed(self.streams[g])   # the upload ring may reuse the slot once this launch is done
            return None

        self._each(run)
        return ShardedBatch(dsts, ranges, list(self.devices))


class Replicated:
    """An operand every shard needs its own copy of (maps, LUTs, matrices): ``per_shard[g]`` lives on ``devices[g]``."""

    def __init__(self, per_shard: List[Any]):
        self.per_shard = per_shard


class ShardedImages:
    """A batch of device images spread over shards: ``shards[g]`` is the list of ``Image`` of slice ``ranges[g]`` on
    ``devices[g]``, each ordered on that shard's stream.  Nothing has been synchronised."""

    def __init__(self, shards: List[List[Any]], ranges: List[Tuple[int, int]], devices: List[int]):
        self.shards, self.ranges, self.devices = shards, ranges, devices

    def __len__(self) -> int:
        return sum(len(s) for s in self.shards)

    def numpy(self) -> List[np.ndarray]:
        """Host copies in batch order (one D2H copy per image, on its owner's stream)."""
        return [img.cpu().numpy() if hasattr(img, "cpu") else img.numpy() for shard in self.shards for img in shard]


def plan(n_items: int, world: int) -> List[Tuple[int, int]]:
    """The partition of ``n_items`` independent units over ``world`` shards / ranks: contiguous, balanced, in order, covering
    ``[0, n_items)`` exactly once.  BASELINE configs[4]: ``plan(2048, 8)`` = 256 images per GPU."""
    return [shard_range(n_items, g, world) for g in range(world)]


class ShardedImgproc(ShardPool):
    """Any ``kornia_rs.imgproc`` operator over a batch of independent images, sharded across devices.

        sp = ShardedImgproc([0, 1, 2, 3])
        batch = sp.scatter(host_images)                                   # slice g -> device g, on stream g
        mx, my = sp.replicate_fn(lambda st: imgproc.generate_correction_map_polynomial(K, D, (w, h), st))
        out = sp.map(imgproc.remap, batch, mx, my)                        # every image on its owner's thread + stream
        out = sp.map(imgproc.warp_perspective, out, H)
        host = out.numpy()

    Operands wrapped in ``Replicated`` are resolved to the shard's own copy; everything else is passed through.  There is no
    collective and no cross-device access: an image is only ever touched by the device that owns its slice (the reference's
    same-device rule, P/cuda/dispatch.rs:51-53, holds per shard by construction)."""

    def scatter(self, images: Sequence[Any]) -> ShardedImages:
        """Upload a host batch: images ``[lo_g, hi_g)`` to device ``devices[g]`` on ``streams[g]``.  Accepts numpy HWC arrays or
        host ``Image`` objects."""
        from .image import Image
        imgs = list(images)
        ranges = plan(len(imgs), self.world)

        def up(g: int):
            lo, hi = ranges[g]
            out = []
            for k in range(lo, hi):
                im = imgs[k] if isinstance(imgs[k], Image) else Image.from_numpy(np.asarray(imgs[k]))
                if im.is_device:
                    raise ValueError("scatter takes host images; device images already have an owner")
                out.append(im.to_hip(self.streams[g]))
            return out

        return ShardedImages(self._each(up), ranges, list(self.devices))

    def replicate(self, value: Any) -> Replicated:
        """One device copy per shard of a host ``Image`` / numpy array (maps, LUT images)."""
        from .image import Image

        def up(g: int):
            im = value if isinstance(value, Image) else Image.from_numpy(np.asarray(value))
            return im.to_hip(self.streams[g])

        return Replicated(self._each(up))

    def replicate_fn(self, make: Callable[[Any], Any]):
        """Build a replicated operand ON each device: ``make(stream_g)`` runs on shard ``g``'s thread (e.g. the undistortion maps,
        generated in f64 on the device — 66 MB per camera never cross PCIe).  A tuple result becomes a tuple of ``Replicated``."""
        made = self._each(lambda g: make(self.streams[g]))
        if made and isinstance(made[0], tuple):
            return tuple(Replicated([m[i] for m in made]) for i in range(len(made[0])))
        return Replicated(made)

    def map(self, fn: Callable[..., Any], batch: ShardedImages, *args: Any, **kwargs: Any) -> ShardedImages:
        """``fn(image, *args, **kwargs)`` for every image of the batch, on the worker thread and stream of the shard that owns it."""
        if batch.devices != self.devices:
            raise ValueError(f"batch was scattered over devices {batch.devices}, this sharder runs on {self.devices}")

        def run(g: int):
            a = [x.per_shard[g] if isinstance(x, Replicated) else x for x in args]
            kw = {k: (v.per_shard[g] if isinstance(v, Replicated) else v) for k, v in kwargs.items()}
            return [fn(img, *a, **kw) for img in batch.shards[g]]

        return ShardedImages(self._each(run), list(batch.ranges), list(batch.devices))

    def undistort_warp(self, batch: ShardedImages, intrinsic: Sequence[float], distortion: Sequence[float],
                       homography: Sequence[float], interpolation: str = "bilinear", maps: Optional[Tuple[Replicated, Replicated]] = None
                       ) -> ShardedImages:
        """BASELINE configs[4]: ``remap`` with the Brown-Conrady correction maps of (intrinsic, distortion), then
        ``warp_perspective`` with ``homography``; maps replicated per device (built there unless ``maps`` is given), ``H``
        replicated by value."""
        from . import imgproc
        first = next((s[0] for s in batch.shards if s), None)
        if first is None:
            return ShardedImages([[] for _ in self.devices], list(batch.ranges), list(batch.devices))
        size = (first.width, first.height)
        if maps is None:
            maps = self.replicate_fn(lambda st: imgproc.generate_correction_map_polynomial(intrinsic, distortion, size, st))
        mx, my = maps
        und = self.map(imgproc.remap, batch, mx, my, interpolation)
        return self.map(imgproc.warp_perspective, und, homography, (size[1], size[0]), interpolation)  # the Python API takes (height, width)
