"""Batch sharding across the GPUs of one node — the only multi-GPU logic the path needs.

Frames / images are independent units: shard ``g`` of ``G`` owns the contiguous slice ``[g*N/G, (g+1)*N/G)`` on its own
device, with its own non-default stream and its own host thread; kernel parameters are replicated; there is NO data-path
collective.  The reference has no multi-device logic beyond a same-device check (P/cuda/dispatch.rs:51-53) and
per-ordinal streams (kornia-py/src/cuda_ext/mod.rs:61-84) — this module is the piece a batch server adds on top:

* ``shard_range``            the contiguous, balanced partition (also used by ``bench.py`` ranks);
* ``ShardedPreprocessor``    in-process sharder for the fused camera preprocess: one ``Preprocessor`` + ``Stream`` per
                             device, one worker thread per device (``hipSetDevice`` is per-thread state; ctypes calls drop
                             the GIL, so uploads and launches of different devices overlap);
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


class ShardedPreprocessor:
    """One ``Preprocessor`` per device behind the single-device call shape.

    ``run_raw_batch(frames, src_w, src_h, out_h, out_w)`` takes either
      * a host array ``[N, frame_bytes]`` uint8 (or a list of N 1-D uint8 arrays): slice ``g`` is staged through shard
        ``g``'s persistent page-locked buffer and uploaded on shard ``g``'s stream, or
      * a list with ONE entry per shard, each a device buffer on that shard's device holding its slice back to back
        (``frame_stride`` bytes apart) together with the slice length: ``[(buf0, n0), (buf1, n1), ...]``
    and returns a ``ShardedBatch``.  Every shard runs on its own worker thread: device selection, upload, launch.
    """

    def __init__(self, devices: Sequence[int], **preprocessor_kwargs: Any):
        from . import hip
        from .preprocess import Preprocessor
        if not devices:
            raise ValueError("ShardedPreprocessor needs at least one device ordinal")
        n_dev = hip.device_count()
        for d in devices:
            if not (0 <= int(d) < n_dev):
                raise ValueError(f"device ordinal {d} out of range (this process sees {n_dev} HIP device(s))")
        if "stream" in preprocessor_kwargs:
            raise ValueError("ShardedPreprocessor creates one stream per device; do not pass stream=")
        self.devices = [int(d) for d in devices]
        self.streams = [hip.Stream.new(d) for d in self.devices]
        self.shards = [Preprocessor(stream=s, **preprocessor_kwargs) for s in self.streams]
        self._pool = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix="kornia-shard")

    @property
    def world(self) -> int:
        return len(self.devices)

    def close(self) -> None:
        self._pool.shutdown(wait=True)

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
            return None

        self._each(run)
        return ShardedBatch(dsts, ranges, list(self.devices))

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
