#!/usr/bin/env python3
"""Dev tooling (GPU box): A/B the north-star kernel's DRAM-side layout through the REAL library entry.

DESIGN.md section 9.1: the three-stream plane store is the limit (5.9 TB/s by itself); what has not been measured is whether the
frame / plane placement in HBM matters.  This sweeps, for 1024 frames of 1920x1080 NV12 -> CHW f32 with the production kernel
(kh_preprocess_to_chw, identity fast path):
  * dst_frame_stride padded from 3*W*H floats (24 883 200 B, 4 KiB-aligned only) up to 64 KiB / 2 MiB multiples,
  * the destination base offset (0 / 2 KiB / 64 KiB past a 2 MiB boundary),
  * src_frame_stride padded to 4 KiB / 2 MiB multiples,
interleaved over several rounds, and prints median ms + algorithmic TB/s per variant.  Nothing here changes the product.

    python scripts/ab_north_star_stride.py [--frames 1024] [--rounds 5] > gpurun_out/ab_stride.log
"""
import argparse
import ctypes as C
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "kornia-rs_amd"))

import numpy as np  # noqa: E402


def up(x, a):
    return (x + a - 1) // a * a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    from kornia_rs import IMAGENET_MEAN, IMAGENET_STD, Preprocessor, hip
    from kornia_rs._ffi import check, lib
    W, H, N = 1920, 1080, args.frames
    frame = W * H * 3 // 2
    plane3 = 3 * W * H  # floats per destination frame
    stream = hip.Stream.new(0)
    pre = Preprocessor(mode="stretch", format="nv12", sampling="bilinear", mean=IMAGENET_MEAN, std=IMAGENET_STD, stream=stream)
    MiB2 = 2 << 20
    src_strides = {"tight": frame, "4KiB": up(frame, 4096), "2MiB": up(frame, MiB2)}
    dst_strides = {"tight": plane3, "64KiB": up(plane3 * 4, 65536) // 4, "2MiB": up(plane3 * 4, MiB2) // 4,
                   "2MiB+4KiB": up(plane3 * 4, MiB2) // 4 + 1024, "2MiB+64KiB": up(plane3 * 4, MiB2) // 4 + 16384}
    offsets = {"0": 0, "2KiB": 2048, "64KiB": 65536}
    src_buf = hip.DeviceBuffer(max(src_strides.values()) * N + MiB2, stream, zeroed=True)
    dst_buf = hip.DeviceBuffer(max(dst_strides.values()) * 4 * N + 2 * MiB2, stream, zeroed=False)
    src_base, dst_base = up(src_buf.ptr, MiB2), up(dst_buf.ptr, MiB2)
    rng = np.random.default_rng(0).integers(0, 256, frame, dtype=np.uint8)
    one = hip.DeviceBuffer.from_numpy(rng, stream)
    variants = []
    for sname, ss in src_strides.items():
        variants.append((f"src {sname:6s} dst tight      +0", ss, plane3, 0))
    for dname, ds in dst_strides.items():
        for oname, off in offsets.items():
            if dname == "tight" and oname == "0":
                continue
            variants.append((f"src tight  dst {dname:10s} +{oname}", frame, ds, off))
    times = {v[0]: [] for v in variants}
    e0, e1 = hip.Event(timing=True), hip.Event(timing=True)
    for rnd in range(args.rounds):
        for name, ss, ds, off in variants:
            for k in range(N):  # same frame bytes at every stride
                check(lib.kh_memcpy_d2d_async(src_base + k * ss, one.ptr, frame, stream.cuda_stream_ptr))
            p = pre._params(W, H, W, 1, pre.source_format.fmt_code, W, H, N, ss, False, False)
            p.dst_frame_stride = ds
            check(lib.kh_preprocess_to_chw(stream.cuda_stream_ptr, src_base, dst_base + off, C.byref(p)))  # warm-up
            e0.record(stream)
            for _ in range(args.reps):
                check(lib.kh_preprocess_to_chw(stream.cuda_stream_ptr, src_base, dst_base + off, C.byref(p)))
            e1.record(stream)
            stream.synchronize()
            times[name].append(e0.elapsed_ms(e1) / args.reps)
    alg = N * (frame + 12 * W * H)
    print(f"# north star, {N} frames, production kernel; algorithmic bytes {alg / 1e9:.2f} GB; median of {args.rounds} interleaved rounds x {args.reps} launches")
    print(f"{'variant':42s} {'med ms':>8s} {'min ms':>8s} {'TB/s@med':>9s}")
    for name, *_ in variants:
        t = times[name]
        med = statistics.median(t)
        print(f"{name:42s} {med:8.3f} {min(t):8.3f} {alg / max(med, 1e-9) / 1e9:9.3f}")


if __name__ == "__main__":
    main()
