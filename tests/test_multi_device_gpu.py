"""Device affinity of the host layer: everything below drives DEVICE 1 from a thread whose current device is 0 — the
situation of ``ShardedBatch.numpy()``, of a pool worker that serves two shards, of a service thread that owns several GPUs.

No test box of any round had two GPUs, so these run where two devices exist: on the host simulator with ``KH_HOSTSIM_DEVICES=2``
(tests/test_zz_hostsim.py; its runtime refuses, like HIP, an event recorded on a stream of another device and — stricter than
HIP — a launch on a stream of another device than the current one), and on any real multi-GPU node.  ADVICE r04 (high): the
double-buffered pageable copies recorded per-thread events on streams of whatever device; found by review, reproduced by the
simulator (9 failures in test_sharding_gpu.py before the fix), pinned here.
"""
import threading

import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits

pytestmark = pytest.mark.gpu

IMAGENET = dict(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))
W, H = 64, 32
FRAME = W * H * 3 // 2


@pytest.fixture
def two_devices(gpu_stream):
    from kornia_rs import hip
    if hip.device_count() < 2:
        pytest.skip("needs two HIP devices (host simulator: KH_HOSTSIM_DEVICES=2)")
    hip.set_device(0)
    yield hip
    hip.set_device(0)


def test_pageable_copies_for_another_device_from_one_thread(two_devices):
    """h2d / d2h through the per-(thread, device) bounce buffer: device 0 and device 1 alternately, large enough for both halves and
    the second trip through each (> 2 x 8 MiB), on created streams and on the null stream of device 1."""
    hip = two_devices
    from kornia_rs.hip import DeviceBuffer, Stream
    a = O.pattern_u8(20 * (1 << 20) + 12345)
    s0, s1 = Stream.new(0), Stream.new(1)
    for stream in (s1, s0, Stream.default(1), s1):
        buf = DeviceBuffer.from_numpy(a, stream)
        assert hip.current_device() == 0                       # the caller's device is restored
        assert buf.device_id == stream.device and hip.pointer_domain(buf.ptr)[1] == stream.device   # allocated where its stream lives
        assert np.array_equal(buf.to_numpy(np.uint8, a.shape), a)
        buf.free()
    assert hip.current_device() == 0
    hip.release_thread_staging()
    hip.release_thread_staging()                               # idempotent


def test_worker_thread_serves_both_devices(two_devices):
    """A pool worker is not pinned to a shard: the same thread copies for device 1, then 0, then 1."""
    from kornia_rs.hip import DeviceBuffer, Stream, release_thread_staging
    a = O.pattern_u8(3 * (1 << 20))
    err = []

    def work():
        try:
            for d in (1, 0, 1):
                b = DeviceBuffer.from_numpy(a, Stream.new(d))
                assert np.array_equal(b.to_numpy(np.uint8, a.shape), a)
            release_thread_staging()
        except BaseException as e:  # noqa: BLE001
            err.append(e)

    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert not err, err


def test_preprocessor_on_device_1_from_a_device_0_thread(two_devices):
    """The upload ring (events, pinned + device slots, copy stream) and the kernels of a preprocessor whose stream lives on device 1,
    called with device 0 current: host batch (staged and zero-copy), single frame, device-resident source."""
    hip = two_devices
    from kornia_rs import Preprocessor, Tensor
    from kornia_rs.hip import DeviceBuffer, PinnedBuffer, Stream
    s1 = Stream.new(1)
    pre = Preprocessor(mode="stretch", format="nv12", stream=s1, **IMAGENET)
    base = O.pattern_u8(FRAME + 31 * 6)
    frames = [base[31 * k: 31 * k + FRAME].copy() for k in range(6)]
    want = np.concatenate([O.preprocess(f, W, H, W, H, fmt="nv12", mode="stretch", **IMAGENET) for f in frames])
    for rnd in range(4):                                       # both ring slots, twice
        dst = Tensor.uninit((6, 3, H, W), "float32", s1)
        pre.run_host_batch(frames, W, H, dst)
        assert hip.current_device() == 0
        assert_same_bits(dst.numpy(), want, f"host batch round {rnd}")
    with hip._device_guard(1):
        cap = PinnedBuffer(6 * FRAME)
    view = cap.view()
    for k in range(6):
        view[k * FRAME:(k + 1) * FRAME] = frames[k]
    dst = Tensor.uninit((6, 3, H, W), "float32", s1)
    pre.run_host_batch([view[k * FRAME:(k + 1) * FRAME] for k in range(6)], W, H, dst, zero_copy=True)
    pre.wait_uploads()
    assert_same_bits(dst.numpy(), want, "zero-copy on device 1")
    out = pre.run(frames[2], W, H, H, W)
    assert out.device_id == 1
    assert_same_bits(out.numpy(), want[2:3], "single frame")
    src = DeviceBuffer.from_numpy(frames[4], s1)
    dst1 = Tensor.uninit((1, 3, H, W), "float32", s1)
    pre.run_raw(src, W, H, dst1)
    assert_same_bits(dst1.numpy(), want[4:5], "device-resident source")
    assert hip.current_device() == 0


def test_imgproc_on_device_1_from_a_device_0_thread(two_devices):
    """Residency dispatch for images that live on device 1: the launch, the scratch (u8 separable resize, Lanczos tables), the fence
    between two streams of device 1 and the readback, with device 0 current in the calling thread."""
    hip = two_devices
    from kornia_rs import Image, Stream, imgproc
    s1, other = Stream.new(1), Stream.new(1)
    f = Image.from_numpy(O.pattern_f32(129 * 97 * 3).reshape(97, 129, 3)).to_hip(s1)
    assert f.device == "cuda:1"
    assert np.array_equal(imgproc.resize(f, (48, 64), "bilinear").numpy(), O.resize(f.numpy(), 64, 48))
    assert np.array_equal(imgproc.gaussian_blur(f, (7, 7), (1.5, 1.5)).numpy(), O.gaussian_blur(f.numpy(), (7, 7), (1.5, 1.5)))
    dst = Image.zeros(129, 97, 3, "float32", stream=other)     # another stream of the same device: fenced in and back
    imgproc.gaussian_blur(f, (5, 5), (1.0, 1.0), dst=dst)
    assert np.array_equal(dst.numpy(), O.gaussian_blur(f.numpy(), (5, 5), (1.0, 1.0)))
    u8 = Image.from_numpy(O.pattern_u8(160 * 120 * 3).reshape(120, 160, 3)).to_hip(s1)
    small = imgproc.resize(u8, (33, 47), "lanczos")
    assert small.device_id == 1 and np.array_equal(small.numpy(), O.resize_fast_u8(u8.numpy(), 47, 33, "lanczos")[0])
    assert hip.current_device() == 0
    # an operand on another DEVICE is a typed error, never a silent peer access (P/cuda/dispatch.rs:51-53)
    g0 = Image.from_numpy(f.numpy()).to_hip(Stream.new(0))
    with pytest.raises(Exception) as e:
        imgproc.gaussian_blur(f, (5, 5), (1.0, 1.0), dst=Image.zeros(129, 97, 3, "float32", stream=g0.stream))
    assert "device" in str(e.value).lower()


def test_stream_fence_between_devices(two_devices):
    """kh_stream_fence(producer on device 1, consumer on device 0): the event is created for the producer's device whatever device
    is current (hipEventRecord refuses another device's event); waiting on it from device 0's stream is allowed."""
    hip = two_devices
    from kornia_rs import _ffi
    from kornia_rs.hip import Stream
    s0, s1 = Stream.new(0), Stream.new(1)
    for cur in (0, 1):
        hip.set_device(cur)
        _ffi.check(_ffi.lib.kh_stream_fence(s1.cuda_stream_ptr, s0.cuda_stream_ptr))
        _ffi.check(_ffi.lib.kh_stream_fence(s0.cuda_stream_ptr, s1.cuda_stream_ptr))
        assert hip.current_device() == cur
    hip.set_device(0)


def test_shard_pool_close_releases_every_worker(two_devices):
    """ShardPool.close: one release task per worker (barrier), every (thread, device) bounce buffer freed, errors not swallowed,
    idempotent."""
    from kornia_rs import hip as H_
    from kornia_rs.sharding import ShardedImgproc
    sp = ShardedImgproc([0, 1, 1])
    imgs = [O.pattern_f32(40 * 30 * 3 + 31 * k)[31 * k:].reshape(30, 40, 3).copy() for k in range(5)]
    held = []
    together = threading.Barrier(3)   # the pool starts its threads on demand: without this a quick first task lets its worker take a second one

    def touch(g):   # every worker copies through its own bounce buffer
        from kornia_rs.hip import DeviceBuffer
        together.wait(timeout=60.0)
        raw = imgs[g].reshape(-1).view(np.uint8)
        b = DeviceBuffer.from_numpy(raw, sp.streams[g])
        assert np.array_equal(b.to_numpy(np.uint8, raw.shape), raw)
        held.append((threading.current_thread().name, dict(H_._stage_local.slots)))
    sp._each(touch)
    assert len({n for n, _ in held}) == 3 and all(slots for _, slots in held)
    sp.close()
    for _, slots in held:
        assert all(buf.ptr is None for buf, _ in slots.values())   # PinnedBuffer.free() ran on the worker that owned it
    sp.close()
