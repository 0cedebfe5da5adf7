"""In-process multi-device sharder: contiguous batch slices, one host thread + one non-default stream per
device, no collective.  Every frame of every shard is compared with the CPU oracle, bit for bit.

Runs on ``min(2, device_count)`` devices; on a one-GPU box the second shard is a second stream + worker thread on the
same device (``devices=[0, 0]``), which exercises the same slicing, staging, threading and per-shard launch code.
"""
import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits

pytestmark = pytest.mark.gpu

IMAGENET = dict(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))
W, H = 64, 32
FRAME = W * H * 3 // 2


def _device_lists():
    from kornia_rs import hip
    n = hip.device_count()
    lists = [[0], [0, 0], [0, 0, 0]]
    if n >= 2:
        lists += [[0, 1], [1, 0, 1]]
    return lists


def _frames(n):
    base = O.pattern_u8(FRAME + 31 * n)
    return np.stack([base[31 * k: 31 * k + FRAME] for k in range(n)])


def _want(frames, ow, oh, mode):
    return np.concatenate([O.preprocess(f, W, H, ow, oh, fmt="nv12", mode=mode, **IMAGENET) for f in frames])


@pytest.mark.parametrize("n_frames", [1, 2, 7])
@pytest.mark.parametrize("geom", [(W, H, "stretch"), (40, 40, "letterbox")])
def test_host_batch_sharded_equals_oracle(gpu_stream, n_frames, geom):
    from kornia_rs import Preprocessor
    ow, oh, mode = geom
    frames = _frames(n_frames)
    want = _want(frames, ow, oh, mode)
    pre = Preprocessor(mode=mode, format="nv12", stream=gpu_stream, **IMAGENET)
    for devices in _device_lists():
        batch = pre.run_raw_batch(frames, W, H, (oh, ow), devices=devices)
        assert len(batch) == n_frames and len(batch.shards) == len(devices)
        # contiguous, balanced, in order
        assert batch.ranges[0][0] == 0 and batch.ranges[-1][1] == n_frames
        assert all(a[1] == b[0] for a, b in zip(batch.ranges, batch.ranges[1:]))
        sizes = [hi - lo for lo, hi in batch.ranges]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
        for g, t in enumerate(batch.shards):
            assert t.device_id == devices[g] and t.stream.device == devices[g]
            assert t.stream.cuda_stream_ptr != 0  # a non-default stream per shard
        assert_same_bits(batch.numpy(), want, f"devices={devices}")


def test_shards_use_distinct_streams_and_threads(gpu_stream):
    import threading
    from kornia_rs import Preprocessor
    pre = Preprocessor(mode="stretch", format="nv12", stream=gpu_stream, **IMAGENET)
    sp = pre.sharded([0, 0])
    assert sp is pre.sharded([0, 0])  # cached per device list
    assert len({s.cuda_stream_ptr for s in sp.streams}) == 2
    names = sp._each(lambda g: threading.current_thread().name)
    assert all(n.startswith("kornia-shard") for n in names) and threading.current_thread().name not in names


def test_device_resident_slices_and_caller_outputs(gpu_stream):
    """Slices already resident on their devices ((buffer, n) per shard) into caller-provided per-shard tensors."""
    from kornia_rs import Preprocessor, Tensor
    from kornia_rs.hip import DeviceBuffer
    from kornia_rs.sharding import shard_range
    frames = _frames(5)
    want = _want(frames, W, H, "stretch")
    pre = Preprocessor(mode="stretch", format="nv12", stream=gpu_stream, **IMAGENET)
    for devices in _device_lists()[1:]:
        sp = pre.sharded(devices)
        parts, outs = [], []
        for g in range(len(devices)):
            lo, hi = shard_range(5, g, len(devices))
            parts.append((DeviceBuffer.from_numpy(frames[lo:hi].reshape(-1), sp.streams[g]), hi - lo))
            outs.append(Tensor.uninit((hi - lo, 3, H, W), "float32", sp.streams[g]))
        batch = pre.run_raw_batch(parts, W, H, outs, frame_stride=FRAME, devices=devices)
        assert batch.shards == outs
        assert_same_bits(batch.numpy(), want, f"devices={devices}")


def test_sharded_f16_and_more_shards_than_frames(gpu_stream):
    from kornia_rs import Preprocessor
    frames = _frames(2)
    pre = Preprocessor(mode="letterbox", format="nv12", f16=True, stream=gpu_stream, **IMAGENET)
    batch = pre.run_raw_batch(frames, W, H, (24, 40), devices=[0, 0, 0])
    assert [hi - lo for lo, hi in batch.ranges] == [1, 1, 0]
    got = batch.numpy().view(np.uint16)
    want = np.concatenate([O.preprocess(f, W, H, 40, 24, fmt="nv12", mode="letterbox", f16=True, **IMAGENET) for f in frames])
    assert np.array_equal(got, want)


def test_sharder_argument_errors(gpu_stream):
    from kornia_rs import Preprocessor, PreprocessError, Tensor, hip
    from kornia_rs.sharding import ShardedPreprocessor
    pre = Preprocessor(mode="stretch", format="nv12", stream=gpu_stream)
    with pytest.raises(ValueError, match="out of range"):
        pre.sharded([hip.device_count()])
    with pytest.raises(ValueError, match="at least one"):
        ShardedPreprocessor([])
    with pytest.raises(ValueError, match="do not pass stream"):
        ShardedPreprocessor([0], stream=gpu_stream)
    with pytest.raises(PreprocessError) as e:  # one destination per device
        pre.run_raw_batch(_frames(2), W, H, [Tensor.uninit((2, 3, H, W), "float32", gpu_stream)], devices=[0, 0])
    assert e.value.kind == "BatchMismatch"
    with pytest.raises(PreprocessError) as e:  # ragged host frames
        pre.run_raw_batch([_frames(1)[0], _frames(1)[0][:-2]], W, H, (H, W), devices=[0, 0])
    assert e.value.kind == "InvalidRawSource"
    with pytest.raises(PreprocessError) as e:  # a shard's destination with the wrong batch size
        sp = pre.sharded([0, 0])
        outs = [Tensor.uninit((2, 3, H, W), "float32", sp.streams[0]), Tensor.uninit((2, 3, H, W), "float32", sp.streams[1])]
        pre.run_raw_batch(_frames(3), W, H, outs, devices=[0, 0])
    assert e.value.kind == "BatchMismatch"


def test_timed_steps_common_barrier(gpu_stream):
    from kornia_rs import Preprocessor
    pre = Preprocessor(mode="stretch", format="nv12", stream=gpu_stream, **IMAGENET)
    sp = pre.sharded([0, 0])
    frames = _frames(4)
    parts, stride = sp.upload(frames)
    outs = sp.alloc_output(4, H, W)
    calls = [0, 0]

    def step(g):
        calls[g] += 1
        sp.shards[g].run_raw_batch(parts[g][0], W, H, outs[g], frame_stride=stride)

    dt = sp.timed_steps(step, steps=5, warmup=2)
    assert calls == [7, 7] and dt > 0
    got = np.concatenate([t.numpy() for t in outs])
    assert_same_bits(got, _want(frames, W, H, "stretch"), "timed_steps")


# ---- ShardedImgproc: any imgproc operator, sharded (round 3) ----------------------------------------------
INTR = (60.0, 62.0, 48.5, 30.25)
DIST = (1.7547749280929563, 0.0097926277667284, -0.027250492945313457, 2.1092164516448975, 0.462927520275116,
        -0.08215277642011642, -0.00005535508171073161, 0.00003768636770639569)   # examples/undistort_image/src/main.rs:30-50
IW, IH = 97, 61
HM = [1.03, 0.05, -3.0 * IW / 129.0, -0.02, 0.97, 4.0 * IH / 97.0, 2.0 / (IH * IW), 1.5 / (IW * IH), 1.0]  # the projective H of bench C5


def _images_f32(n):
    base = O.pattern_f32(IW * IH * 3 + 31 * n)
    return [base[31 * k: 31 * k + IW * IH * 3].reshape(IH, IW, 3).copy() for k in range(n)]


@pytest.mark.parametrize("n_images", [1, 5])
def test_sharded_undistort_warp_equals_oracle(gpu_stream, n_images):
    """BASELINE configs[4] through the product sharder: remap (device-built Brown-Conrady maps, replicated) then warp_perspective,
    every image of every shard bit-equal to the restatement."""
    from kornia_rs.sharding import ShardedImgproc, plan
    imgs = _images_f32(n_images)
    mx, my = O.correction_map(INTR, DIST, IW, IH)
    want = [O.warp_perspective(O.remap(im, mx, my), HM, IW, IH) for im in imgs]
    for devices in _device_lists():
        sp = ShardedImgproc(devices)
        batch = sp.scatter(imgs)
        assert batch.ranges == plan(n_images, len(devices)) and len(batch) == n_images
        out = sp.undistort_warp(batch, INTR, DIST, HM)
        for g, shard in enumerate(out.shards):
            assert all(im.is_device and im.device_id == devices[g] and im.stream is sp.streams[g] for im in shard)
        got = out.numpy()
        assert len(got) == n_images
        for k in range(n_images):
            assert_same_bits(got[k], want[k], f"devices={devices} image {k}")
        sp.close()


def test_sharded_map_of_other_operators(gpu_stream):
    """`map` takes any imgproc operator: a colour map, a separable filter, a u8 gather with a replicated host operand."""
    from kornia_rs import imgproc
    from kornia_rs.sharding import ShardedImgproc
    n = 4
    base = O.pattern_u8(IW * IH * 3 + 31 * n)
    u8 = [base[31 * k: 31 * k + IW * IH * 3].reshape(IH, IW, 3).copy() for k in range(n)]
    f32 = _images_f32(n)
    mx, my = O.correction_map(INTR, DIST, IW, IH)
    sp = ShardedImgproc([0, 0, 0])
    gray = sp.map(imgproc.gray_from_rgb, sp.scatter(u8)).numpy()
    blur = sp.map(imgproc.gaussian_blur, sp.scatter(f32), (7, 7), (1.5, 1.5)).numpy()
    rm = sp.map(imgproc.remap, sp.scatter(u8), sp.replicate(mx.reshape(IH, IW, 1)), sp.replicate(my.reshape(IH, IW, 1))).numpy()
    for k in range(n):
        assert np.array_equal(gray[k].reshape(IH, IW), O.gray_from_rgb_u8(u8[k]).reshape(IH, IW)), k
        assert_same_bits(blur[k], O.gaussian_blur(f32[k], (7, 7), (1.5, 1.5)), f"blur {k}")
        assert np.array_equal(rm[k], O.remap_u8(u8[k], mx, my)), k
    with pytest.raises(ValueError, match="scattered over devices"):
        ShardedImgproc([0]).map(imgproc.gray_from_rgb, sp.scatter(u8))
    sp.close()


def test_a_failing_shard_surfaces_its_error_and_never_hangs_the_timer(gpu_stream):
    from kornia_rs.sharding import ShardedImgproc
    sp = ShardedImgproc([0, 0])

    def step(g):
        if g == 1:
            raise RuntimeError("boom on shard 1")
    with pytest.raises(RuntimeError, match="boom on shard 1|another shard failed"):
        sp.timed_steps(step, steps=2, warmup=1)
    assert sp.timed_steps(lambda g: None, steps=2, warmup=1) >= 0.0  # the pool is still usable
    sp.close()
