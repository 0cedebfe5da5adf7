"""Fused per-pixel pipelines: CPU pins of the oracle against the reference's own `cpu_reference`
(P/cuda/fusion.rs:700-750, restated in numpy) and GPU parity / API tests after its device tests
(:752-960)."""
import numpy as np
import pytest

import oracle_ffi as O

IMAGENET_SCALE = [1.0 / 255.0 / 0.229, 1.0 / 255.0 / 0.224, 1.0 / 255.0 / 0.225]
IMAGENET_BIAS = [-0.485 / 0.229, -0.456 / 0.224, -0.406 / 0.225]


def cpu_reference(src, dw, dh, scale, bias, gray):
    """fusion.rs:703-750 in vectorised f32 numpy."""
    f = np.float32
    sh, sw, _ = src.shape
    ax, ay = f(sw) / f(dw), f(sh) / f(dh)
    bx, by = f(0.5) * ax - f(0.5), f(0.5) * ay - f(0.5)
    sxf = np.maximum(ax * np.arange(dw, dtype=f) + bx, f(0))
    syf = np.maximum(ay * np.arange(dh, dtype=f) + by, f(0))
    sx0, sy0 = np.minimum(sxf.astype(np.int64), sw - 1), np.minimum(syf.astype(np.int64), sh - 1)
    sx1, sy1 = np.minimum(sx0 + 1, sw - 1), np.minimum(sy0 + 1, sh - 1)
    wx, wy = (sxf - sx0.astype(f))[None, :, None], (syf - sy0.astype(f))[:, None, None]
    p = src.astype(f)
    v = ((f(1) - wy) * (f(1) - wx) * p[sy0][:, sx0] + (f(1) - wy) * wx * p[sy0][:, sx1]
         + wy * (f(1) - wx) * p[sy1][:, sx0] + wy * wx * p[sy1][:, sx1])
    n = v * np.array(scale, f) + np.array(bias, f)
    if gray:
        return (f(0.299) * n[..., 0] + f(0.587) * n[..., 1] + f(0.114) * n[..., 2])[None]
    return n.transpose(2, 0, 1)


def close(got, want):
    return np.all(np.abs(got - want) <= 1e-4 * np.maximum(np.abs(want), 1.0))  # the reference's tolerance


def test_oracle_matches_reference_cpu_form():
    src = O.pattern_u8(129 * 97 * 3).reshape(97, 129, 3)
    got = O.fused_pipeline(src, 64, 48, [("normalize", IMAGENET_SCALE, IMAGENET_BIAS)], "chw")
    assert close(got, cpu_reference(src, 64, 48, IMAGENET_SCALE, IMAGENET_BIAS, False))
    src = O.pattern_u8(100 * 80 * 3).reshape(80, 100, 3)
    got = O.fused_pipeline(src, 47, 33, [("normalize", [1 / 255.0] * 3, [0.0] * 3), ("gray",)], "c1")
    assert close(got, cpu_reference(src, 47, 33, [1 / 255.0] * 3, [0.0] * 3, True))
    # no maps: the raw bilinear value in [0, 255]; an identity grid returns the source bytes
    ident = O.fused_pipeline(src, 100, 80, [], "chw")
    assert np.array_equal(ident, src.astype(np.float32).transpose(2, 0, 1))


# ---- GPU --------------------------------------------------------------------------------------------------------

def _dev_tensor(gpu_stream, a):
    from kornia_rs import Tensor
    return Tensor.from_numpy(np.ascontiguousarray(a)).to_hip(gpu_stream)


@pytest.mark.gpu
def test_fused_pipelines_match_oracle(gpu_stream):  # fusion.rs:752-830
    from kornia_rs import Tensor
    from kornia_rs.fusion import FusedPipeline, Normalize, ReadU8RgbBilinear, RgbToGray, WriteC1F32, WriteChwF32
    from gpu_util import assert_same_bits
    cases = [((129, 97), (64, 48), [("normalize", IMAGENET_SCALE, IMAGENET_BIAS)], "chw"),
             ((100, 80), (47, 33), [("normalize", [1 / 255.0] * 3, [0.0] * 3), ("gray",)], "c1"),
             ((63, 41), (127, 90), [], "chw"),
             ((33, 21), (33, 21), [("gray",), ("normalize", [2.0, 3.0, 4.0], [0.5, -0.5, 1.0]), ("gray",)], "chw"),
             ((1, 1), (5, 4), [("normalize", [0.5] * 3, [1.0] * 3)], "c1"),
             ((1920, 1080), (640, 640), [("normalize", [1 / 255.0] * 3, [0.0] * 3)], "chw")]
    for (sw, sh), (dw, dh), maps, sink in cases:
        src = O.pattern_u8(sw * sh * 3).reshape(sh, sw, 3)
        stages = [ReadU8RgbBilinear(sw, sh, dw, dh)]
        stages += [Normalize(m[1], m[2]) if m[0] == "normalize" else RgbToGray() for m in maps]
        stages.append(WriteChwF32() if sink == "chw" else WriteC1F32())
        pipe = FusedPipeline.build(stages, dw, dh)
        d_src = _dev_tensor(gpu_stream, src)
        d_dst = Tensor.uninit((3 if sink == "chw" else 1, dh, dw), "float32", gpu_stream)
        pipe.launch(gpu_stream, d_src, d_dst)
        want = O.fused_pipeline(src, dw, dh, maps, sink)
        assert_same_bits(d_dst.numpy(), want, f"fused {sw}x{sh}->{dw}x{dh} {[m[0] for m in maps]} {sink}")
        text = pipe.generated_source()
        assert "read_u8rgb_bilinear" in text and ("write_chw_f32" if sink == "chw" else "write_c1_f32") in text


@pytest.mark.gpu
def test_batched_matches_single_launches_and_errors(gpu_stream):  # fusion.rs:905-1010
    from kornia_rs import Tensor
    from kornia_rs.fusion import FusedPipeline, FusionError, Normalize, ReadU8RgbBilinear, RgbToGray, WriteChwF32
    from gpu_util import assert_same_bits
    sw, sh, dw, dh = 129, 97, 64, 48
    out_elems = 3 * dw * dh
    stages = [ReadU8RgbBilinear(sw, sh, dw, dh), Normalize([1 / 255.0] * 3, [-0.5] * 3), WriteChwF32()]
    n = 40  # more images than one launch carries pointers for
    hosts = [((O.pattern_u8(sw * sh * 3).astype(np.uint16) + 37 * i) % 256).astype(np.uint8).reshape(sh, sw, 3) for i in range(n)]
    batched = FusedPipeline.build_batched(stages, dw, dh, n, out_elems + 16)  # padded per-image stride
    srcs = [_dev_tensor(gpu_stream, h) for h in hosts]
    d_dst = Tensor.zeros((n, out_elems + 16), "float32", gpu_stream)
    batched.launch_batched(gpu_stream, srcs, d_dst)
    got = d_dst.numpy()
    for i in range(n):
        want = O.fused_pipeline(hosts[i], dw, dh, [("normalize", [1 / 255.0] * 3, [-0.5] * 3)], "chw")
        assert_same_bits(got[i, :out_elems].reshape(3, dh, dw), want, f"batched image {i}")
        assert (got[i, out_elems:] == 0).all()
    with pytest.raises(FusionError) as e:
        FusedPipeline.build([ReadU8RgbBilinear(8, 8, 4, 4)], 4, 4)
    assert e.value.kind == "Pipeline" and "source and a sink" in str(e.value)
    with pytest.raises(FusionError) as e:
        FusedPipeline.build([Normalize([1] * 3, [0] * 3), WriteChwF32()], 4, 4)
    assert e.value.kind == "Pipeline"
    with pytest.raises(FusionError) as e:
        FusedPipeline.build_batched(stages, dw, dh, 4, out_elems - 1)
    assert e.value.kind == "Pipeline" and "out_elems_per_image" in str(e.value)
    with pytest.raises(FusionError) as e:
        FusedPipeline.build([ReadU8RgbBilinear(8, 8, 4, 4)] + [RgbToGray()] * 13 + [WriteChwF32()], 4, 4)
    assert e.value.kind == "ParamsTooLarge"
    with pytest.raises(FusionError) as e:  # wrong number of sources
        batched.launch_batched(gpu_stream, srcs[:3], d_dst)
    assert e.value.kind == "Pipeline" and "built for batch" in str(e.value)
    single = FusedPipeline.build(stages, dw, dh)
    with pytest.raises(FusionError):  # destination too short
        single.launch(gpu_stream, srcs[0], Tensor.uninit((out_elems - 1,), "float32", gpu_stream))
    with pytest.raises(FusionError):  # source too short
        single.launch(gpu_stream, _dev_tensor(gpu_stream, hosts[0][:-1]), d_dst)
