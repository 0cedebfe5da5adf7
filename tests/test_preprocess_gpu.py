"""GPU parity: fused preprocess HIP kernels vs the CPU oracle, through the C ABI.

Bar: bit-exact for nearest / bilinear (integer decode + uncontracted f32, same expression tree
as crates/kornia-imgproc/src/preprocess.rs:430-622), f16 bit-exact, Lanczos bit-exact too since round 3 (the
axis weights are built per geometry with the host's libm sinf — the restatement's and the reference CPU side's function —
instead of the device sinf that left round 2 at 2e-4; the reference itself only compares Lanczos loosely, preprocess.rs:1685).
Also restates the reference's own GPU tests for this kernel (preprocess.rs:1560-1939).
"""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

IMAGENET = dict(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))


def _pre(stream, **kw):
    from kornia_rs import Preprocessor
    return Preprocessor(stream=stream, **kw)


def _run(stream, raw, w, h, dw, dh, *, fmt, mode="letterbox", sampling="bilinear", f16=False,
         mean=None, std=None, pad_value=114, force_generic=False):
    from kornia_rs import Tensor
    from kornia_rs.hip import DeviceBuffer
    pre = _pre(stream, mode=mode, format=fmt, sampling=sampling, f16=f16, mean=mean, std=std,
               pad_value=pad_value)
    src = DeviceBuffer.from_numpy(np.ascontiguousarray(raw).reshape(-1), stream)
    dst = Tensor.uninit((1, 3, dh, dw), "float16" if f16 else "float32", stream)
    pre.run_raw(src, w, h, dst, _force_generic=force_generic) if fmt in ("nv12", "yuyv", "gray") else \
        pre.run_surface(src, w, h, w * (4 if fmt in ("rgba", "bgra") else 3),
                        4 if fmt in ("rgba", "bgra") else 3, dst)
    out = dst.numpy_raw()
    return out.view(np.uint16) if f16 else out


def _raw_for(fmt, w, h, seed=0):
    n = {"rgb": 3 * w * h, "bgr": 3 * w * h, "rgba": 4 * w * h, "bgra": 4 * w * h, "gray": w * h,
         "nv12": w * h * 3 // 2, "yuyv": 2 * w * h}[fmt]
    return np.roll(O.pattern_u8(n + seed), -seed)[:n].copy()


def _assert_bits_equal(got, want, what):
    g = got.view(np.uint32) if got.dtype == np.float32 else got
    w_ = want.view(np.uint32) if want.dtype == np.float32 else want
    bad = np.nonzero(g != w_)
    assert bad[0].size == 0, f"{what}: {bad[0].size} mismatching elements, first at {tuple(b[0] for b in bad)}: " \
                             f"{got[tuple(b[0] for b in bad)]} vs {want[tuple(b[0] for b in bad)]}"


@pytest.mark.parametrize("fmt", ["rgb", "bgr", "rgba", "bgra", "gray", "nv12", "yuyv"])
@pytest.mark.parametrize("mode", ["letterbox", "stretch"])
@pytest.mark.parametrize("sampling", ["nearest", "bilinear"])
@pytest.mark.parametrize("f16", [False, True])
def test_matches_oracle_bit_exact(gpu_stream, fmt, mode, sampling, f16):
    for (w, h, dw, dh) in [(46, 34, 31, 27), (22, 18, 57, 41), (8, 6, 7, 5)]:
        raw = _raw_for(fmt, w, h)
        got = _run(gpu_stream, raw, w, h, dw, dh, fmt=fmt, mode=mode, sampling=sampling, f16=f16, **IMAGENET)
        want = O.preprocess(raw, w, h, dw, dh, fmt=fmt, mode=mode, sampling=sampling, f16=f16, **IMAGENET)
        _assert_bits_equal(got, want, f"{fmt}/{mode}/{sampling}/f16={f16} {w}x{h}->{dw}x{dh}")


@pytest.mark.parametrize("fmt", ["rgb", "bgr", "rgba", "bgra", "nv12", "yuyv", "gray"])
@pytest.mark.parametrize("mode", ["letterbox", "stretch"])
def test_lanczos_matches_oracle_bit_for_bit(gpu_stream, fmt, mode):
    """Lanczos-3 through the row-window loader (rows of >= 8 pixels: one or two wide loads per window row, taps picked by shifts,
    which is also the border replication) and through the per-tap loads narrower sources take; up- and down-scaling so windows
    hang over every edge.  Bit-exact: the weights come from host-built tables (libm sinf), north_star's 1e-6 is met with room."""
    for (w, h, dw, dh) in [(46, 34, 31, 27), (8, 6, 21, 17), (6, 4, 9, 7), (64, 10, 16, 20)]:
        raw = _raw_for(fmt, w, h)
        got = _run(gpu_stream, raw, w, h, dw, dh, fmt=fmt, mode=mode, sampling="lanczos")
        want = O.preprocess(raw, w, h, dw, dh, fmt=fmt, mode=mode, sampling="lanczos")
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (fmt, mode, w, h, dw, dh, float(np.abs(got - want).max()))


@pytest.mark.parametrize("sampling", ["nearest", "bilinear", "lanczos"])
def test_solid_all_sampling(gpu_stream, sampling):  # preprocess.rs:1560-1590
    src = np.tile(np.array((10, 20, 30), np.uint8), (3, 5, 1))
    out = _run(gpu_stream, src, 5, 3, 4, 4, fmt="rgb", mode="stretch", sampling=sampling)[0]
    for c, v in enumerate((10.0, 20.0, 30.0)):
        assert np.abs(out[c] - v / 255.0).max() < 1e-4


def test_pitched_surface_matches_tight(gpu_stream):  # preprocess.rs:1596-1641
    from kornia_rs import Tensor
    from kornia_rs.hip import DeviceBuffer
    w, h, pitch = 23, 17, 23 * 4 + 13
    tight = O.pattern_u8(w * h * 4).reshape(h, w * 4)
    pitched = np.full((h, pitch), 0xAA, np.uint8)
    pitched[:, : w * 4] = tight
    for mode in ("letterbox", "stretch"):
        pre = _pre(gpu_stream, mode=mode, format="rgba", **IMAGENET)
        a = Tensor.uninit((1, 3, 6, 8), "float32", gpu_stream)
        b = Tensor.uninit((1, 3, 6, 8), "float32", gpu_stream)
        pre.run_surface(DeviceBuffer.from_numpy(tight, gpu_stream), w, h, w * 4, 4, a)
        pre.run_surface(DeviceBuffer.from_numpy(pitched, gpu_stream), w, h, pitch, 4, b)
        assert np.array_equal(a.numpy(), b.numpy())


@pytest.mark.parametrize("sampling", ["nearest", "bilinear", "lanczos"])
def test_f16_matches_f32_rounded(gpu_stream, sampling):  # preprocess.rs:1646-1680
    w, h = 23, 17
    raw = _raw_for("rgb", w, h)
    f32 = _run(gpu_stream, raw, w, h, 8, 6, fmt="rgb", sampling=sampling, **IMAGENET)
    f16 = _run(gpu_stream, raw, w, h, 8, 6, fmt="rgb", sampling=sampling, f16=True, **IMAGENET)
    assert np.array_equal(f32.astype(np.float16).view(np.uint16), f16)


def test_f16_overflow_quirk(gpu_stream):
    """|v| >= 2^16 follows the reference's manual f2h (NaN pattern unless the mantissa is 0)."""
    src = np.tile(np.array((255, 128, 0), np.uint8), (4, 4, 1))
    kw = dict(fmt="rgb", mode="stretch", mean=(0.0, 0.0, 0.0), std=(1e-6, 2.0 ** -17, 1.0), f16=True)
    got = _run(gpu_stream, src, 4, 4, 4, 4, **kw)
    want = O.preprocess(src, 4, 4, 4, 4, **kw)
    assert np.array_equal(got, want)
    assert want[0, 0, 0, 0] == 0x7E00  # 1e6 is finite in f32 but not representable: NaN pattern


@pytest.mark.parametrize("sampling", ["nearest", "bilinear", "lanczos"])
@pytest.mark.parametrize("fmt", ["nv12", "yuyv", "gray", "bgr"])
def test_fused_formats_match_chained(gpu_stream, fmt, sampling):  # preprocess.rs:1777-1848
    w, h = 8, 6
    raw = np.array([(i * 7 + 13) % 251 for i in range({"nv12": w * h * 3 // 2, "yuyv": w * h * 2,
                                                         "gray": w * h, "bgr": w * h * 3}[fmt])], np.uint8)
    rgb = {"nv12": lambda: O.rgb_from_nv12(raw, w, h), "yuyv": lambda: O.rgb_from_yuyv(raw, w, h),
           "gray": lambda: np.repeat(raw.reshape(h, w, 1), 3, axis=2),
           "bgr": lambda: raw.reshape(h, w, 3)[:, :, ::-1]}[fmt]()
    fused = _run(gpu_stream, raw, w, h, 7, 5, fmt=fmt, sampling=sampling)
    chained = _run(gpu_stream, rgb, w, h, 7, 5, fmt="rgb", sampling=sampling)
    assert np.abs(fused - chained).max() <= 1e-6


def test_run_raw_batch_matches_single(gpu_stream):  # preprocess.rs:1852-1893
    from kornia_rs import PreprocessError, Tensor
    from kornia_rs.hip import DeviceBuffer
    w, h = 8, 6
    n = w * h * 3 // 2
    base = np.array([(i * 7 + 13) % 251 for i in range(n)], np.uint32)
    raws = [((base + k * 31) & 0xFF).astype(np.uint8) for k in range(3)]
    pre = _pre(gpu_stream, format="nv12")
    bufs = [DeviceBuffer.from_numpy(r, gpu_stream) for r in raws]
    batch = Tensor.uninit((3, 3, 5, 7), "float32", gpu_stream)
    pre.run_raw_batch(bufs, w, h, batch)  # separate allocations: arbitrary strides
    got = batch.numpy()
    for k in range(3):
        one = Tensor.uninit((1, 3, 5, 7), "float32", gpu_stream)
        pre.run_raw(bufs[k], w, h, one)
        assert np.array_equal(got[k], one.numpy()[0]), f"frame {k}"
    # contiguous frames -> one batched launch; same answer
    packed = DeviceBuffer.from_numpy(np.concatenate(raws), gpu_stream)
    batch2 = Tensor.uninit((3, 3, 5, 7), "float32", gpu_stream)
    pre.run_raw_batch(packed, w, h, batch2, frame_stride=n)
    assert np.array_equal(batch2.numpy(), got)
    bad = Tensor.uninit((2, 3, 5, 7), "float32", gpu_stream)
    with pytest.raises(PreprocessError) as e:
        pre.run_raw_batch(bufs, w, h, bad)
    assert e.value.kind == "BatchMismatch" and e.value.fields == {"dst_n": 2, "frames": 3}


def test_run_raw_validates_source(gpu_stream):  # preprocess.rs:1897-1937
    from kornia_rs import PreprocessError, Tensor
    from kornia_rs.hip import DeviceBuffer
    pre = _pre(gpu_stream, format="nv12")
    dst = Tensor.uninit((1, 3, 4, 4), "float32", gpu_stream)
    short = DeviceBuffer.from_numpy(np.zeros(60, np.uint8), gpu_stream)
    with pytest.raises(PreprocessError) as e:
        pre.run_raw(short, 8, 6, dst)
    assert e.value.kind == "InvalidRawSource" and e.value.fields["need"] == 72
    odd = DeviceBuffer.from_numpy(np.zeros(8 * 5 * 2, np.uint8), gpu_stream)
    with pytest.raises(PreprocessError) as e:
        pre.run_raw(odd, 8, 5, dst)
    assert e.value.kind == "InvalidRawSource"
    surf = DeviceBuffer.from_numpy(np.zeros(8 * 6 * 4, np.uint8), gpu_stream)
    with pytest.raises(PreprocessError) as e:
        pre.run_surface(surf, 8, 6, 32, 4, dst)
    assert e.value.kind == "FormatNeedsRawBuffer"
    host = Tensor.zeros((1, 3, 4, 4), "float32")
    with pytest.raises(PreprocessError) as e:
        pre.run_raw(odd, 8, 6, host)
    assert e.value.kind == "NotDeviceTensor"


# ---- the north-star variant ---------------------------------------------------------------------

@pytest.mark.parametrize("w,h", [(64, 32), (1920, 1080), (20, 6), (4, 2)])
@pytest.mark.parametrize("sampling", ["bilinear", "nearest"])
def test_nv12_identity_fast_path_equals_generic_and_oracle(gpu_stream, w, h, sampling):
    import ctypes as C
    from kornia_rs import _ffi
    raw = _raw_for("nv12", w, h, seed=3)
    kw = dict(fmt="nv12", mode="stretch", sampling=sampling, **IMAGENET)
    fast = _run(gpu_stream, raw, w, h, w, h, **kw)
    slow = _run(gpu_stream, raw, w, h, w, h, force_generic=True, **kw)
    want = O.preprocess(raw, w, h, w, h, **kw)
    _assert_bits_equal(fast, want, "fast path vs oracle")
    _assert_bits_equal(slow, want, "generic vs oracle")
    # and the dispatcher really picks the specialised kernel for this geometry
    pre = _pre(gpu_stream, mode="stretch", format="nv12", sampling=sampling, **IMAGENET)
    p = pre._params(w, h, w, 1, _ffi.KH_FMT_NV12, w, h, 1, 0, False, False)
    assert _ffi.lib.kh_preprocess_variant(C.byref(p)) == b"nv12_identity"
    p.flags = _ffi.KH_PRE_FORCE_GENERIC   # (scale 1: every bilinear tap sits on a whole pixel — the generic kernel's one-tap form)
    assert _ffi.lib.kh_preprocess_variant(C.byref(p)) == (b"generic_bilinear_on_grid" if sampling == "bilinear" else b"generic")


def test_nv12_identity_batch_1080p(gpu_stream):
    """Batched launch at the BASELINE frame size: every frame equals its single-frame result,
    and a D2H checksum-of-frames matches the oracle on two sampled frames."""
    from kornia_rs import Tensor
    from kornia_rs.hip import DeviceBuffer
    w, h, n = 1920, 1080, 8
    fb = w * h * 3 // 2
    base = O.pattern_u8(fb + 31 * n)
    frames = np.stack([base[31 * k: 31 * k + fb] for k in range(n)])
    pre = _pre(gpu_stream, mode="stretch", format="nv12", **IMAGENET)
    src = DeviceBuffer.from_numpy(frames.reshape(-1), gpu_stream)
    dst = Tensor.uninit((n, 3, h, w), "float32", gpu_stream)
    pre.run_raw_batch(src, w, h, dst, frame_stride=fb)
    got = dst.numpy()
    gen = Tensor.uninit((n, 3, h, w), "float32", gpu_stream)
    pre.run_raw_batch(src, w, h, gen, frame_stride=fb, _force_generic=True)
    assert np.array_equal(got.view(np.uint32), gen.numpy().view(np.uint32))
    for k in (0, n - 1):
        want = O.preprocess(frames[k], w, h, w, h, fmt="nv12", mode="stretch", **IMAGENET)[0]
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), f"frame {k}"


@pytest.mark.parametrize("lut", [-1, 0])   # production: the per-block table of binary16 results (round 6); 0: the arithmetic kernel
@pytest.mark.parametrize("w,h", [(64, 32), (1920, 1080), (24, 6), (8, 2), (20, 6), (4104, 2)])
@pytest.mark.parametrize("sampling", ["bilinear", "nearest"])
def test_nv12_identity_f16_fast_path_equals_generic_and_oracle(gpu_stream, dev_option, w, h, sampling, lut):
    """run_raw_f16 on the north-star geometry (P/preprocess.rs:1234-1256): since round 6 its own kernel — eight pixels per thread, three
    16-byte stores of eight binary16 values — where the width is a multiple of eight; the oracle's bits, the generic kernel's bits
    (forced), and a width of 20 (not a multiple of eight) keeps the generic kernel."""
    import ctypes as C
    from kornia_rs import _ffi
    dev_option("pre_f16_lut", lut)
    raw = _raw_for("nv12", w, h, seed=5)
    kw = dict(fmt="nv12", mode="stretch", sampling=sampling, f16=True, **IMAGENET)
    fast = _run(gpu_stream, raw, w, h, w, h, **kw)
    slow = _run(gpu_stream, raw, w, h, w, h, force_generic=True, **kw)
    want = O.preprocess(raw, w, h, w, h, **kw)
    _assert_bits_equal(fast, want, "f16 fast path vs oracle")
    _assert_bits_equal(slow, want, "f16 generic vs oracle")
    pre = _pre(gpu_stream, mode="stretch", format="nv12", sampling=sampling, f16=True, **IMAGENET)
    p = pre._params(w, h, w, 1, _ffi.KH_FMT_NV12, w, h, 1, 0, True, False)
    assert _ffi.lib.kh_preprocess_variant(C.byref(p)) == (b"nv12_identity_f16" if w % 8 == 0 else (b"generic_bilinear_on_grid" if sampling == "bilinear" else b"generic"))


def test_nv12_identity_f16_batch_strided_and_list(gpu_stream):
    """A batch of 1080p frames into binary16 planes, equally spaced and as a list of separately allocated frames: every frame equals the
    generic kernel's result, two sampled frames equal the oracle, and the f16 overflow quirk (values >= 2^16 become NaN patterns, not Inf)
    survives the fast path."""
    from kornia_rs import Tensor
    from kornia_rs.hip import DeviceBuffer
    w, h, n = 1920, 1080, 6
    fb = w * h * 3 // 2
    base = O.pattern_u8(fb + 31 * n)
    frames = np.stack([base[31 * k: 31 * k + fb] for k in range(n)])
    pre = _pre(gpu_stream, mode="stretch", format="nv12", f16=True, **IMAGENET)
    src = DeviceBuffer.from_numpy(frames.reshape(-1), gpu_stream)
    dst = Tensor.uninit((n, 3, h, w), "float16", gpu_stream)
    pre.run_raw_batch(src, w, h, dst, frame_stride=fb)
    got = dst.numpy_raw().view(np.uint16)
    gen = Tensor.uninit((n, 3, h, w), "float16", gpu_stream)
    pre.run_raw_batch(src, w, h, gen, frame_stride=fb, _force_generic=True)
    assert np.array_equal(got, gen.numpy_raw().view(np.uint16))
    for k in (0, n - 1):
        want = O.preprocess(frames[k], w, h, w, h, fmt="nv12", mode="stretch", f16=True, **IMAGENET)[0]
        assert np.array_equal(got[k], want.view(np.uint16)), f"frame {k}"
    bufs = [DeviceBuffer.from_numpy(frames[k], gpu_stream) for k in range(n)]
    lst = Tensor.uninit((n, 3, h, w), "float16", gpu_stream)
    pre.run_raw_batch(bufs, w, h, lst)
    assert np.array_equal(lst.numpy_raw().view(np.uint16), got)
    big = _pre(gpu_stream, mode="stretch", format="nv12", f16=True, mean=(0.0, 0.0, 0.0), std=(1e-6, 2.0 ** -17, 1.0))
    raw = _raw_for("nv12", 64, 8, seed=1)
    q = Tensor.uninit((1, 3, 8, 64), "float16", gpu_stream)
    big.run_raw(DeviceBuffer.from_numpy(raw, gpu_stream), 64, 8, q)
    want = O.preprocess(raw, 64, 8, 64, 8, fmt="nv12", mode="stretch", f16=True, mean=(0.0, 0.0, 0.0), std=(1e-6, 2.0 ** -17, 1.0))
    assert np.array_equal(q.numpy_raw().view(np.uint16), want.view(np.uint16))


def test_unaligned_nv12_identity_falls_back_to_generic_kernel(gpu_stream):
    """Width not divisible by 4: the dispatcher must use the generic kernel (still on device) and
    stay bit-exact."""
    w, h = 22, 10
    raw = _raw_for("nv12", w, h)
    got = _run(gpu_stream, raw, w, h, w, h, fmt="nv12", mode="stretch")
    want = O.preprocess(raw, w, h, w, h, fmt="nv12", mode="stretch")
    _assert_bits_equal(got, want, "22x10 identity")


def test_python_run_api(gpu_stream):
    """kornia_rs.Preprocessor.run(frame, w, h, oh, ow) with numpy frames (uploads on its stream)."""
    w, h = 32, 16
    raw = _raw_for("nv12", w, h)
    pre = _pre(gpu_stream, mode="letterbox", format="nv12", **IMAGENET)
    out = pre.run(raw, w, h, 24, 24)
    assert out.shape == (1, 3, 24, 24) and out.device == "cuda:0" and out.dtype == "float32"
    want = O.preprocess(raw, w, h, 24, 24, fmt="nv12", **IMAGENET)
    assert np.array_equal(out.numpy(), want)
    outs = pre.run([raw, raw[::-1].copy()], w, h, 24, 24)
    assert outs.shape == (2, 3, 24, 24)
    assert np.array_equal(outs.numpy()[0], want[0])
    cai = out.__cuda_array_interface__
    assert cai["shape"] == (1, 3, 24, 24) and cai["typestr"] == "<f4" and cai["data"][0] == out.data_ptr


@pytest.mark.parametrize("fmt", ["nv12", "rgb", "yuyv"])
def test_quotient_shortcut_equals_ieee_division_path(gpu_stream, fmt, dev_option):
    """The generic kernel replaces (o - pad) / scale by a host-verified 3-op quotient; the test option pre_ieee_div = 1
    forces the plain divisions.  Both must give the oracle's bits on awkward scales."""
    for (w, h), (dw, dh), mode in [((1920, 1080), (640, 640), "letterbox"), ((130, 98), (97, 55), "stretch"),
                                   ((64, 48), (333, 171), "letterbox"), ((258, 194), (224, 224), "stretch")]:
        raw = _raw_for(fmt, w, h, seed=5)
        kw = dict(fmt=fmt, mode=mode, sampling="bilinear", **IMAGENET)
        want = O.preprocess(raw, w, h, dw, dh, **kw)
        dev_option("pre_ieee_div", -1)
        _assert_bits_equal(_run(gpu_stream, raw, w, h, dw, dh, **kw), want, f"{fmt} {w}x{h}->{dw}x{dh} shortcut")
        dev_option("pre_ieee_div", 1)
        _assert_bits_equal(_run(gpu_stream, raw, w, h, dw, dh, **kw), want, f"{fmt} {w}x{h}->{dw}x{dh} ieee")


@pytest.mark.parametrize("fmt", ["nv12", "rgb", "bgra", "yuyv", "gray"])
@pytest.mark.parametrize("f16", [False, True])
def test_bilinear_on_whole_pixel_grid_equals_the_four_tap_kernel_and_the_oracle(gpu_stream, fmt, f16, dev_option):
    """When every source coordinate of a launch is a whole number (1080p -> 640 letterbox: sx = 3 ox exactly) the bilinear weights are 0
    and the kernel decodes one tap per pixel; the test option pre_grid = 0 keeps the four-tap kernel.  Both must give the restatement's bits — on
    such geometries, on near misses (a fractional pad, scale 1/5 whose f32 quotients are not all whole) and on an upscale."""
    import ctypes as C
    from kornia_rs import _ffi
    cases = [((1920, 1080), (640, 640), "letterbox", True), ((96, 64), (48, 32), "stretch", True), ((90, 60), (30, 20), "stretch", True),
             ((96, 64), (48, 40), "letterbox", True), ((90, 60), (31, 20), "letterbox", False), ((100, 50), (20, 10), "stretch", None),
             ((20, 10), (40, 20), "stretch", False), ((46, 34), (31, 27), "stretch", False)]
    for (w, h), (dw, dh), mode, on_grid in cases:
        raw = _raw_for(fmt, w, h, seed=3)
        kw = dict(fmt=fmt, mode=mode, sampling="bilinear", f16=f16, **IMAGENET)
        want = O.preprocess(raw, w, h, dw, dh, **kw)
        dev_option("pre_grid", -1)
        if on_grid is not None and fmt == "nv12" and not f16:
            pre = _pre(gpu_stream, mode=mode, format=fmt, sampling="bilinear", **IMAGENET)
            p = pre._params(w, h, w, 1, _ffi.KH_FMT_NV12, dw, dh, 1, 0, False, False)
            assert _ffi.lib.kh_preprocess_variant(C.byref(p)) == (b"generic_bilinear_on_grid" if on_grid else b"generic"), (w, h, dw, dh, mode)
        _assert_bits_equal(_run(gpu_stream, raw, w, h, dw, dh, **kw), want, f"{fmt} {w}x{h}->{dw}x{dh} {mode} default")
        dev_option("pre_grid", 0)
        _assert_bits_equal(_run(gpu_stream, raw, w, h, dw, dh, **kw), want, f"{fmt} {w}x{h}->{dw}x{dh} {mode} four taps")


def test_python_run_reuses_pinned_staging_in_a_frame_loop(gpu_stream):
    """PY/cuda_ext/mod.rs:647-745: host frames go through PERSISTENT page-locked + device buffers (grown on demand, never per call)
    — here a two-deep ring on a copy stream — and a slot's previous upload is waited before its pinned bytes are reused, so
    back-to-back calls with different frames stay correct."""
    w, h = 64, 32
    pre = _pre(gpu_stream, mode="letterbox", format="nv12", **IMAGENET)
    frames = [_raw_for("nv12", w, h, seed=7 * k) for k in range(7)]
    outs = [pre.run(fr, w, h, 24, 40) for fr in frames]          # no sync between calls: slots 0, 1, 0, 1, ...
    assert pre._staging.allocations == 4                          # one pinned + one device allocation per ring slot, in total
    for fr, out in zip(frames, outs):
        want = O.preprocess(fr, w, h, 40, 24, fmt="nv12", mode="letterbox", sampling="bilinear", **IMAGENET)
        _assert_bits_equal(out.numpy()[0], want[0] if want.ndim == 4 else want, "frame loop")
    batch = pre.run(frames[:4], w, h, 24, 40)                     # grows one slot's buffers
    again = pre.run(frames[:4], w, h, 24, 40)                     # ... and the other's
    assert pre._staging.allocations == 8 and batch.shape == (4, 3, 24, 40)
    for k in range(4):
        want = O.preprocess(frames[k], w, h, 40, 24, fmt="nv12", mode="letterbox", sampling="bilinear", **IMAGENET)
        _assert_bits_equal(batch.numpy()[k], want[0] if want.ndim == 4 else want, f"batch frame {k}")
    for _ in range(3):
        third = pre.run(frames[:4], w, h, 24, 40)
    assert pre._staging.allocations == 8
    _assert_bits_equal(again.numpy(), batch.numpy(), "staging reuse")
    _assert_bits_equal(third.numpy(), batch.numpy(), "staging reuse, later turn")


def test_staging_ring_overlaps_and_stays_ordered(gpu_stream):
    """The ring under load: 24 back-to-back batches of DIFFERENT frames into two alternating outputs, no host sync in between — every
    kernel must have read the upload of its own call (a slot's device buffer is not overwritten before the kernel two calls back has
    finished; the pinned bytes are not overwritten before their DMA has).  Then the zero-copy form (opt-in): frames that already live
    in page-locked memory are DMA'd in place (no host copy, no pinned allocation).  Without the flag page-locked frames are staged
    like any others — the reference's contract: a frame may be rewritten as soon as the call returns."""
    from kornia_rs import Tensor
    from kornia_rs.hip import PinnedBuffer
    w, h, n = 64, 34, 6
    fb = w * h * 3 // 2
    pre = _pre(gpu_stream, mode="stretch", format="nv12", **IMAGENET)
    outs = [Tensor.uninit((n, 3, h, w), "float32", gpu_stream) for _ in range(24)]
    batches = [[_raw_for("nv12", w, h, seed=100 * r + k) for k in range(n)] for r in range(24)]
    for r in range(24):
        pre.run_host_batch(batches[r], w, h, outs[r])
    for r in (0, 1, 2, 11, 22, 23):
        got = outs[r].numpy_raw()
        for k in range(n):
            want = O.preprocess(batches[r][k], w, h, w, h, fmt="nv12", mode="stretch", **IMAGENET)[0]
            _assert_bits_equal(got[k], want, f"ring round {r} frame {k}")
    base_allocs, base_zc = pre._staging.allocations, pre._staging.zero_copy_uploads
    cap = PinnedBuffer(3 * n * fb)                                # three capture buffers of n frames each, page-locked
    view = cap.view()
    zc = [Tensor.uninit((n, 3, h, w), "float32", gpu_stream) for _ in range(6)]
    for r in range(6):
        b = r % 3
        if r >= 3:
            pre.wait_uploads()                                    # the capture side may rewrite a buffer once its DMA has left it
        for k in range(n):
            view[(b * n + k) * fb: (b * n + k + 1) * fb] = batches[r][k]
        pre.run_host_batch([view[(b * n + k) * fb: (b * n + k + 1) * fb] for k in range(n)], w, h, zc[r], zero_copy=True)
    assert pre._staging.zero_copy_uploads - base_zc == 6 and pre._staging.allocations == base_allocs
    for r in range(6):
        got = zc[r].numpy_raw()
        for k in range(n):
            want = O.preprocess(batches[r][k], w, h, w, h, fmt="nv12", mode="stretch", **IMAGENET)[0]
            _assert_bits_equal(got[k], want, f"zero-copy round {r} frame {k}")
    # default: page-locked frames are copied into the ring's own slot, so they may be scribbled on the moment the call returns
    pre.wait_uploads()
    base_zc = pre._staging.zero_copy_uploads
    safe = [Tensor.uninit((n, 3, h, w), "float32", gpu_stream) for _ in range(4)]
    for r in range(4):
        for k in range(n):
            view[k * fb: (k + 1) * fb] = batches[r][k]
        pre.run_host_batch([view[k * fb: (k + 1) * fb] for k in range(n)], w, h, safe[r])
        view[: n * fb] = 0xA5                                      # no wait_uploads(): the staged copy has already left these bytes
    assert pre._staging.zero_copy_uploads == base_zc
    for r in range(4):
        got = safe[r].numpy_raw()
        for k in range(n):
            want = O.preprocess(batches[r][k], w, h, w, h, fmt="nv12", mode="stretch", **IMAGENET)[0]
            _assert_bits_equal(got[k], want, f"staged page-locked round {r} frame {k}")


@pytest.mark.parametrize("fmt", ["nv12", "rgb", "bgra", "yuyv", "gray"])
@pytest.mark.parametrize("sampling", ["bilinear", "nearest"])
@pytest.mark.parametrize("quads", [-1, 0, 1])
@pytest.mark.parametrize("f16", [False, True])
def test_flattened_quad_kernel_equals_the_per_pixel_kernel_and_the_oracle(gpu_stream, fmt, sampling, quads, f16, dev_option):
    """(f16 = True, round 6: the same kernel with one 8-byte store of four binary16 values per plane.)
    f32 outputs of the one-tap samplers (nearest, on-grid bilinear) whose rows are whole quads go through preprocess_generic_quads (a
    lane owns one four-pixel quad of the flattened destination, 16-byte plane stores); the test option pre_quads = 0 keeps the
    per-pixel kernel everywhere, 1 takes the quad kernel for four-tap bilinear as well.  Geometries: the
    1080p letterboxes (on-grid 640, off-grid 608), a tail that does not fill the last block, quads that straddle the padding edge
    (pad 2.5 px), an upscale, a batch, and a ragged width that must fall back to the per-pixel kernel on its own."""
    from kornia_rs import Tensor
    from kornia_rs.hip import DeviceBuffer
    dev_option("pre_quads", quads)
    cases = [((1920, 1080), (640, 640), "letterbox"), ((1920, 1080), (608, 608), "letterbox"), ((130, 98), (96, 55), "stretch"),
             ((64, 48), (332, 171), "letterbox"), ((50, 90), (28, 40), "letterbox"), ((46, 34), (36, 27), "stretch"), ((258, 194), (224, 224), "stretch"),
             ((46, 34), (31, 27), "stretch"), ((8, 6), (4, 1), "stretch")]
    for (w, h), (dw, dh), mode in cases:
        raw = _raw_for(fmt, w, h, seed=7)
        kw = dict(fmt=fmt, mode=mode, sampling=sampling, f16=f16, **IMAGENET)
        _assert_bits_equal(_run(gpu_stream, raw, w, h, dw, dh, **kw), O.preprocess(raw, w, h, dw, dh, **kw), f"{fmt} {sampling} {w}x{h}->{dw}x{dh} {mode} quads={quads} f16={f16}")
    if fmt == "nv12":   # batched launch: frame k at its own source / destination stride; and the same frames as a list
        w, h, dw, dh, n = 64, 34, 40, 24, 5
        fb = w * h * 3 // 2
        frames = np.stack([_raw_for(fmt, w, h, seed=k) for k in range(n)])
        pre = _pre(gpu_stream, mode="letterbox", format="nv12", sampling=sampling, f16=f16, **IMAGENET)
        dst = Tensor.uninit((n, 3, dh, dw), "float16" if f16 else "float32", gpu_stream)
        pre.run_raw_batch(DeviceBuffer.from_numpy(frames.reshape(-1), gpu_stream), w, h, dst, frame_stride=fb)
        got = dst.numpy_raw()
        lst = Tensor.uninit((n, 3, dh, dw), "float16" if f16 else "float32", gpu_stream)
        pre.run_raw_batch([DeviceBuffer.from_numpy(frames[k], gpu_stream) for k in range(n)], w, h, lst)
        for k in range(n):
            want = O.preprocess(frames[k], w, h, dw, dh, fmt="nv12", mode="letterbox", sampling=sampling, f16=f16, **IMAGENET)[0]
            _assert_bits_equal(got[k].view(np.uint16) if f16 else got[k], want.view(np.uint16) if f16 else want, f"batch frame {k} quads={quads} f16={f16}")
            _assert_bits_equal(lst.numpy_raw()[k].view(np.uint16) if f16 else lst.numpy_raw()[k], want.view(np.uint16) if f16 else want, f"list frame {k} quads={quads} f16={f16}")
