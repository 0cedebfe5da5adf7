"""GPU parity for the fused LDS separable filter (gaussian / box / sobel / scharr / generic).
Bit-exact against the CPU oracle, which is itself pinned on the reference's exact 25-float vectors
(P/filter/ops.rs:2185-2262); shapes follow f32_filters_device_equal_host_bitexact
(P/filter/cuda.rs:254: 67x43)."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits, dev, fptr, out_buf

pytestmark = pytest.mark.gpu


def img(w, h, c, seed=0):
    return np.roll(O.pattern_f32(w * h * c + seed), -seed)[: w * h * c].reshape(h, w, c).copy()


def run(gpu_stream, name, src, *args, batch=1):
    from kornia_rs import _ffi
    h, w, c = src.shape[-3:]
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, src.nbytes)
    _ffi.check(getattr(_ffi.lib, name)(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, c, *args, batch,
                                       h * w * c, h * w * c))
    return d_dst.to_numpy(np.float32, src.shape)


def test_gaussian_exact_reference_vectors(gpu_stream):
    src = np.arange(25, dtype=np.float32).reshape(5, 5, 1)
    want = np.array([0.57097936, 1.4260278, 2.3195207, 3.213014, 3.5739717, 4.5739717, 5.999999, 7.0, 7.999999,
                     7.9349294, 9.041435, 10.999999, 12.0, 12.999998, 12.402394, 13.5089, 15.999998, 17.0,
                     17.999996, 16.86986, 15.58594, 18.230816, 19.124311, 20.017801, 18.588936], np.float32)
    assert np.array_equal(run(gpu_stream, "kh_gaussian_blur_f32", src, 3, 3, 0.5, 0.5).reshape(-1), want)
    want = np.array([0.573374, 1.4282724, 2.3214629, 3.2134287, 3.5740836, 4.5745554, 5.999999, 7.000791, 7.997888,
                     7.9328527, 9.039831, 10.997623, 11.999999, 12.996041, 12.399015, 13.500337, 15.989445,
                     16.992872, 17.987333, 16.858635, 15.576923, 18.21976, 19.117384, 20.004917, 18.577633], np.float32)
    assert np.array_equal(run(gpu_stream, "kh_gaussian_blur_f32", src, 0, 0, 0.5, 0.5).reshape(-1), want)
    want = np.array([0.002010752, 1.001341, 2.001006, 3.0006707, 3.9986594, 4.998659, 6.0, 7.0000005, 8.0, 8.996648,
                     9.996984, 11.0, 12.000002, 13.0, 13.994974, 14.995307, 16.0, 17.0, 18.000002, 18.9933,
                     19.985254, 20.991283, 21.990952, 22.990616, 23.981903], np.float32)
    assert np.array_equal(run(gpu_stream, "kh_gaussian_blur_f32", src, 3, 3, 0.0, 0.0).reshape(-1), want)


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("shape", [(67, 43), (300, 70), (1, 1), (5, 200), (257, 9)])
@pytest.mark.parametrize("ks", [((3, 3), (0.8, 0.8)), ((7, 7), (1.5, 1.5)), ((5, 9), (1.0, 2.0)), ((13, 3), (2.5, 0.0)), ((1, 7), (0.0, 1.5)), ((9, 1), (2.0, 0.0)),
                                ((3, 15), (0.8, 3.0))])
def test_gaussian_matches_oracle(gpu_stream, c, shape, ks):
    (w, h), ((kx, ky), (sx, sy)) = shape, ks
    src = img(w, h, c)
    got = run(gpu_stream, "kh_gaussian_blur_f32", src, kx, ky, sx, sy)
    assert_same_bits(got, O.gaussian_blur(src, (kx, ky), (sx, sy)), f"gaussian {shape} c{c} k{kx}x{ky}")


@pytest.mark.parametrize("ks", [((3, 7), (0.8, 1.5)), ((9, 5), (2.0, 1.0)), ((7, 7), (1.5, 1.5)), ((5, 5), (1.0, 1.0)), ((1, 7), (0.0, 1.5)), ((15, 1), (3.0, 0.0)), ((3, 15), (0.8, 3.0))])
def test_non_finite_pixels_poison_only_their_own_window(gpu_stream, ks):
    """An Inf / NaN pixel reaches exactly the outputs whose n-tap windows contain it — also when kx and ky differ in length
    (a kernel padded with zero taps would compute 0 * Inf = NaN outside the reference's window)."""
    img = O.pattern_f32(61 * 47 * 3).reshape(47, 61, 3).copy()  # noqa: F811 (local image)
    img[20, 30, 1] = np.inf
    img[5, 7, 0] = np.nan
    img[40, 50, 2] = -np.inf
    (kx, ky), (sx, sy) = ks
    got, want = run(gpu_stream, "kh_gaussian_blur_f32", img, kx, ky, sx, sy), O.gaussian_blur(img, (kx, ky), (sx, sy))
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))
    assert_same_bits(np.nan_to_num(got, nan=7.0, posinf=8.0, neginf=9.0), np.nan_to_num(want, nan=7.0, posinf=8.0, neginf=9.0), f"non-finite {ks}")


@pytest.fixture(params=["roll", "roll4", "tile"])
def kernel_path(request, dev_option):
    """The device kernels behind the filter entry points: the rolling-column fast path (one column per lane), its four-columns-per-lane
    form (rows of a multiple of four floats >= 1024, kernels up to 9 taps: the default where it applies) and the LDS-tile kernel —
    each forced through the library's test options so that every geometry below reaches it."""
    dev_option("filter_four_columns", 1 if request.param == "roll4" else 0)
    if request.param == "tile":
        dev_option("filter_force_tile", 1)
    return request.param


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("ks", [((3, 3), (0.8, 0.8)), ((7, 7), (1.5, 1.5)), ((5, 9), (1.0, 2.0)), ((15, 15), (3.0, 3.0)),
                                ((13, 3), (2.5, 0.0)), ((17, 17), (3.0, 3.0)), ((31, 5), (6.0, 1.0))])
def test_both_kernels_match_oracle(gpu_stream, kernel_path, c, ks):
    (kx, ky), (sx, sy) = ks
    for (w, h) in [(67, 43), (300, 200), (64, 91), (1030, 37), (1028, 37), (344, 50), (2052, 9)]:
        src = img(w, h, c, seed=3)
        got = run(gpu_stream, "kh_gaussian_blur_f32", src, kx, ky, sx, sy)
        assert_same_bits(got, O.gaussian_blur(src, (kx, ky), (sx, sy)), f"{kernel_path} {w}x{h} c{c} k{kx}x{ky}")
    src = img(131, 120, c)
    for kind, n in [(0, 3), (0, 5), (1, 3)]:
        got = run(gpu_stream, "kh_gradient_magnitude_f32", src, kind, n)
        assert_same_bits(got, O.gradient_magnitude(src, kind, n), f"{kernel_path} grad {kind} {n} c{c}")


def test_gaussian_4k_tile_seams_and_batch(gpu_stream):
    """config[3] geometry: 3840x2160x3, 7x7 sigma 1.5 — one full-size image checked against the
    oracle everywhere (covers every tile seam), plus a 2-image batched launch."""
    w, h = 3840, 2160
    src = img(w, h, 3)
    got = run(gpu_stream, "kh_gaussian_blur_f32", src, 7, 7, 1.5, 1.5)
    assert_same_bits(got, O.gaussian_blur(src, (7, 7), (1.5, 1.5)), "4K gaussian")
    small = np.stack([img(300, 130, 3, seed=k) for k in range(2)])
    got = run(gpu_stream, "kh_gaussian_blur_f32", small, 7, 7, 1.5, 1.5, batch=2)
    for k in range(2):
        assert_same_bits(got[k], O.gaussian_blur(small[k], (7, 7), (1.5, 1.5)), f"batch {k}")


@pytest.mark.parametrize("ks", [(3, 3), (5, 5), (3, 7), (1, 1)])
def test_box_blur(gpu_stream, ks):
    src = img(67, 43, 3)
    got = run(gpu_stream, "kh_box_blur_f32", src, ks[0], ks[1])
    assert_same_bits(got, O.separable_filter(src, O.box_kernel_1d(ks[0]), O.box_kernel_1d(ks[1])), f"box {ks}")


def test_separable_impulse_and_generic(gpu_stream):  # filter/separable_filter.rs:270-306
    from kornia_rs import _ffi
    src = np.zeros((5, 5, 1), np.float32)
    src[2, 2] = 1.0
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, src.nbytes)
    k = fptr([1, 1, 1])
    _ffi.check(_ffi.lib.kh_separable_filter_f32(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, 5, 5, 1, k, 3, k, 3, 1, 0, 0))
    out = d_dst.to_numpy(np.float32, (5, 5))
    want = np.zeros((5, 5), np.float32)
    want[1:4, 1:4] = 1.0
    assert np.array_equal(out, want) and out.sum() == 9.0
    src = img(131, 77, 3)
    kx, ky = np.array([0.1, -0.3, 0.7, 0.2, 0.05], np.float32), np.array([-1.0, 2.0, 0.5], np.float32)
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, src.nbytes)
    _ffi.check(_ffi.lib.kh_separable_filter_f32(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, 131, 77, 3, fptr(kx), 5,
                                                fptr(ky), 3, 1, 0, 0))
    assert_same_bits(d_dst.to_numpy(np.float32, src.shape), O.separable_filter(src, kx, ky), "generic taps")


@pytest.mark.parametrize("kind,n", [(0, 3), (0, 5), (1, 3)])
@pytest.mark.parametrize("c", [1, 3])
def test_sobel_scharr(gpu_stream, kind, n, c):
    for (w, h) in [(67, 43), (300, 70)]:
        src = img(w, h, c)
        got = run(gpu_stream, "kh_gradient_magnitude_f32", src, kind, n)
        assert_same_bits(got, O.gradient_magnitude(src, kind, n), f"grad kind{kind} k{n} {w}x{h} c{c}")


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("kind,n", [(0, 3), (0, 5), (1, 3)])
def test_gradient_magnitude_wide_rows(gpu_stream, kernel_path, c, kind, n):
    """sobel / scharr on rows of >= 1024 floats: the four-columns-per-lane rolling kernel carries the gradient pair since round 6 (a
    second register ring; 16-byte loads and stores) where the rows are whole float4s on aligned images; the one-column kernel and the
    LDS-tile kernel (forced) and rows that are not whole float4s give the same bits; a batch."""
    for (w, h) in [(1028, 37), (1030, 21), (344, 50), (2052, 9), (3840, 12)]:
        src = img(w, h, c, seed=5)
        got = run(gpu_stream, "kh_gradient_magnitude_f32", src, kind, n)
        assert_same_bits(got, O.gradient_magnitude(src, kind, n), f"{kernel_path} grad kind{kind} k{n} {w}x{h} c{c}")
    both = np.stack([img(1028, 23, c, seed=k) for k in range(2)])
    got = run(gpu_stream, "kh_gradient_magnitude_f32", both, kind, n, batch=2)
    for k in range(2):
        assert_same_bits(got[k], O.gradient_magnitude(both[k], kind, n), f"{kernel_path} batch image {k}")


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("k", [17, 19, 21, 23, 25, 27, 29, 31])
def test_wide_kernels_rolling_path(gpu_stream, dev_option, c, k):
    """Gaussians of 17..31 taps (sigma 3-5) take the wide rolling kernel since round 6 (the LDS-tile kernel before: 12-65x slower per 4K
    image): the oracle's bits on images shorter than the kernel, narrower than a wave, with partial tiles and several strips; the tile
    kernel (forced) the same; a batch; box blur of the same size."""
    sig = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    for (w, h) in [(67, 43), (300, 130), (20, 9), (1030, 37), (64, 400)]:
        src = img(w, h, c, seed=7)
        want = O.gaussian_blur(src, (k, k), (sig, sig))
        for opt in (-1, 1):
            dev_option("filter_force_tile", opt)
            assert_same_bits(run(gpu_stream, "kh_gaussian_blur_f32", src, k, k, sig, sig), want, f"gaussian {k}x{k} {w}x{h} c{c} force_tile={opt}")
    dev_option("filter_force_tile", -1)
    both = np.stack([img(131, 77, c, seed=s_) for s_ in range(2)])
    got = run(gpu_stream, "kh_gaussian_blur_f32", both, k, k, sig, sig, batch=2)
    for i in range(2):
        assert_same_bits(got[i], O.gaussian_blur(both[i], (k, k), (sig, sig)), f"batch image {i} {k}x{k} c{c}")
    src = img(150, 60, c)
    assert_same_bits(run(gpu_stream, "kh_box_blur_f32", src, k, k), O.separable_filter(src, O.box_kernel_1d(k), O.box_kernel_1d(k)), f"box {k}x{k} c{c}")


@pytest.mark.parametrize("policy", [-1, 0, 1])
def test_row_store_policy_changes_no_bit(gpu_stream, dev_option, policy):
    """Rows that are not whole 128-byte lines take write-back stores in the rolling kernels, line-aligned rows the streaming policy
    (round 6, kh_common.h::plain_row_stores); test option row_stores forces either: the oracle's bits every way, on row lengths
    either side of the rule, for the one-column, four-column and wide kernels, a gradient and a batch."""
    dev_option("row_stores", policy)
    for (w, h, c) in [(1024, 12, 1), (1023, 9, 1), (1028, 7, 1), (352, 11, 3), (351, 8, 3), (344, 9, 3), (256, 6, 4), (257, 5, 4)]:
        src = img(w, h, c, seed=w + c)
        for k, sig in ((5, 1.0), (11, 2.0), (17, 3.0)):
            assert_same_bits(run(gpu_stream, "kh_gaussian_blur_f32", src, k, k, sig, sig), O.gaussian_blur(src, (k, k), (sig, sig)), f"gaussian {w}x{h} c{c} k{k} row_stores={policy}")
        assert_same_bits(run(gpu_stream, "kh_gradient_magnitude_f32", src, 0, 3), O.gradient_magnitude(src, 0, 3), f"sobel {w}x{h} c{c} row_stores={policy}")
    batch = np.stack([img(1031, 9, 1, seed=k) for k in range(3)])
    got = run(gpu_stream, "kh_gaussian_blur_f32", batch, 5, 5, 1.0, 1.0, batch=3)
    for k in range(3):
        assert_same_bits(got[k], O.gaussian_blur(batch[k], (5, 5), (1.0, 1.0)), f"batch frame {k} row_stores={policy}")


def test_filter_validation(gpu_stream):
    from kornia_rs import _ffi
    lib, s = _ffi.lib, gpu_stream.cuda_stream_ptr
    a, b = C.c_void_p(256), C.c_void_p(512)
    assert lib.kh_gaussian_blur_f32(s, a, b, 8, 8, 3, 4, 3, 1.0, 1.0, 1, 0, 0) == _ffi.KH_ERR_INVALID_ARG  # even kernel
    assert lib.kh_gaussian_blur_f32(s, a, b, 8, 8, 3, 0, 0, 0.0, 0.0, 1, 0, 0) == _ffi.KH_ERR_INVALID_ARG
    assert lib.kh_gradient_magnitude_f32(s, a, b, 8, 8, 3, 0, 4, 1, 0, 0) == _ffi.KH_ERR_INVALID_ARG
    assert lib.kh_gradient_magnitude_f32(s, a, b, 8, 8, 3, 1, 5, 1, 0, 0) == _ffi.KH_ERR_INVALID_ARG
    assert lib.kh_box_blur_f32(s, a, b, 8, 8, 3, 65, 3, 1, 0, 0) == _ffi.KH_ERR_UNSUPPORTED
    assert lib.kh_gaussian_blur_f32(s, a, a, 8, 8, 3, 3, 3, 1.0, 1.0, 1, 0, 0) == _ffi.KH_ERR_INVALID_ARG  # in place
    t = (C.c_float * 7)()
    assert lib.kh_gaussian_kernel_1d(7, 1.5, t) == 0
    assert np.array_equal(np.array(list(t), np.float32), O.gaussian_kernel_1d(7, 1.5))
