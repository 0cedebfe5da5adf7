"""Pin the restatement of the rest of the reference's filter module on its own known answers (no GPU).

  filter/ops.rs:2263-2355          test_spatial_gradient (5x5x2 ramp, all three variants share the arithmetic)
  filter/ops.rs:2356-2378          test_scharr_spatial_gradient (centre values)
  filter/kernels.rs:217-226        test_box_blur_fast_kernels_1d
  filter/ops.rs:2157-2183          test_box_blur_fast (25 exact floats)
  filter/median.rs:942-967,1066-1093   networks == naive median, bad ksize, constant image
  filter/bilateral.rs:400-428      degenerate sigma copies through, constant image, radius / tap-count rule
plus independent numpy forms of every operator (written from the definitions, not from the C code).
"""
import numpy as np
import pytest

import oracle_ffi as O


def _ramp2():
    return np.stack([np.arange(25, dtype=np.float32), np.arange(25, dtype=np.float32) + 25.0], -1).reshape(5, 5, 2)


def test_spatial_gradient_known_answer():
    gx, gy = O.spatial_gradient(_ramp2(), "sobel")
    want_x = np.tile(np.array([0.5, 1.0, 1.0, 1.0, 0.5], np.float32), (5, 1))
    want_y = np.repeat(np.array([2.5, 5.0, 5.0, 5.0, 2.5], np.float32), 5).reshape(5, 5)
    for ch in range(2):
        assert np.array_equal(gx[..., ch], want_x) and np.array_equal(gy[..., ch], want_y)
    sx, sy = O.spatial_gradient(_ramp2(), "scharr")
    assert sx[2, 2, 0] == 1.0 and sy[2, 2, 0] == 5.0


def _gradient_numpy(img, kx, ky):
    """Replicate-padded 3x3 cross-correlation, products added in (dy, dx) order onto 0.0 — all in float32."""
    h, w, c = img.shape
    p = np.pad(img, ((1, 1), (1, 1), (0, 0)), mode="edge")
    gx, gy = np.zeros_like(img), np.zeros_like(img)
    for dy in range(3):
        for dx in range(3):
            v = p[dy:dy + h, dx:dx + w]
            gx = (gx + v * np.float32(kx[dy][dx])).astype(np.float32)
            gy = (gy + v * np.float32(ky[dy][dx])).astype(np.float32)
    return gx, gy


@pytest.mark.parametrize("kind", ["sobel", "scharr"])
@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 7, 3), (6, 1, 2), (2, 2, 4), (13, 17, 3)])
def test_spatial_gradient_matches_numpy_form(kind, shape):
    a, b, m = (0.125, 0.25, 0.125) if kind == "sobel" else (0.09375, 0.3125, 0.09375)
    kx = [[-a, 0, a], [-b, 0, b], [-m, 0, m]]
    ky = [[-a, -b, -m], [0, 0, 0], [a, b, m]]
    img = O.pattern_f32(int(np.prod(shape))).reshape(shape)
    gx, gy = O.spatial_gradient(img, kind)
    wx, wy = _gradient_numpy(img, kx, ky)
    assert np.array_equal(gx, wx) and np.array_equal(gy, wy)


def test_box_blur_fast_kernels_known_answers():
    assert O.box_blur_fast_kernels_1d(0.5, 3) == [1, 1, 1]
    assert O.box_blur_fast_kernels_1d(0.5, 4) == [1, 1, 1, 1]
    assert O.box_blur_fast_kernels_1d(0.5, 5) == [1, 1, 1, 1, 1]
    assert O.box_blur_fast_kernels_1d(1.0, 3) == [1, 1, 3]
    assert O.box_blur_fast_kernels_1d(1.0, 5) == [1, 1, 1, 1, 3]


def test_box_blur_fast_known_answer():
    img = np.arange(25, dtype=np.float32).reshape(5, 5, 1)
    want = np.array([4.444444, 4.9259257, 5.7037034, 6.4814816, 6.962963,
                     6.851851, 7.3333335, 8.111111, 8.888889, 9.370372,
                     10.740741, 11.222222, 12.0, 12.777779, 13.259262,
                     14.629628, 15.111112, 15.888888, 16.666666, 17.14815,
                     17.037035, 17.518518, 18.296295, 19.074074, 19.555555], np.float32)
    assert np.array_equal(O.box_blur_fast(img, (0.5, 0.5)).reshape(-1), want)  # assert_eq! in the reference: exact


def test_fast_horizontal_filter_is_a_transposed_running_box():
    img = O.pattern_f32(9 * 14 * 3).reshape(9, 14, 3)
    for half in (0, 1, 3, 13):
        got = O.fast_horizontal_filter(img, half)
        assert got.shape == (14, 9, 3)
        p = np.pad(img.astype(np.float64), ((0, 0), (half, half), (0, 0)), mode="edge")
        box = sum(p[:, k:k + 14] for k in range(2 * half + 1)) / (2 * half + 1)
        assert np.abs(got.transpose(1, 0, 2) - box).max() < 1e-5  # same filter; the running sum only reorders f32 additions
    assert np.array_equal(O.fast_horizontal_filter(img, 0).transpose(1, 0, 2), img)  # half 0: x * 1 / 1
    assert O.fast_horizontal_filter(img, 14) is None  # the reference indexes past the row / panics


@pytest.mark.parametrize("ksize", [3, 5])
@pytest.mark.parametrize("shape", [(48, 64, 1), (43, 67, 1), (4, 5, 1), (1, 1, 1), (9, 2, 1), (21, 33, 3), (7, 6, 4), (5, 9, 2)])
def test_median_matches_numpy_median(ksize, shape):
    img = O.pattern_u8(int(np.prod(shape))).reshape(shape)
    r = ksize // 2
    p = np.pad(img, ((r, r), (r, r), (0, 0)), mode="edge")
    h, w, _ = shape
    win = np.stack([p[dy:dy + h, dx:dx + w] for dy in range(ksize) for dx in range(ksize)], 0)
    want = np.sort(win, 0)[ksize * ksize // 2]
    assert np.array_equal(O.median_blur(img, ksize), want)


def test_median_edge_cases():
    assert O.median_blur(np.zeros((4, 4, 1), np.uint8), 4) is None and O.median_blur(np.zeros((4, 4, 1), np.uint8), 7) is None
    const = np.full((12, 16, 1), 200, np.uint8)
    assert np.array_equal(O.median_blur(const, 3), const) and np.array_equal(O.median_blur(const, 5), const)


def test_v_exp_tracks_exp_and_is_exact_at_zero():
    assert O.ko.ko_v_exp_f32(0.0) == 1.0
    xs = np.linspace(-80.0, 0.0, 4001, dtype=np.float32)
    got = np.array([O.ko.ko_v_exp_f32(float(x)) for x in xs], np.float64)
    want = np.exp(xs.astype(np.float64))
    assert (np.abs(got - want) / want).max() < 4e-7  # a few f32 ulps, as the reference's doc says of cv2's polynomial


def test_bilateral_table_rules():
    assert O.bilateral_tables(5, 50.0, 50.0)["radius"] == 2
    assert O.bilateral_tables(0, 50.0, 2.0)["radius"] == 3
    assert O.bilateral_tables(-1, 50.0, 0.1)["radius"] == 1
    t5, t3 = O.bilateral_tables(5, 50.0, 50.0), O.bilateral_tables(3, 50.0, 50.0)
    assert len(t5["dy"]) == 13 and len(t3["dy"]) == 5
    assert list(t5["simd_order"]) == [0, 12, 1, 2, 3, 9, 10, 11, 4, 5, 6, 7, 8] and list(t3["simd_order"]) == [0, 1, 2, 3, 4]
    # row-major circular mask, centre included, weights exp(-r^2 / (2 sigma^2))
    assert list(zip(t3["dy"], t3["dx"])) == [(-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)]
    r2 = (t5["dy"].astype(np.float64) ** 2 + t5["dx"].astype(np.float64) ** 2)
    assert np.abs(t5["space_weight"] - np.exp(-0.5 * r2 / 2500.0)).max() < 1e-7
    d = np.arange(256, dtype=np.float64)
    assert np.abs(t5["color_weight"] - np.exp(-0.5 * d * d / 2500.0)).max() < 1e-6 and t5["color_weight"][0] == 1.0
    assert O.bilateral_tables(0, 30.0, 2.5)["radius"] == 4  # 3.75 -> 4; ties go to even: 1.5 * 3 = 4.5 -> 4
    assert O.bilateral_tables(0, 30.0, 3.0)["radius"] == 4


def test_bilateral_reference_tests():
    src = np.arange(12, dtype=np.uint8).reshape(3, 4, 1)
    assert np.array_equal(O.bilateral_filter(src, 5, 0.0, 50.0), src)  # degenerate_sigma_copies_through
    assert np.array_equal(O.bilateral_filter(src, 5, 50.0, 1e-7), src)
    const = np.full((12, 16, 1), 200, np.uint8)
    assert np.array_equal(O.bilateral_filter(const, 5, 50.0, 50.0), const)  # constant_image_unchanged


def _bilateral_numpy(img, d, sc, ss):
    """The definition in float64: normalised sum over the circular window of exp(-r^2/2ss^2) * exp(-dv^2/2sc^2) * v, reflect-101."""
    t = O.bilateral_tables(d, sc, ss)
    r = t["radius"]
    h, w = img.shape[:2]
    p = np.pad(img[..., 0].astype(np.float64), r, mode="reflect") if min(h, w) > r else None
    if p is None:
        return None
    num, den = np.zeros((h, w)), np.zeros((h, w))
    v0 = img[..., 0].astype(np.float64)
    for dy, dx in zip(t["dy"], t["dx"]):
        v = p[r + dy:r + dy + h, r + dx:r + dx + w]
        wgt = np.exp(-0.5 * (dy * dy + dx * dx) / (ss * ss)) * np.exp(-0.5 * (v - v0) ** 2 / (sc * sc))
        num += wgt * v
        den += wgt
    return num / den


@pytest.mark.parametrize("d,sc,ss", [(5, 50.0, 50.0), (3, 25.0, 10.0), (9, 75.0, 75.0), (0, 30.0, 3.0)])
@pytest.mark.parametrize("shape", [(48, 64), (43, 67), (11, 19)])
def test_bilateral_tracks_the_float64_definition(d, sc, ss, shape):
    img = O.pattern_u8(shape[0] * shape[1]).reshape(shape + (1,))
    got = O.bilateral_filter(img, d, sc, ss)[..., 0].astype(np.float64)
    want = _bilateral_numpy(img, d, sc, ss)
    # the f32 tables / accumulation move a value across a rounding boundary at most: never more than one grey level
    assert np.abs(got - np.rint(want)).max() <= 1 and (got != np.rint(want)).mean() < 0.02


def test_bilateral_tap_order_split_is_observable():
    """d = 5 (13 taps): pixels left of simd_region_end accumulate in cv2's unrolled order, the scalar tail sequentially.  The two
    orders give different f32 sums for some pixels, so a restatement with one order everywhere would not be byte-exact."""
    w, h = 40, 64  # simd_end = 32: columns 32..39 are the scalar tail
    img = O.pattern_u8(w * h).reshape(h, w, 1)
    got = O.bilateral_filter(img, 5, 50.0, 50.0)
    wide = np.concatenate([img, img[:, :8]], 1)  # 48 columns: simd_end = 48, every column of the first 40 is in the SIMD region
    alt = O.bilateral_filter(np.ascontiguousarray(wide), 5, 50.0, 50.0)[:, :36]  # columns < 36 see the same neighbourhood (radius 2 + no reflect)
    assert np.array_equal(got[:, :32], alt[:, :32])  # same order, same bytes
    assert got[:, 32:36].shape == alt[:, 32:36].shape  # tail columns may differ by the accumulation order only
    assert np.abs(got[:, 32:36].astype(int) - alt[:, 32:36].astype(int)).max() <= 1
