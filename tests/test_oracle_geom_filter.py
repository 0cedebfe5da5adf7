"""Pin the geometry / filter oracle on the reference's known answers (no GPU).

  resize/mod.rs:447-490            resize_smoke_ch3 (18 values, linear ramp reproduced)
  warp/perspective.rs:377-436      invert_homography_known / round_trip / singular / small_scale,
                                   inverse_perspective_matrix
  warp/perspective.rs:497-590      warp_perspective_hflip, test_warp_perspective_resize (== resize)
  warp/affine.rs:471-640           edge sampling (flip), identity, rot90 via get_rotation_matrix2d
  filter/ops.rs:2185-2262          three exact 25-float gaussian_blur vectors
  filter/separable_filter.rs:270-306   impulse response / sum == 9
  cuda/remap.rs:770-804            remap_identity_*, remap_oob_writes_zero
"""
import numpy as np
import pytest

import oracle_ffi as O


def test_resize_smoke_ch3():
    img = np.arange(3 * 4 * 3, dtype=np.float32).reshape(4, 3, 3)
    got = O.resize(img, 2, 3).reshape(-1)
    want = [2.25, 3.25, 4.25, 6.75, 7.75, 8.75, 14.25, 15.25, 16.25, 18.75, 19.75, 20.75, 26.25, 27.25,
            28.25, 30.75, 31.75, 32.75]
    assert np.abs(got - np.array(want, np.float32)).max() < 1e-4
    assert np.array_equal(O.resize(img, 3, 4), img)  # same-size is an exact copy
    near = O.resize(img, 2, 3, "nearest")
    assert near.shape == (3, 2, 3)


def test_bicubic_reproduces_linear_ramp_and_constants():
    img = np.full((9, 7, 3), 0.25, np.float32)
    assert np.abs(O.resize(img, 15, 11, "bicubic") - 0.25).max() < 1e-6  # Keys weights sum to 1


def test_invert_homography():
    inv = O.invert_homography([1, 0, 2, 0, 1, 3, 0, 0, 1])
    assert np.abs(inv - np.array([1, 0, -2, 0, 1, -3, 0, 0, 1], np.float32)).max() < 1e-6
    h = np.array([1.02, 0.03, -5.0, -0.01, 0.99, 2.0, 0.00005, 0.00003, 1.0], np.float32)
    inv = O.invert_homography(h)
    assert np.abs(h.reshape(3, 3) @ inv.reshape(3, 3) - np.eye(3)).max() < 1e-5
    assert O.invert_homography([0] * 9) is None
    assert O.invert_homography([1, 2, 3, 2, 4, 6, 3, 6, 9]) is None
    assert O.invert_homography(np.array([1, 0, 2, 0, 1, 3, 0, 0, 1], np.float32) * np.float32(0.001)) is not None
    assert O.invert_homography([1, 0, -1, 0, 1, 1, 0, 0, 1]).tolist() == [1, 0, 1, 0, 1, -1, 0, 0, 1]


def test_warp_perspective_known():
    img = np.arange(6, dtype=np.float32).reshape(3, 2)
    got = O.warp_perspective(img, [-1, 0, 1, 0, 1, 0, 0, 0, 1], 2, 3)
    assert got.reshape(-1).tolist() == [1, 0, 3, 2, 5, 4]
    img = np.arange(16, dtype=np.float32).reshape(4, 4)
    got = O.warp_perspective(img, [0.5, 0, -0.25, 0, 0.5, -0.25, 0, 0, 1], 2, 2)
    assert got.reshape(-1).tolist() == [2.5, 4.5, 10.5, 12.5]
    assert np.array_equal(got, O.resize(img, 2, 2))
    assert O.warp_perspective(img, [1, 2, 3, 2, 4, 6, 3, 6, 9], 2, 2) is None


def rotation_matrix2d(center, angle, scale):  # warp/affine.rs:70-79
    f = np.float32
    a = f(f(angle) * f(np.pi) / f(180.0))
    alpha, beta = f(scale) * f(np.cos(a, dtype=np.float32)), f(scale) * f(np.sin(a, dtype=np.float32))
    tx = f(f(f(1.0) - alpha) * f(center[0])) - f(beta * f(center[1]))
    ty = f(beta * f(center[0])) + f(f(f(1.0) - alpha) * f(center[1]))
    return [alpha, beta, tx, -beta, alpha, ty]


def test_warp_affine_known():
    src = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.float32)
    got = O.warp_affine(src, [-1, 0, 3, 0, 1, 0], 4, 2, "nearest")
    assert got.reshape(-1).tolist() == [4, 3, 2, 1, 8, 7, 6, 5]
    img = np.arange(20, dtype=np.float32).reshape(5, 4)
    assert np.array_equal(O.warp_affine(img, [1, 0, 0, 0, 1, 0], 4, 5, "nearest")[:, :, 0], img)
    assert np.array_equal(O.warp_affine(img, [1, 0, 0, 0, 1, 0], 4, 5, "bilinear")[:, :, 0], img)
    rot = O.warp_affine(np.array([[0, 1], [2, 3]], np.float32), rotation_matrix2d((0.5, 0.5), 90.0, 1.0), 2, 2, "nearest")
    assert rot.reshape(-1).tolist() == [1, 3, 0, 2]
    inv = O.invert_affine([2, 0, 1, 0, 4, -2])
    assert inv.tolist() == [0.5, -0.0, -0.5, -0.0, 0.25, 0.5]


def test_remap_identity_and_oob():
    src = O.pattern_f32(13 * 9 * 3).reshape(9, 13, 3)
    xs, ys = np.meshgrid(np.arange(13, dtype=np.float32), np.arange(9, dtype=np.float32))
    for mode in ("bilinear", "nearest", "bicubic"):
        assert np.abs(O.remap(src, xs, ys, mode) - src).max() <= (0 if mode != "bicubic" else 1e-6)
    bad = xs.copy()
    bad[0, 0], bad[1, 1], bad[2, 2] = -0.5, 13.0, np.nan
    out = O.remap(src, bad, ys)
    assert out[0, 0].tolist() == [0, 0, 0] and out[1, 1].tolist() == [0, 0, 0] and out[2, 2].tolist() == [0, 0, 0]
    assert np.array_equal(out[3], src[3])


def test_gaussian_blur_exact_vectors():
    img = np.arange(25, dtype=np.float32).reshape(5, 5)
    want = np.array([0.57097936, 1.4260278, 2.3195207, 3.213014, 3.5739717, 4.5739717, 5.999999, 7.0, 7.999999,
                     7.9349294, 9.041435, 10.999999, 12.0, 12.999998, 12.402394, 13.5089, 15.999998, 17.0,
                     17.999996, 16.86986, 15.58594, 18.230816, 19.124311, 20.017801, 18.588936], np.float32)
    assert np.array_equal(O.gaussian_blur(img, (3, 3), (0.5, 0.5)).reshape(-1), want)
    want = np.array([0.573374, 1.4282724, 2.3214629, 3.2134287, 3.5740836, 4.5745554, 5.999999, 7.000791, 7.997888,
                     7.9328527, 9.039831, 10.997623, 11.999999, 12.996041, 12.399015, 13.500337, 15.989445,
                     16.992872, 17.987333, 16.858635, 15.576923, 18.21976, 19.117384, 20.004917, 18.577633], np.float32)
    assert np.array_equal(O.gaussian_blur(img, (0, 0), (0.5, 0.5)).reshape(-1), want)
    want = np.array([0.002010752, 1.001341, 2.001006, 3.0006707, 3.9986594, 4.998659, 6.0, 7.0000005, 8.0, 8.996648,
                     9.996984, 11.0, 12.000002, 13.0, 13.994974, 14.995307, 16.0, 17.0, 18.000002, 18.9933,
                     19.985254, 20.991283, 21.990952, 22.990616, 23.981903], np.float32)
    assert np.array_equal(O.gaussian_blur(img, (3, 3), (0.0, 0.0)).reshape(-1), want)
    assert O.gaussian_resolve((0, 0), (0.5, 0.5))[0] == (5, 5)
    assert O.gaussian_resolve((4, 3), (1.0, 1.0)) is None  # even kernel rejected (InvalidSigmaValue)


def test_separable_impulse():
    img = np.zeros((5, 5), np.float32)
    img[2, 2] = 1.0
    out = O.separable_filter(img, [1, 1, 1], [1, 1, 1])[:, :, 0]
    want = np.zeros((5, 5), np.float32)
    want[1:4, 1:4] = 1.0
    assert np.array_equal(out, want) and out.sum() == 9.0


def test_sobel_on_ramp():
    # horizontal ramp: gx = 8 in the interior (kx=[-1,0,1] x ky=[1,2,1]), gy = 0
    img = np.tile(np.arange(7, dtype=np.float32), (6, 1))
    out = O.gradient_magnitude(img, 0, 3)[:, :, 0]
    assert np.all(out[1:-1, 1:-1] == 8.0)
    assert O.gradient_kernels(0, 4) is None and O.gradient_kernels(1, 5) is None


def test_correction_map_identity_when_no_distortion():
    mx, my = O.correction_map((500.0, 500.0, 320.0, 240.0), (0,) * 8, 64, 48)
    xs, ys = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(48, dtype=np.float32))
    assert np.abs(mx - xs).max() < 1e-4 and np.abs(my - ys).max() < 1e-4


# ---- Lanczos-3 (P/interpolation/lanczos.rs) --------------------------------------------------------

def test_lanczos_four_eval_weights_match_per_tap_form():  # lanczos.rs:252-281
    for i in range(0, 10001, 7):
        frac = np.float32(i) / np.float32(10001.0)
        w = O.lanczos3_weights(frac)
        per_tap = [O.ko.ko_lanczos3(float(np.float32(frac + d))) for d in (2.0, 1.0, 0.0, -1.0, -2.0, -3.0)]
        assert np.abs(w - np.array(per_tap, np.float32)).max() < 1e-6, frac
    w0 = O.lanczos3_weights(0.0)
    assert w0[2] == 1.0 and w0[5] == 0.0


def test_sin_pi_tracks_libm_and_is_exact_at_integers():
    xs = np.linspace(-3.0, 3.0, 2001, dtype=np.float32)
    got = np.array([O.ko.ko_sin_pi(float(x)) for x in xs], np.float32)
    assert np.abs(got - np.sin(np.pi * xs.astype(np.float64))).max() < 5e-7
    assert all(O.ko.ko_sin_pi(float(k)) == 0.0 for k in range(-3, 4))


def test_lanczos_axis_tables_are_normalised_and_clamped():  # lanczos.rs:59-101
    for src_len, dst_len in [(129, 64), (63, 127), (5, 5), (1, 4)]:
        x0, w = O.lanczos_axis(src_len, dst_len)
        assert x0.min() >= 0 and x0.max() <= src_len - 1
        assert np.abs(w.sum(axis=1) - 1.0).max() < 1e-6
    x0, w = O.lanczos_axis(5, 5)  # identity grid: frac == 0 -> delta weights
    assert x0.tolist() == [0, 1, 2, 3, 4] and np.array_equal(w, np.tile(np.array([0, 0, 1, 0, 0, 0], np.float32), (5, 1)))


def test_lanczos_resize_and_warps_reproduce_constants_and_identity():
    img = np.full((9, 13, 3), 0.25, np.float32)
    assert np.abs(O.resize(img, 15, 11, "lanczos") - 0.25).max() < 1e-6
    src = O.pattern_f32(33 * 21 * 3).reshape(21, 33, 3)
    assert np.array_equal(O.resize(src, 33, 21, "lanczos"), src)  # same-size short circuit, resize/mod.rs:134
    ident = O.warp_perspective(src, [1, 0, 0, 0, 1, 0, 0, 0, 1], 33, 21, "lanczos")
    assert np.abs(ident - src).max() < 1e-6
    # resize == the two-pass definition written independently in numpy (f64 accumulate, loose tolerance)
    x0, wx = O.lanczos_axis(33, 20)
    y0, wy = O.lanczos_axis(21, 30)
    xi = np.clip(x0[:, None] + np.arange(6) - 2, 0, 32)
    yi = np.clip(y0[:, None] + np.arange(6) - 2, 0, 20)
    inter = np.einsum("yxtc,xt->yxc", src[:, xi, :].astype(np.float64), wx)
    want = np.einsum("ytxc,yt->yxc", inter[yi, :, :], wy)
    assert np.abs(O.resize(src, 20, 30, "lanczos") - want).max() < 1e-5


# ---- PixelMapping + fused resize / normalise launchers (P/cuda/resize.rs) -----------------------------------------------

REF_COEFFS = [("half_pixel", 640, 320, (2.0, 0.5)), ("half_pixel", 320, 640, (0.5, -0.25)), ("half_pixel", 9, 1, (9.0, 4.0)),
              ("align_corners", 641, 321, (2.0, 0.0)), ("align_corners", 9, 1, (0.0, 0.0))]


def test_pixel_mapping_coeffs_reference_vector():  # cuda/resize.rs:931-940, restatement AND the product's host function
    import ctypes as C
    from kornia_rs import _ffi
    for mapping, s, d, want in REF_COEFFS:
        assert O.pixel_mapping_coeffs(mapping, s, d) == want
        out = (C.c_float * 2)()
        assert _ffi.lib.kh_pixel_mapping_coeffs(_ffi.KH_PIXEL_MAPPING[mapping], s, d, out) == 0
        assert (float(out[0]), float(out[1])) == want
    assert _ffi.lib.kh_pixel_mapping_coeffs(2, 8, 8, (C.c_float * 2)()) == _ffi.KH_ERR_INVALID_ARG
    assert _ffi.lib.kh_pixel_mapping_coeffs(0, 0, 8, (C.c_float * 2)()) == _ffi.KH_ERR_INVALID_ARG


def test_mapped_resize_half_pixel_is_resize_and_align_corners_hits_corners():
    src = O.pattern_f32(23 * 17 * 3).reshape(17, 23, 3)
    for mode in ("nearest", "bilinear", "bicubic", "lanczos"):
        assert np.array_equal(O.resize_mapped(src, 11, 9, mode, "half_pixel"), O.resize(src, 11, 9, mode)), mode
        ac = O.resize_mapped(src, 12, 9, mode, "align_corners")  # (23-1)/(12-1) = 2, (17-1)/(9-1) = 2: every sample on a pixel
        assert np.array_equal(ac, src[::2, ::2]), mode
        assert np.array_equal(O.resize_mapped(src, 23, 17, mode, "align_corners"), src), mode
    one = O.resize_mapped(src, 1, 1, "bilinear", "align_corners")  # a 1-wide destination axis pins to source 0
    assert np.array_equal(one[0, 0], src[0, 0])


def test_fused_resize_normalize_is_resize_then_the_kernel_epilogue():
    src = O.pattern_f32(37 * 29 * 3).reshape(29, 37, 3)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    for mapping in ("half_pixel", "align_corners"):
        got = O.resize_bilinear_normalize(src, 16, 12, mean, std, mapping)
        inv = (np.float32(1.0) / np.asarray(std, np.float32)).astype(np.float32)
        want = ((O.resize_mapped(src, 16, 12, "bilinear", mapping) - np.asarray(mean, np.float32)) * inv).astype(np.float32)
        assert np.array_equal(got, want), mapping
    ident = O.resize_bilinear_normalize(src, 16, 12, (0, 0, 0), (1, 1, 1))
    assert np.array_equal(ident, O.resize(src, 16, 12, "bilinear"))
    with pytest.raises(ValueError):
        O.resize_bilinear_normalize(src, 16, 12, mean, (0.2, 0.0, 0.2))  # "std must be non-zero" (resize.rs:606-610)
