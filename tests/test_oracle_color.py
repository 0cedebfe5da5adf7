"""Pin the colour-family oracle on the reference's own known-answer tests (no GPU).

  color/gray/mod.rs:271-300,417-441   f32 gray regressions
  color/yuv/kernels.rs ycc_u8_known_value_gray, ycc_u8_round_trip_close, ycc_f32_round_trip
  color/hsv/mod.rs:170-192            hsv_from_rgb_regression (18 values)
  color/hls/mod.rs:193-222            hls_from_rgb_regression
  color/sepia.rs:184-241              sepia_u8_known_value / sepia_f32_known_value
  color/rgb/mod.rs tests              rgb_from_rgba drop-alpha (verified with opencv)
"""
import numpy as np

import oracle_ffi as O


def test_gray_f32_regressions():
    src = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 1, .5, .5, .5], np.float32)
    got = O.color_map("gray_from_rgb_f32", src, 1)
    assert np.abs(got - np.array([0.299, 0.587, 0.114, 0.0, 1.0, 0.5], np.float32)).max() < 1e-6


def test_ycc_known_gray_and_round_trips():
    assert O.color_map("ycc_from_rgb_u8", np.array([128, 128, 128], np.uint8), 3, 0).tolist() == [128, 128, 128]
    rgb = np.arange(64 * 3, dtype=np.uint32).astype(np.uint8)  # ramp_u8
    back = O.color_map("rgb_from_ycc_u8", O.color_map("ycc_from_rgb_u8", rgb, 3, 0), 3, 0)
    assert np.abs(back.astype(int) - rgb.astype(int)).max() <= 3
    f = np.array([(v % 17) / 16.0 for v in range(33)], np.float32)
    back = O.color_map("rgb_from_ycc_f32", O.color_map("ycc_from_rgb_f32", f, 3, 1), 3, 1)
    assert np.abs(back - f).max() <= 5e-4
    back = O.color_map("rgb_from_ycc_f32", O.color_map("ycc_from_rgb_f32", f, 3, 0), 3, 0)
    assert np.abs(back - f).max() <= 1e-5


def test_hsv_regression():
    src = np.array([0, 128, 255, 255, 128, 0, 128, 255, 0, 255, 0, 128, 0, 128, 255, 255, 128, 0], np.float32)
    want = np.array([148.66667, 255, 255, 21.333334, 255, 255, 63.666668, 255, 255, 233.66667, 255, 255,
                     148.66667, 255, 255, 21.333334, 255, 255], np.float32)
    assert np.abs(O.color_map("hsv_from_rgb_f32", src, 3) - want).max() < 1e-3


def test_hls_regression():
    src = np.array([255, 0, 0, 0, 255, 0, 0, 0, 255], np.float32)
    want = np.array([0, 127.5, 255, 85.0, 127.5, 255, 170.0, 127.5, 255], np.float32)
    assert np.abs(O.color_map("hls_from_rgb_f32", src, 3) - want).max() < 1e-3


def test_hsv_hls_round_trip():
    f = O.pattern_u8(3 * 500).astype(np.float32)
    for fwd, inv in (("hsv_from_rgb_f32", "rgb_from_hsv_f32"), ("hls_from_rgb_f32", "rgb_from_hls_f32")):
        back = O.color_map(inv, O.color_map(fwd, f, 3), 3)
        assert np.abs(back - f).max() < 1e-2


def test_sepia_known_values():
    assert O.color_map("sepia_from_rgb_u8", np.array([255, 255, 255], np.uint8), 3).tolist() == [255, 255, 240]
    d = O.color_map("sepia_from_rgb_f32", np.array([100.0, 150.0, 200.0], np.float32), 3)
    want = [0.393 * 100 + 0.769 * 150 + 0.189 * 200, 0.349 * 100 + 0.686 * 150 + 0.168 * 200,
            0.272 * 100 + 0.534 * 150 + 0.131 * 200]
    assert np.abs(d - np.array(want, np.float32)).max() < 1e-3


def test_rgba_swizzles():
    src = np.array([0, 1, 2, 255, 3, 4, 5, 255], np.uint8)
    assert O.color_map("rgb_from_rgba_u8", src, 3, 0, None).tolist() == [0, 1, 2, 3, 4, 5]
    assert O.color_map("rgb_from_rgba_u8", src, 3, 1, None).tolist() == [2, 1, 0, 5, 4, 3]
    assert O.color_map("rgba_from_rgb_u8", np.array([9, 8, 7], np.uint8), 4, 0).tolist() == [9, 8, 7, 255]
    assert O.color_map("rgba_from_rgb_u8", np.array([9, 8, 7], np.uint8), 4, 1).tolist() == [7, 8, 9, 255]
    # alpha blend: a=0 -> background, a=255 -> colour, a=128 -> round(c*a + bg*(1-a))
    import ctypes as C
    bg = (C.c_uint8 * 3)(100, 50, 200)
    src = np.array([10, 20, 30, 0, 10, 20, 30, 255, 10, 20, 30, 128], np.uint8)
    got = O.color_map("rgb_from_rgba_u8", src, 3, 0, C.cast(bg, C.c_void_p)).tolist()
    a = np.float32(128) / np.float32(255)
    mid = [int(np.round(np.float32(c) * a + np.float32(b) * (np.float32(1) - a))) for c, b in zip((10, 20, 30), (100, 50, 200))]
    assert got == [100, 50, 200, 10, 20, 30] + mid


# ---- Bayer demosaic (P/color/bayer/mod.rs:100-203) -----------------------------------------------------

RAMP4 = np.array([10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110, 120, 130, 140, 150, 160], np.uint8).reshape(4, 4)


def _numpy_demosaic(m, pattern):
    """Independent form: replicate-pad, per-colour-plane bilinear fill with rounded means, then the cv2 frame rule."""
    h, w = m.shape
    p = np.pad(m.astype(np.int64), 1, mode="edge")
    n, s, wv, e = p[:-2, 1:-1], p[2:, 1:-1], p[1:-1, :-2], p[1:-1, 2:]
    nw, ne, sw_, se = p[:-2, :-2], p[:-2, 2:], p[2:, :-2], p[2:, 2:]
    cross, diag, horiz, vert = (n + s + wv + e + 2) >> 2, (nw + ne + sw_ + se + 2) >> 2, (wv + e + 1) >> 1, (n + s + 1) >> 1
    rr, cc = np.mgrid[0:h, 0:w]
    r0, c0 = {"rggb": (0, 0), "bggr": (1, 1), "grbg": (0, 1), "gbrg": (1, 0)}[pattern]  # where the red sensel sits in the 2x2 cell
    is_r = ((rr & 1) == r0) & ((cc & 1) == c0)
    is_b = ((rr & 1) == 1 - r0) & ((cc & 1) == 1 - c0)
    g_on_r_row = ((rr & 1) == r0) & ~is_r
    c = m.astype(np.int64)
    red = np.where(is_r, c, np.where(is_b, diag, np.where(g_on_r_row, horiz, vert)))
    blue = np.where(is_b, c, np.where(is_r, diag, np.where(g_on_r_row, vert, horiz)))
    green = np.where(is_r | is_b, cross, c)
    out = np.stack([red, green, blue], -1).astype(np.uint8)
    if h >= 3:
        out[0], out[-1] = out[1], out[-2]
    if w >= 3:
        out[:, 0], out[:, -1] = out[:, 1], out[:, -2]
    return out


def test_bayer_reference_known_answers():
    for p in O.BAYER:  # flat_mosaic_is_flat
        assert (O.rgb_from_bayer(np.full((6, 6), 200, np.uint8), p) == 200).all(), p
    out = O.rgb_from_bayer(RAMP4, "rggb")
    assert out[1, 1].tolist() == [60, 60, 60] and out[1, 2].tolist() == [70, 70, 70]  # rggb_interior_known_value
    assert out[0, 0].tolist() == [60, 60, 60] and np.array_equal(out[0, 2], out[1, 2])  # corners_use_replicate_border


def test_bayer_matches_independent_numpy_form():
    for (w, h) in [(4, 4), (5, 5), (7, 6), (33, 4), (40, 9), (3, 3), (2, 2), (1, 5), (6, 1), (1, 1), (2, 7)]:  # mod.rs:107-119 sizes + degenerate ones
        data = ((np.arange(w * h) * 37 + 11) % 256).astype(np.uint8).reshape(h, w)
        for p in O.BAYER:
            assert np.array_equal(O.rgb_from_bayer(data, p), _numpy_demosaic(data, p)), (p, w, h)
