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
