"""CPU pins for the u8 resize cascade and the OpenCV-compatible resize (oracle/ko_resize_u8.c):
the reference's cv2 golden vectors (tests/golden/opencv_resize, P/tests/opencv_compat.rs), its unit
tests (P/resize/opencv_compat.rs:253-330, P/resize/mod.rs:563-645) and independent numpy forms."""
import pathlib

import numpy as np
import pytest

import oracle_ffi as O

GOLDEN = pathlib.Path(__file__).parent / "golden" / "opencv_resize"
KEYS = sorted(p.name[:-4] for p in GOLDEN.glob("*.dst"))


def load_case(key):
    dtype, ch, s, _to, d, interp = key.split("_")
    c = int(ch[1:])
    sh, sw = map(int, s.split("x"))
    dh, dw = map(int, d.split("x"))
    dt = np.uint8 if dtype == "u8" else np.dtype("<f4")
    src = np.fromfile(GOLDEN / f"{key}.src", dt).reshape(sh, sw, c)
    want = np.fromfile(GOLDEN / f"{key}.dst", dt).reshape(dh, dw, c)
    return src, want, {"linear": "bilinear", "nearest": "nearest"}[interp]


def corridor_ok(got, want, mode):
    """opencv_compat.rs:81-125: exact for nearest, <= 2 LSB (u8) / <= 4 ulp (f32) for linear."""
    if got.dtype == np.uint8:
        d = np.abs(got.astype(np.int32) - want.astype(np.int32)).max()
        return d <= (0 if mode == "nearest" else 2), d
    d = np.abs(got.view(np.uint32).astype(np.int64) - np.ascontiguousarray(want).view(np.uint32).astype(np.int64)).max()
    return d <= (0 if mode == "nearest" else 4), d


def test_all_cv2_golden_vectors_are_present():
    assert len(KEYS) == 72


@pytest.mark.parametrize("key", KEYS)
def test_opencv_resize_matches_cv2_golden_vectors(key):
    src, want, mode = load_case(key)
    got = O.resize_opencv(src, want.shape[1], want.shape[0], mode)
    ok, d = corridor_ok(got, want, mode)
    assert ok, f"{key}: max deviation {d}"


def test_opencv_resize_unit_vectors():  # opencv_compat.rs:253-330
    src = np.array([[0, 100, 200, 255], [0, 100, 200, 255]], np.uint8)
    assert O.resize_opencv(src, 2, 1, "bilinear").reshape(-1).tolist() == [50, 228]
    assert O.resize_opencv(np.array([[10, 20, 30, 40]], np.uint8), 2, 1, "nearest").reshape(-1).tolist() == [10, 30]
    out = O.resize_opencv(np.array([[0.125, 0.875]], np.float32), 4, 1, "bilinear").reshape(-1)
    assert out[0] == 0.125 and out[3] == 0.875
    assert out[1] == np.float32(0.125 * 0.75 + 0.875 * 0.25) and out[2] == np.float32(0.125 * 0.25 + 0.875 * 0.75)
    with pytest.raises(ValueError):
        O.resize_opencv(src, 2, 2, "bicubic")


def test_resize_u8_routing():  # resize_u8_path, mod.rs:283-340
    rgb = O.pattern_u8(130 * 98 * 3).reshape(98, 130, 3)
    assert O.resize_fast_u8(rgb, 65, 49, "bilinear")[1] == "pyrdown2x"
    assert O.resize_fast_u8(rgb, 260, 196, "bilinear")[1] == "pyrup2x"
    assert O.resize_fast_u8(rgb, 64, 49, "bilinear")[1] == "bilinear"
    assert O.resize_fast_u8(rgb[:, :, :1], 65, 49, "bilinear")[1] == "bilinear"  # fast paths are RGB-only
    assert O.resize_fast_u8(rgb, 65, 49, "nearest")[1] == "nearest"
    assert O.resize_fast_u8(rgb, 65, 49, "bicubic")[1] == "separable"
    assert O.resize_fast_u8(rgb[:, :, :2], 65, 49, "nearest")[1] == "nearest"  # any channel count
    with pytest.raises(ValueError):
        O.resize_fast_u8(rgb[:, :, :2], 64, 48, "bilinear")
    with pytest.raises(ValueError):
        O.resize_fast_u8(rgb[:1], 64, 48, "bilinear")  # 1-pixel axis: typed error, mod.rs:318-324


def test_pyrdown_is_the_rounded_box_mean_and_pyrup_keeps_corners():  # mod.rs:593-645
    rgb = O.pattern_u8(34 * 12 * 3).reshape(12, 34, 3)
    down = O.resize_fast_u8(rgb, 17, 6)[0]
    s = rgb.astype(np.uint32)
    want = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(down, want.astype(np.uint8))
    for w, h in [(2, 2), (3, 4), (17, 9), (32, 5), (33, 6)]:
        img = (np.arange(w * h * 3) % 251).astype(np.uint8).reshape(h, w, 3)
        up, path = O.resize_fast_u8(img, 2 * w, 2 * h)
        assert path == "pyrup2x"
        for (dy, dx), (sy, sx) in [((0, 0), (0, 0)), ((0, -1), (0, -1)), ((-1, 0), (-1, 0)), ((-1, -1), (-1, -1))]:
            assert np.array_equal(up[dy, dx], img[sy, sx])
        # independent vectorised form: 75/25 blends as nested rounding halving adds, rows then columns
        s = img.astype(np.uint32)
        rh = lambda a, b: (a + b + 1) >> 1
        hz = np.empty((h, 2 * w, 3), np.uint32)
        hz[:, 0], hz[:, -1] = s[:, 0], s[:, -1]
        hz[:, 1:-1:2] = rh(s[:, :-1], rh(s[:, :-1], s[:, 1:]))
        hz[:, 2:-1:2] = rh(s[:, 1:], rh(s[:, :-1], s[:, 1:]))
        want = np.empty((2 * h, 2 * w, 3), np.uint32)
        want[0], want[-1] = hz[0], hz[-1]
        want[1:-1:2] = rh(hz[:-1], rh(hz[:-1], hz[1:]))
        want[2:-1:2] = rh(hz[1:], rh(hz[1:], hz[:-1]))
        assert np.array_equal(up, want.astype(np.uint8))


def test_q14_bilinear_against_float_form_and_identity():
    img = O.pattern_u8(63 * 41 * 3).reshape(41, 63, 3)
    assert np.array_equal(O.resize_fast_u8(img, 63, 41, "bilinear")[0], img)  # scale 1: fq == 0 everywhere
    got = O.resize_fast_u8(img, 127, 90, "bilinear")[0].astype(np.float64)
    ys = np.clip((np.arange(90) + 0.5) * (41 / 90) - 0.5, 0, 40)
    xs = np.clip((np.arange(127) + 0.5) * (63 / 127) - 0.5, 0, 62)
    y0, x0 = np.minimum(np.floor(ys).astype(int), 39), np.minimum(np.floor(xs).astype(int), 61)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    s = img.astype(np.float64)
    want = ((s[y0][:, x0] * (1 - fx) + s[y0][:, x0 + 1] * fx) * (1 - fy)
            + (s[y0 + 1][:, x0] * (1 - fx) + s[y0 + 1][:, x0 + 1] * fx) * fy)
    assert np.abs(got - want).max() <= 0.51


@pytest.mark.parametrize("filt", ["cubic", "lanczos3"])
@pytest.mark.parametrize("antialias", [True, False])
def test_contrib_tables(filt, antialias):  # common.rs:62-125
    for s, d in [(129, 64), (63, 127), (1024, 50), (33, 33)]:
        ofs, w = O.resize_contribs(s, d, filt, antialias)
        assert np.all(w.sum(axis=1) == 16384)
        support = (2.0 if filt == "cubic" else 3.0) * (max(s / d, 1.0) if antialias else 1.0)
        assert w.shape[1] == max(int(np.ceil(support)) * 2, 2)
        assert np.abs(w).max() <= 32767  # packs into i16 (pack_xw_i16)
    ofs, w = O.resize_contribs(33, 33, filt, antialias)  # identity grid: delta weights
    assert all(w[i].tolist().count(16384) == 1 and np.abs(w[i]).sum() == 16384 for i in range(33))


def test_separable_resize_properties():
    const = np.full((40, 50, 3), 93, np.uint8)
    img = O.pattern_u8(100 * 80 * 4).reshape(80, 100, 4)
    for mode in ("bicubic", "lanczos"):
        for aa in (True, False):
            assert np.array_equal(O.resize_fast_u8(const, 23, 17, mode, aa)[0], np.full((17, 23, 3), 93, np.uint8))
            assert np.array_equal(O.resize_fast_u8(img, 100, 80, mode, aa)[0], img)
    # antialiased downscale of a smooth ramp stays close to the ideal ramp
    ramp = np.tile(np.linspace(0, 255, 256).astype(np.uint8)[None, :, None], (64, 1, 1))
    got = O.resize_fast_u8(ramp, 64, 16, "lanczos", True)[0][:, 2:-2, 0].astype(np.float64)
    want = np.linspace(0, 255, 256).reshape(64, 4).mean(axis=1)[2:-2]
    assert np.abs(got - want[None, :]).max() <= 2.0


# ---- fused resize + normalise + CHW (P/resize/fused.rs) ----------------------------------------------------
IMAGENET = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])


def test_fused_2x_matches_f64_reference():  # fused.rs:1046-1090
    dw, dh = 37, 5
    src = ((np.arange(2 * dh * 2 * dw * 3) * 7 + 3) % 256).astype(np.uint8).reshape(2 * dh, 2 * dw, 3)
    scale, bias = O.normalize_params(*IMAGENET)
    got, path = O.resize_normalize_to_chw(src, dw, dh, scale, bias, "bilinear")
    assert path == "box2x"
    s = src.astype(np.float64)
    avg = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2]) / 4.0
    want = ((avg / 255.0 - np.array(IMAGENET[0])) / np.array(IMAGENET[1])).transpose(2, 0, 1)
    assert np.abs(got - want).max() < 1e-4
    zero, _ = O.resize_normalize_to_chw(np.zeros((4, 32, 3), np.uint8), 16, 2, *O.normalize_params([0.5, 0.25, 0.75], [0.5, 0.25, 0.75]))
    assert np.abs(zero + 1.0).max() < 1e-6  # fused_2x_normalize_zero_input: (0 - mean)/std = -1


@pytest.mark.parametrize("mode", ["nearest", "bilinear", "bicubic", "lanczos"])
def test_fused_paths_track_resize_then_normalize(mode):
    src = O.pattern_u8(96 * 64 * 3).reshape(64, 96, 3)
    from test_oracle_u8 import hash_image
    src = O.gaussian_blur_u8(hash_image(64, 96, 3), (7, 7), (2.0, 2.0))[0]  # smooth: resampling differences stay small
    scale, bias = O.normalize_params(*IMAGENET)
    got, path = O.resize_normalize_to_chw(src, 40, 30, scale, bias, mode, True)
    assert path == {"nearest": "nearest", "bilinear": "bilinear"}.get(mode, "separable")
    u8 = O.resize_fast_u8(src, 40, 30, mode, True)[0].astype(np.float32)
    want = (u8 * scale + bias).transpose(2, 0, 1)
    tol = {"nearest": 1e-6, "bilinear": 0.03, "bicubic": 0.012, "lanczos": 0.012}[mode]  # u8 rounding = 0.5/255/std
    assert np.abs(got - want).max() <= tol
    const = np.full((9, 14, 3), 200, np.uint8)
    got, _ = O.resize_normalize_to_chw(const, 5, 4, scale, bias, mode, True)
    assert np.abs(got - (np.float32(200) * scale + bias)[:, None, None]).max() < 1e-5
