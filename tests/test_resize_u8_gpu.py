"""GPU parity for the u8 resize cascade (kh_resize_fast_u8) and the OpenCV-compatible resize
(kh_resize_opencv_{u8,f32}): byte-exact against the CPU oracle on the reference's device==host shapes
(P/resize/cuda.rs:386-470) and on the reference's cv2 golden vectors (tests/golden/opencv_resize)."""
import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits, dev, out_buf
from test_oracle_resize_u8 import KEYS, corridor_ok, load_case

pytestmark = pytest.mark.gpu


def pat(w, h, c, seed=0):
    return np.roll(O.pattern_u8(w * h * c + seed), -seed)[: w * h * c].reshape(h, w, c).copy()


def resize_gpu(gpu_stream, src, dw, dh, mode, antialias=True, batch=1):
    from kornia_rs import _ffi
    h, w, c = src.shape[-3:]
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, batch * dh * dw * c)
    rc = _ffi.lib.kh_resize_fast_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, dw, dh, c, O.MODE[mode],
                                    int(antialias), batch, h * w * c, dh * dw * c)
    if rc != 0:
        return rc
    return d_dst.to_numpy(np.uint8, (batch, dh, dw, c))


def check(gpu_stream, s, d, c, mode, aa=True):
    src = pat(s[0], s[1], c)
    got = resize_gpu(gpu_stream, src, d[0], d[1], mode, aa)[0]
    want, path = O.resize_fast_u8(src, d[0], d[1], mode, aa)
    assert_same_bits(got, want, f"{s}->{d} c{c} {mode} aa={aa} ({path})")
    return path


def test_pyr2x_fast_paths(gpu_stream):  # cuda.rs:386-392
    assert check(gpu_stream, (130, 98), (65, 49), 3, "bilinear") == "pyrdown2x"
    assert check(gpu_stream, (65, 49), (130, 98), 3, "bilinear") == "pyrup2x"
    for w, h in [(2, 2), (3, 4), (17, 9), (32, 5), (33, 6)]:  # mod.rs:593-645
        assert check(gpu_stream, (w, h), (2 * w, 2 * h), 3, "bilinear") == "pyrup2x"
        assert check(gpu_stream, (2 * w, 2 * h), (w, h), 3, "bilinear") == "pyrdown2x"
    assert check(gpu_stream, (130, 98), (65, 49), 1, "bilinear") == "bilinear"  # RGB-only fast paths


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_nearest(gpu_stream, c):  # cuda.rs:394-403
    for s, d in [((129, 97), (64, 48)), ((63, 41), (127, 90)), ((1, 1), (5, 3)), ((7, 5), (1, 1))]:
        assert check(gpu_stream, s, d, c, "nearest") == "nearest"


@pytest.mark.parametrize("c", [1, 3, 4])
def test_bilinear_q14(gpu_stream, c):  # cuda.rs:405-418
    for s, d in [((129, 97), (64, 48)), ((63, 41), (127, 90)), ((33, 21), (33, 21)), ((2, 2), (9, 7)), ((1920, 1080), (224, 224))]:
        assert check(gpu_stream, s, d, c, "bilinear") == "bilinear"


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("mode", ["nearest", "bilinear"])
def test_simple_paths_four_pixels_per_lane(gpu_stream, dev_option, c, mode):
    """nearest / Q14 bilinear / the exact-2x RGB paths write four consecutive pixels of a row per lane as dwords where the destination rows
    are whole quads on 4-byte-aligned images (round 6); test option resize_u8_px = 1 keeps one pixel per thread with byte stores.  Same
    bytes as the oracle: down- and upscales, exact 2x both ways, widths that leave partial 256-pixel tiles, a batch whose destination
    stride is / is not a multiple of four (the latter keeps the byte-store kernel), a destination 2 bytes off alignment."""
    from kornia_rs import _ffi
    if mode == "bilinear" and c == 2:
        pytest.skip("Q14 bilinear: 1, 3 or 4 channels")
    shapes = [((129, 97), (64, 48)), ((63, 41), (128, 90)), ((130, 98), (260, 196)), ((264, 40), (132, 20)), ((1920, 24), (1280, 16)), ((50, 30), (300, 7)), ((40, 9), (4, 3))]
    for s, d in shapes:
        src = pat(s[0], s[1], c, seed=3)
        want, _ = O.resize_fast_u8(src, d[0], d[1], mode, True)
        for opt in (-1, 1, 4):   # the launcher's choice (quads for bilinear only up to a 2x downscale), one pixel per thread, quads wherever possible
            dev_option("resize_u8_px", opt)
            assert_same_bits(resize_gpu(gpu_stream, src, d[0], d[1], mode)[0], want, f"{s}->{d} c{c} {mode} option {opt}")
    dev_option("resize_u8_px", -1)
    n, (sw, sh), (dw, dh) = 3, (96, 40), (132, 27)   # dw * dh * c: a multiple of four only for some c -> both kernels across the parametrisation
    srcs = np.stack([pat(sw, sh, c, seed=31 * k) for k in range(n)])
    got = resize_gpu(gpu_stream, srcs, dw, dh, mode, batch=n)
    for k in range(n):
        assert_same_bits(got[k], O.resize_fast_u8(srcs[k], dw, dh, mode, True)[0], f"batch image {k} c{c} {mode}")
    src = pat(sw, sh, c)
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, dh * dw * c + 8)
    _ffi.check(_ffi.lib.kh_resize_fast_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr + 2, sw, sh, dw, dh, c, O.MODE[mode], 1, 1, 0, 0))
    assert_same_bits(d_dst.to_numpy(np.uint8, (dh * dw * c + 8,))[2:2 + dh * dw * c].reshape(dh, dw, c), O.resize_fast_u8(src, dw, dh, mode, True)[0], "destination 2 bytes off")


def test_nearest_many_ratios(gpu_stream, dev_option):
    """Nearest over many width ratios — upscales, downscales, primes, ratios whose f64 column product lands on or next to whole numbers
    (written for a round-6 experiment that divided in integers where the host proved the two forms equal: no faster, not kept; the
    cases stay) — quads and one pixel per thread."""
    rng = np.random.default_rng(11)
    pairs = [(63, 127), (127, 63), (100, 300), (300, 100), (1920, 1280), (1280, 1920), (7, 1000), (1000, 7), (333, 999), (997, 1009), (1009, 997), (64, 4096), (3, 5), (5, 3), (640, 224), (224, 640)]
    pairs += [(int(rng.integers(1, 1500)), int(rng.integers(1, 1500))) for _ in range(24)]
    for (sw, dw) in pairs:
        for c in (1, 3):
            src = pat(sw, 3, c, seed=sw + dw)
            want, _ = O.resize_fast_u8(src, dw, 5, "nearest", True)
            for opt in (-1, 1):
                dev_option("resize_u8_px", opt)
                assert_same_bits(resize_gpu(gpu_stream, src, dw, 5, "nearest")[0], want, f"nearest {sw}->{dw} c{c} resize_u8_px={opt}")
    dev_option("resize_u8_px", -1)


@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("api", ["fast", "opencv"])
def test_nearest_upscale_of_one_channel_with_column_selectors(gpu_stream, dev_option, api, c):
    """One-channel nearest upscales (label maps, masks): a lane owns sixteen destination columns for a strip of rows, evaluates the
    reference's column index once and re-indexes the sixteen source bytes at its first column with byte selectors (round 6).  The
    oracle's bytes for integer and fractional factors from 1x to 8x, widths that are not multiples of 16 or 4, the narrowest source
    (16 pixels), destination heights around the 32-row strips, a batch, a destination off a dword; resize_u8_px = 2 keeps the quad /
    per-pixel kernels; both index formulas (resize_fast_u8 and the cv2-compatible one)."""
    from kornia_rs import _ffi
    rng = np.random.default_rng(17)
    run = (lambda s_, dw_, dh_: resize_gpu(gpu_stream, s_, dw_, dh_, "nearest")[0]) if api == "fast" else (lambda s_, dw_, dh_: cv_gpu(gpu_stream, s_, dw_, dh_, "nearest"))
    ref = (lambda s_, dw_, dh_: O.resize_fast_u8(s_, dw_, dh_, "nearest", True)[0]) if api == "fast" else (lambda s_, dw_, dh_: O.resize_opencv(s_, dw_, dh_, "nearest"))
    cases = [(16, 3, 16, 3), (16, 2, 17, 5), (16, 4, 128, 33), (17, 5, 51, 64), (100, 7, 300, 21), (100, 9, 257, 31), (640, 6, 1920, 18), (333, 4, 1000, 9), (960, 5, 3840, 40),
             (1000, 3, 1001, 3), (4096, 2, 4100, 5), (63, 9, 4097, 11), (20, 30, 37, 65), (6, 3, 9, 4), (6, 2, 48, 5), (100, 4, 150, 6), (100, 4, 149, 6), (101, 3, 203, 7)]
    # (three channels: the same kernel on the row BYTES, where every lane's sixteen bytes come from at most sixteen source bytes — factors of about 1.5 and more)
    for (sw, sh, dw, dh) in cases:
        src = rng.integers(0, 256, (sh, sw, c), dtype=np.uint8)
        want = ref(src, dw, dh)
        for opt in ((-1, 2) if sw in (16, 100, 333, 63) else (-1,)):
            dev_option("resize_u8_px", opt)
            assert_same_bits(run(src, dw, dh), want, f"{api} nearest upscale {sw}x{sh} -> {dw}x{dh} resize_u8_px={opt}")
    dev_option("resize_u8_px", -1)
    sw, sh, dw, dh, n = 301, 7, 903, 20, 3
    src = rng.integers(0, 256, (n, sh, sw, c), dtype=np.uint8)
    m = dw * dh * c
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, n * m + 8)
    fn = _ffi.lib.kh_resize_fast_u8 if api == "fast" else _ffi.lib.kh_resize_opencv_u8
    args = (gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr + 3, sw, sh, dw, dh, c, O.MODE["nearest"]) + ((1,) if api == "fast" else ()) + (n, sw * sh * c, m)
    _ffi.check(fn(*args))
    got = d_dst.to_numpy(np.uint8, (n * m + 8,))
    assert got[:3].tolist() == [255] * 3 and got[3 + n * m:3 + n * m + 5].tolist() == [255] * 5, "bytes outside the destination were written"
    for i in range(n):
        assert_same_bits(got[3 + i * m:3 + (i + 1) * m].reshape(dh, dw, c), ref(src[i], dw, dh), f"offset destination frame {i}")


@pytest.mark.parametrize("c", [1, 4])
def test_exact_half_bilinear_box_for_one_and_four_channels(gpu_stream, dev_option, c):
    """The reference has the exact-2x box only for RGB; on 1 / 4 channels its generic Q14 bilinear has fx = fy = 8192 at that scale and
    equals (p00 + p01 + p10 + p11 + 2) >> 2 exactly, which the quad kernel computes on packed bytes (round 6): the oracle's bytes (its
    generic path) for destination rows of whole quads — every byte value meets every other in the pattern — next to sizes that keep the
    generic kernel (destination width not a multiple of four), a batch; resize_u8_px = 2 keeps the generic quad kernel."""
    for (dw, dh) in [(4, 1), (8, 3), (64, 5), (256, 4), (260, 3), (1024, 2), (1028, 3), (960, 7), (6, 4), (65, 9)]:
        src = pat(2 * dw, 2 * dh, c, seed=dw + dh)
        want, path = O.resize_fast_u8(src, dw, dh, "bilinear", True)
        assert path == "bilinear"
        for opt in (-1, 2):
            dev_option("resize_u8_px", opt)
            assert_same_bits(resize_gpu(gpu_stream, src, dw, dh, "bilinear")[0], want, f"exact half c{c} -> {dw}x{dh} resize_u8_px={opt}")
    dev_option("resize_u8_px", -1)
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (3, 36, 520, c), dtype=np.uint8)
    got = resize_gpu(gpu_stream, src, 260, 18, "bilinear", batch=3)
    for k in range(3):
        assert_same_bits(got[k], O.resize_fast_u8(src[k], 260, 18, "bilinear", True)[0], f"batch frame {k}")
    sat = np.full((8, 16, c), 255, np.uint8); sat[::2, 1::2] = 254   # sums that reach 4 * 255 and the rounding either side
    assert_same_bits(resize_gpu(gpu_stream, sat, 8, 4, "bilinear")[0], O.resize_fast_u8(sat, 8, 4, "bilinear", True)[0], "saturated")


@pytest.mark.parametrize("c", [1, 3, 4])
def test_exact_double_bilinear_on_the_rolling_kernels(gpu_stream, dev_option, c):
    """The exact 2x bilinear upscale runs on the rolling pyrup kernels with this resize's arithmetic (round 6): the reference's
    rounding-halving chains for RGB (its pyrup2x path), the generic Q14 weights — (9 a + 3 b + 3 c + d + 8) >> 4 at this scale — for
    one / four channels, replicate borders.  The oracle's bytes on widths either side of the lane / wave / block seams, the narrowest
    sources each kernel takes and the ones it leaves to the per-pixel form, ragged gray widths, two-row sources, a batch, a destination
    off a dword; resize_u8_px = 2 keeps the per-pixel kernels."""
    from kornia_rs import _ffi
    sizes = [(2, 2), (3, 4), (4, 2), (5, 3), (7, 9), (8, 2), (9, 5), (12, 3), (17, 9), (32, 5), (33, 6), (255, 3), (256, 4), (257, 5), (260, 3), (511, 2), (512, 3), (513, 4), (1023, 3), (1024, 2), (1025, 3), (2049, 2), (300, 131)]
    for (w, h) in sizes:
        src = pat(w, h, c, seed=w + h)
        want, path = O.resize_fast_u8(src, 2 * w, 2 * h, "bilinear", True)
        assert path == ("pyrup2x" if c == 3 else "bilinear")
        near = O.resize_fast_u8(src, 2 * w, 2 * h, "nearest", True)[0]
        for opt in ((-1, 2) if w in (2, 9, 257, 1025, 300) else (-1,)):
            dev_option("resize_u8_px", opt)
            assert_same_bits(resize_gpu(gpu_stream, src, 2 * w, 2 * h, "bilinear")[0], want, f"exact double c{c} {w}x{h} resize_u8_px={opt}")
            assert_same_bits(resize_gpu(gpu_stream, src, 2 * w, 2 * h, "nearest")[0], near, f"exact double nearest c{c} {w}x{h} resize_u8_px={opt}")   # (the same walk, no arithmetic)
    dev_option("resize_u8_px", -1)
    rng = np.random.default_rng(7)
    src = rng.integers(0, 256, (3, 19, 301, c), dtype=np.uint8)
    got = resize_gpu(gpu_stream, src, 602, 38, "bilinear", batch=3)
    for k in range(3):
        assert_same_bits(got[k], O.resize_fast_u8(src[k], 602, 38, "bilinear", True)[0], f"batch frame {k}")
    w, h, n = 301, 7, 2
    src = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
    dw, dh = 2 * w, 2 * h
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, n * dw * dh * c + 8)
    _ffi.check(_ffi.lib.kh_resize_fast_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr + 3, w, h, dw, dh, c, O.MODE["bilinear"], 1, n, w * h * c, dw * dh * c))
    got = d_dst.to_numpy(np.uint8, (n * dw * dh * c + 8,))
    assert got[:3].tolist() == [255] * 3 and got[3 + n * dw * dh * c:3 + n * dw * dh * c + 5].tolist() == [255] * 5, "bytes outside the destination were written"
    for i in range(n):
        assert_same_bits(got[3 + i * dw * dh * c:3 + (i + 1) * dw * dh * c].reshape(dh, dw, c), O.resize_fast_u8(src[i], dw, dh, "bilinear", True)[0], f"offset destination frame {i}")


@pytest.mark.parametrize("mode", ["bicubic", "lanczos"])
@pytest.mark.parametrize("aa", [True, False])
def test_separable_q14(gpu_stream, mode, aa):  # cuda.rs:420-440
    for s, d, c in [((129, 97), (64, 48), 3), ((63, 41), (127, 90), 1), ((100, 80), (47, 33), 4), ((33, 21), (33, 21), 3),
                    ((1, 1), (4, 4), 1), ((5, 1), (2, 3), 3)]:
        assert check(gpu_stream, s, d, c, mode, aa) == "separable"


@pytest.mark.parametrize("path", ["staged", "gather"])
@pytest.mark.parametrize("c", [1, 3, 4])
def test_separable_staged_horizontal_pass_tiles(gpu_stream, c, path, dev_option):
    """The LDS-staged passes (horizontal: 64 destination columns x 16 source rows per block, planar signed bytes + v_dot4; vertical:
    128 flat columns x a segment of destination rows): several interior tiles whose rows start at every address alignment (odd
    widths), ragged last tiles, 4-pixel groups that straddle the image border, upsampling, every plane-pitch class, vertical windows
    of 6 .. 390 taps, and spans / windows too large for the LDS budgets (the per-tap kernels); `gather` forces those fallbacks for
    every case."""
    if path == "gather":
        dev_option("resize_u8_gather", 1)
    for s, d, mode, aa in [((517, 70), (300, 40), "lanczos", True), ((1001, 37), (230, 37), "lanczos", True), ((333, 50), (700, 50), "lanczos", False),
                           ((415, 35), (200, 20), "bicubic", True), ((415, 35), (200, 20), "bicubic", False), ((20000, 4), (70, 4), "lanczos", True),
                           ((1100, 20), (64, 20), "lanczos", True),  # 17x
                           ((3000, 20), (200, 20), "lanczos", True),   # 15x: the widest plane-pitch class of the v_dot4 kernel (1280 B)
                           ((40, 600), (30, 50), "lanczos", True),     # 12x vertically: 72-tap windows, segments of 8 destination rows
                           ((40, 2000), (30, 50), "bicubic", True),    # 40x vertically: 160-tap windows, segments of 1 row
                           ((24, 2600), (24, 40), "lanczos", True),    # 65x vertically: windows beyond the LDS budget (the per-tap vertical kernel)
                           ((5, 40), (3, 17), "lanczos", True)]:       # rows narrower than one 4-pixel staging group on the right edge
        assert check(gpu_stream, s, d, c, mode, aa) == "separable"


def test_separable_extreme_downscale_and_batch(gpu_stream):  # cuda.rs:442-448
    check(gpu_stream, (1024, 64), (50, 40), 3, "lanczos", True)
    check(gpu_stream, (1024, 64), (50, 40), 3, "bicubic", True)
    n = 5
    src = np.stack([pat(640, 360, 3, seed=31 * k) for k in range(n)])
    for mode in ("bicubic", "bilinear", "nearest"):
        got = resize_gpu(gpu_stream, src, 224, 224, mode, True, batch=n)
        for k in range(n):
            assert_same_bits(got[k], O.resize_fast_u8(src[k], 224, 224, mode, True)[0], f"{mode} frame {k}")
    got = resize_gpu(gpu_stream, src, 320, 180, "bilinear", True, batch=n)  # pyrdown in a batch
    for k in range(n):
        assert_same_bits(got[k], O.resize_fast_u8(src[k], 320, 180, "bilinear")[0], f"pyrdown frame {k}")
    # second call with the same geometry hits the contribution-table cache
    check(gpu_stream, (1024, 64), (50, 40), 3, "lanczos", True)


def test_error_semantics(gpu_stream):  # cuda.rs:450-470, mod.rs:310-325
    from kornia_rs import _ffi
    assert resize_gpu(gpu_stream, pat(8, 8, 2), 4, 4, "bilinear") == _ffi.KH_ERR_UNSUPPORTED
    assert resize_gpu(gpu_stream, pat(8, 8, 2), 4, 4, "bicubic") == _ffi.KH_ERR_UNSUPPORTED
    assert resize_gpu(gpu_stream, pat(8, 1, 3), 4, 4, "bilinear") == _ffi.KH_ERR_INVALID_ARG
    assert "2x2" in _ffi.last_error()
    assert resize_gpu(gpu_stream, pat(8, 8, 3), 4, 0, "nearest") == _ffi.KH_ERR_INVALID_ARG


# ---- OpenCV-compatible ---------------------------------------------------------------------------------

def cv_gpu(gpu_stream, src, dw, dh, mode):
    from kornia_rs import _ffi
    h, w, c = src.shape
    f32 = src.dtype == np.float32
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, dh * dw * c * (4 if f32 else 1))
    fn = _ffi.lib.kh_resize_opencv_f32 if f32 else _ffi.lib.kh_resize_opencv_u8
    rc = fn(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, dw, dh, c, O.MODE[mode], 1, 0, 0)
    if rc != 0:
        return rc
    return d_dst.to_numpy(src.dtype, (dh, dw, c))


@pytest.mark.parametrize("key", KEYS)
def test_opencv_resize_golden_vectors(gpu_stream, key, dev_option):
    src, want, mode = load_case(key)
    src = np.ascontiguousarray(src.astype(src.dtype.newbyteorder("=")))
    got = cv_gpu(gpu_stream, src, want.shape[1], want.shape[0], mode)
    assert_same_bits(got, O.resize_opencv(src, want.shape[1], want.shape[0], mode), key)  # bit-exact vs oracle
    ok, d = corridor_ok(got, want, mode)                                                   # reference corridor vs cv2
    assert ok, f"{key}: max deviation {d}"
    for opt in (1, 4):   # u8: one pixel per thread / four pixels per lane wherever the rows are whole quads (round 6) — the same bytes
        dev_option("resize_u8_px", opt)
        assert_same_bits(cv_gpu(gpu_stream, src, want.shape[1], want.shape[0], mode), got, f"{key} resize_u8_px = {opt}")


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("mode", ["nearest", "bilinear"])
def test_opencv_resize_u8_four_pixels_per_lane(gpu_stream, dev_option, c, mode):
    """kh_resize_opencv_u8 through the quad kernel (destination widths that are multiples of four) and the byte-store kernel: oracle's bytes."""
    for (sw, sh), (dw, dh) in [((129, 97), (64, 48)), ((63, 41), (128, 90)), ((264, 40), (132, 20)), ((50, 30), (300, 7)), ((1920, 12), (1280, 8)), ((40, 9), (4, 3))]:
        src = pat(sw, sh, c, seed=9)
        want = O.resize_opencv(src, dw, dh, mode)
        for opt in (-1, 1, 4):
            dev_option("resize_u8_px", opt)
            assert_same_bits(cv_gpu(gpu_stream, src, dw, dh, mode), want, f"cv u8 c{c} {mode} {sw}x{sh}->{dw}x{dh} option {opt}")


@pytest.mark.parametrize("c", [1, 3, 4])
def test_opencv_linear_exact_half_is_the_box(gpu_stream, dev_option, c):
    """INTER_LINEAR at an exact 2x downscale: every coefficient is 1024 / 2048 and the reference's shifts drop only zero bits, so the result
    is (p00 + p01 + p10 + p11 + 2) >> 2 — the packed-byte box kernel (round 6).  The oracle's bytes (its generic fixed-point path) on whole-quad
    destination rows, random data and saturated sums; resize_u8_px = 2 keeps the generic quad kernel; other widths never leave it."""
    rng = np.random.default_rng(3)
    for (dw, dh) in [(4, 1), (8, 3), (64, 5), (256, 4), (260, 3), (1024, 2), (1028, 3), (6, 4), (65, 9)]:
        src = rng.integers(0, 256, (2 * dh, 2 * dw, c), dtype=np.uint8)
        want = O.resize_opencv(src, dw, dh, "bilinear")
        for opt in (-1, 2):
            dev_option("resize_u8_px", opt)
            assert_same_bits(cv_gpu(gpu_stream, src, dw, dh, "bilinear"), want, f"cv linear exact half c{c} -> {dw}x{dh} resize_u8_px={opt}")
    dev_option("resize_u8_px", -1)
    sat = np.full((8, 16, c), 255, np.uint8); sat[::2, 1::2] = 254
    assert_same_bits(cv_gpu(gpu_stream, sat, 8, 4, "bilinear"), O.resize_opencv(sat, 8, 4, "bilinear"), "saturated")
    # INTER_NEAREST at an exact 2x upscale takes the rolling walk (column i >> 1) on one / three channels
    for (w, h) in [(2, 2), (4, 3), (9, 5), (33, 6), (256, 4), (257, 5), (513, 3), (1025, 2), (300, 41)]:
        src = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        want = O.resize_opencv(src, 2 * w, 2 * h, "nearest")
        lin = O.resize_opencv(src, 2 * w, 2 * h, "bilinear")   # INTER_LINEAR: the same walk with the reference's fixed-point arithmetic
        for opt in (-1, 2):
            dev_option("resize_u8_px", opt)
            assert_same_bits(cv_gpu(gpu_stream, src, 2 * w, 2 * h, "nearest"), want, f"cv nearest exact double c{c} {w}x{h} resize_u8_px={opt}")
            assert_same_bits(cv_gpu(gpu_stream, src, 2 * w, 2 * h, "bilinear"), lin, f"cv linear exact double c{c} {w}x{h} resize_u8_px={opt}")
    dev_option("resize_u8_px", -1)
    for v0, v1 in ((255, 254), (1, 0), (3, 2), (251, 255)):   # the truncations of (h >> 2) + ((3 h) >> 2) on values around multiples of four
        edge = np.full((6, 12, c), v0, np.uint8); edge[1::2, ::3] = v1; edge[:, 5] = (v0 + v1) // 2
        assert_same_bits(cv_gpu(gpu_stream, edge, 24, 12, "bilinear"), O.resize_opencv(edge, 24, 12, "bilinear"), f"cv linear exact double values {v0} {v1}")


def test_opencv_resize_unit_vectors_and_channels(gpu_stream):  # opencv_compat.rs:253-330
    from kornia_rs import _ffi
    src = np.array([[0, 100, 200, 255], [0, 100, 200, 255]], np.uint8)[:, :, None]
    assert cv_gpu(gpu_stream, src, 2, 1, "bilinear").reshape(-1).tolist() == [50, 228]
    assert cv_gpu(gpu_stream, np.array([[10, 20, 30, 40]], np.uint8)[:, :, None], 2, 1, "nearest").reshape(-1).tolist() == [10, 30]
    out = cv_gpu(gpu_stream, np.array([[0.125, 0.875]], np.float32)[:, :, None], 4, 1, "bilinear").reshape(-1)
    assert out.tolist() == [0.125, np.float32(0.125 * 0.75 + 0.875 * 0.25), np.float32(0.125 * 0.25 + 0.875 * 0.75), 0.875]
    assert cv_gpu(gpu_stream, src, 2, 2, "bicubic") == _ffi.KH_ERR_UNSUPPORTED
    for c in (2, 4):
        u8 = pat(37, 23, c)
        f32 = O.pattern_f32(37 * 23 * c).reshape(23, 37, c)
        for img in (u8, f32):
            for mode in ("nearest", "bilinear"):
                for dw, dh in [(11, 17), (80, 51), (37, 23)]:
                    assert_same_bits(cv_gpu(gpu_stream, img, dw, dh, mode), O.resize_opencv(img, dw, dh, mode),
                                     f"cv {img.dtype} c{c} {mode} {dw}x{dh}")


# ---- fused RGB8 -> normalised CHW f32 (P/resize/fused.rs) -------------------------------------------------

def fused_gpu(gpu_stream, src, dw, dh, scale, bias, mode, aa=True, batch=1):
    import ctypes as C
    from kornia_rs import _ffi
    h, w, c = src.shape[-3:]
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, batch * 3 * dh * dw * 4)
    f3 = C.c_float * 3
    _ffi.check(_ffi.lib.kh_resize_normalize_to_chw_u8_f32(
        gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, dw, dh, f3(*[float(v) for v in scale]),
        f3(*[float(v) for v in bias]), O.MODE[mode], int(aa), batch, h * w * 3, 3 * dh * dw))
    return d_dst.to_numpy(np.float32, (batch, 3, dh, dw))


@pytest.mark.parametrize("mode", ["nearest", "bilinear", "bicubic", "lanczos"])
def test_fused_resize_normalize_matches_oracle(gpu_stream, mode):
    scale, bias = O.normalize_params([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    for (sw, sh), (dw, dh) in [((74, 10), (37, 5)), ((129, 97), (64, 48)), ((63, 41), (127, 90)), ((33, 21), (33, 21)),
                               ((1, 1), (4, 3)), ((640, 360), (224, 224))]:
        src = pat(sw, sh, 3)
        for aa in ((True, False) if mode in ("bicubic", "lanczos") else (True,)):
            got = fused_gpu(gpu_stream, src, dw, dh, scale, bias, mode, aa)[0]
            want, path = O.resize_normalize_to_chw(src, dw, dh, scale, bias, mode, aa)
            assert_same_bits(got, want, f"fused {mode} aa={aa} {sw}x{sh}->{dw}x{dh} ({path})")


def test_fused_resize_normalize_batch_and_host_api(gpu_stream):
    from kornia_rs import Image, imgproc
    scale, bias = O.normalize_params([0.5, 0.25, 0.75], [0.5, 0.25, 0.75])
    n = 4
    src = np.stack([pat(640, 360, 3, seed=31 * k) for k in range(n)])
    for mode, (dw, dh) in [("bilinear", (320, 180)), ("bilinear", (224, 224)), ("lanczos", (224, 224))]:
        got = fused_gpu(gpu_stream, src, dw, dh, scale, bias, mode, True, batch=n)
        for k in range(n):
            assert_same_bits(got[k], O.resize_normalize_to_chw(src[k], dw, dh, scale, bias, mode, True)[0], f"{mode} frame {k}")
    img = Image.from_numpy(src[0]).to_hip(gpu_stream)
    t = img.resize_normalize_to_tensor(224, 224, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    sc, bi = O.normalize_params([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    assert t.shape == (3, 224, 224) and t.dtype == "float32" and t.is_device
    assert_same_bits(t.numpy(), O.resize_normalize_to_chw(src[0], 224, 224, sc, bi, "bilinear")[0], "Image.resize_normalize_to_tensor")
    psc, pbi = imgproc.normalize_params([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    assert np.array_equal(psc, sc) and np.array_equal(pbi, bi)
