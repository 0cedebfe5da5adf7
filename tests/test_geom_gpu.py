"""GPU parity for f32 resize / warp_affine / warp_perspective / remap / undistort maps.

Bit-exact against the CPU oracle (same expression trees, uncontracted f32) — stricter than the
1e-6 the north star allows.  Shapes follow the reference's own device==host tests
(P/resize/cuda.rs:520-577, P/cuda/warp_perspective.rs:785-822, P/warp/cuda.rs:174-329,
P/cuda/remap.rs:770-804)."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits, dev, fptr, out_buf

pytestmark = pytest.mark.gpu
MODES = ["nearest", "bilinear", "bicubic", "lanczos"]


def img(w, h, c, seed=0):
    return np.roll(O.pattern_f32(w * h * c + seed), -seed)[: w * h * c].reshape(h, w, c).copy()


def call(gpu_stream, name, *args):
    from kornia_rs import _ffi
    _ffi.check(getattr(_ffi.lib, name)(gpu_stream.cuda_stream_ptr, *args))


def resize_gpu(gpu_stream, src, dw, dh, mode, batch=1):
    n = batch
    h, w, c = src.shape[-3:]
    d_src = dev(gpu_stream, src)
    d_dst = out_buf(gpu_stream, n * dh * dw * c * 4)
    call(gpu_stream, "kh_resize_f32", d_src.ptr, d_dst.ptr, w, h, dw, dh, c, O.MODE[mode], n, h * w * c, dh * dw * c)
    return d_dst.to_numpy(np.float32, (n, dh, dw, c))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shape", [(129, 97, 64, 48), (63, 41, 127, 90), (64, 48, 64, 48), (7, 5, 1, 1), (1, 1, 9, 4)])
@pytest.mark.parametrize("c", [1, 3, 4])
def test_resize_matches_oracle(gpu_stream, mode, shape, c):
    sw, sh, dw, dh = shape
    src = img(sw, sh, c)
    got = resize_gpu(gpu_stream, src, dw, dh, mode)[0]
    assert_same_bits(got, O.resize(src, dw, dh, mode), f"resize {shape} c{c} {mode}")


@pytest.mark.parametrize("mode", ["nearest", "bilinear", "bicubic"])
def test_resize_one_channel_four_pixels_per_lane(gpu_stream, dev_option, mode):
    """One-channel resizes whose destination rows are whole float4s take four pixels per lane with 16-byte stores (round 6): the oracle's
    bits on up- and downscales, destination widths either side of the 256-pixel tile row, a batch and a list; resize_rows = 0 keeps one
    pixel per lane; widths that are not a multiple of four never leave it."""
    for (sw, sh, dw, dh) in [(129, 97, 64, 48), (63, 41, 128, 90), (64, 48, 256, 7), (100, 9, 260, 20), (300, 5, 1024, 3), (301, 6, 1028, 4), (2, 2, 8, 8), (1, 1, 4, 3), (500, 40, 252, 21), (90, 30, 127, 40)]:
        n = 2
        src = np.stack([img(sw, sh, 1, seed=31 * k + sw) for k in range(n)])
        want = np.stack([O.resize(src[k], dw, dh, mode) for k in range(n)])
        for opt in (-1, 0):
            dev_option("resize_rows", opt)
            assert_same_bits(resize_gpu(gpu_stream, src, dw, dh, mode, batch=n), want, f"one channel {mode} {sw}x{sh} -> {dw}x{dh} resize_rows={opt}")
    dev_option("resize_rows", -1)


@pytest.mark.parametrize("shape", [(128, 96, 30, 22), (1920, 40, 224, 8), (64, 300, 20, 7), (2048, 12, 200, 5), (16, 64, 4, 3)])
@pytest.mark.parametrize("c", [1, 3, 4])
def test_resize_bilinear_row_streamed_kernel(gpu_stream, dev_option, shape, c):
    """Bilinear downscales with whole-float4 rows and a vertical step >= 1.5 take the row-streamed kernel (two source rows per output
    row through LDS): the oracle's bits, in a batch (block -> (image, row) decode), with the last source row / column among the taps;
    test option resize_rows = 0 routes the same call to the gather kernel."""
    sw, sh, dw, dh = shape
    if sw / dw * c * 4 > 128:
        pytest.skip("tap stride wider than one line: the launcher keeps the gather kernel")
    n = 3
    src = np.stack([img(sw, sh, c, seed=31 * k) for k in range(n)])
    got = resize_gpu(gpu_stream, src, dw, dh, "bilinear", batch=n)
    for k in range(n):
        assert_same_bits(got[k], O.resize(src[k], dw, dh, "bilinear"), f"rows kernel {shape} c{c} image {k}")
    for cols in (dw, 32, 7, 1008, 2003, 1):   # columns per part (each part stages its own source-row segments); + 1000 / 2000: 64- / 128-thread blocks
        dev_option("resize_rows", cols)
        assert_same_bits(resize_gpu(gpu_stream, src, dw, dh, "bilinear", batch=n), got, f"part width option {cols}")
    dev_option("resize_rows", 0)
    assert_same_bits(resize_gpu(gpu_stream, src, dw, dh, "bilinear", batch=n), got, "gather kernel vs row-streamed kernel")


@pytest.mark.parametrize("shape", [(128, 96, 30, 22), (1920, 40, 224, 8), (64, 300, 20, 7), (2048, 12, 200, 5), (16, 64, 4, 3), (128, 96, 100, 90)])
def test_resize_bilinear_normalize_row_streamed_kernel(gpu_stream, dev_option, shape):
    """resize_bilinear_normalize_3c (P/cuda/resize.rs:184-236) takes the row-streamed walk of `resize` with the `(px - mean) * inv_std`
    epilogue where a plain resize of the geometry does (round 6): the oracle's bits in a batch and through a pointer list, part widths
    forced through the test option, resize_rows = 0 (the per-pixel kernel) the same; the last shape (a vertical step below 1.5) never
    leaves the per-pixel kernel."""
    from kornia_rs import _ffi
    from kornia_rs.hip import DeviceBuffer
    sw, sh, dw, dh = shape
    n, mean, std = 3, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    src = np.stack([img(sw, sh, 3, seed=31 * k) for k in range(n)])
    want = np.stack([O.resize_bilinear_normalize(src[k], dw, dh, mean, std) for k in range(n)])
    d_src = dev(gpu_stream, src)
    for opt in (-1, 32, 7, 0):
        dev_option("resize_rows", opt)
        d_dst = out_buf(gpu_stream, n * dh * dw * 3 * 4)
        _ffi.check(_ffi.lib.kh_resize_bilinear_normalize_f32(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, sw, sh, dw, dh, fptr(mean), fptr(std), 0, n,
                                                             sh * sw * 3, dh * dw * 3))
        assert_same_bits(d_dst.to_numpy(np.float32, (n, dh, dw, 3)), want, f"resize + normalize {shape} option {opt}")
    dev_option("resize_rows", -1)
    srcs = [DeviceBuffer.from_numpy(src[k].reshape(-1), gpu_stream) for k in range(n)]
    dsts = [out_buf(gpu_stream, dh * dw * 3 * 4) for _ in range(n)]
    _ffi.check(_ffi.lib.kh_resize_bilinear_normalize_f32_list(gpu_stream.cuda_stream_ptr, _ffi.pointer_array([b.ptr for b in srcs]), _ffi.pointer_array([b.ptr for b in dsts]),
                                                              n, sw, sh, dw, dh, fptr(mean), fptr(std), 0))
    for k in range(n):
        assert_same_bits(dsts[k].to_numpy(np.float32, (dh, dw, 3)), want[k], f"list image {k} {shape}")


@pytest.mark.parametrize("shape", [(128, 96, 30, 22), (1920, 40, 224, 8), (64, 300, 20, 7), (128, 96, 64, 48)])
@pytest.mark.parametrize("mode", ["bilinear", "bicubic"])
def test_resize_mapped_align_corners_through_the_new_kernels(gpu_stream, dev_option, shape, mode):
    """PixelMapping::AlignCorners (src = dst * (src_len - 1) / (dst_len - 1)) through the launcher's shape-specialised paths: the
    row-streamed bilinear kernel takes any (a, b) grid; the exact-2x bicubic kernel must NOT be taken (its grid is half-pixel only)."""
    from kornia_rs import _ffi
    sw, sh, dw, dh = shape
    n, c = 2, 3
    src = np.stack([img(sw, sh, c, seed=31 * k) for k in range(n)])
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, n * dh * dw * c * 4)
    for opt in (-1, 0):
        dev_option("resize_rows", opt)
        _ffi.check(_ffi.lib.kh_resize_mapped_f32(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, sw, sh, dw, dh, c, O.MODE[mode], 1, n, sh * sw * c, dh * dw * c))
        got = d_dst.to_numpy(np.float32, (n, dh, dw, c))
        for k in range(n):
            assert_same_bits(got[k], O.resize_mapped(src[k], dw, dh, mode, "align_corners"), f"align_corners {mode} {shape} option {opt} image {k}")


@pytest.mark.parametrize("shape", [(128, 96, 64, 48), (130, 50, 65, 25), (130, 50, 65, 31), (2, 7, 1, 3), (2, 2, 1, 1), (256, 9, 128, 20), (258, 34, 129, 17), (258, 33, 129, 11)])
@pytest.mark.parametrize("c", [1, 3, 4])
def test_resize_bicubic_exact_half_kernel(gpu_stream, dev_option, shape, c):
    """Bicubic with a horizontal step of exactly 2 takes the wave-shift kernel (a lane loads its own two source pixels per row and
    gets the outer columns from its neighbours): the oracle's bits for widths that fill whole waves, leave partial ones, are a single
    pixel; vertical down- and up-scaling (the row taps stay general); a batch.  resize_rows = 0: the gather kernel."""
    sw, sh, dw, dh = shape
    n = 2
    src = np.stack([img(sw, sh, c, seed=31 * k) for k in range(n)])
    got = resize_gpu(gpu_stream, src, dw, dh, "bicubic", batch=n)
    for k in range(n):
        assert_same_bits(got[k], O.resize(src[k], dw, dh, "bicubic"), f"bicubic half {shape} c{c} image {k}")
    for opt in (1, 0):   # 1: one output row per lane even where both steps are 2; 0: the gather kernel
        dev_option("resize_rows", opt)
        assert_same_bits(resize_gpu(gpu_stream, src, dw, dh, "bicubic", batch=n), got, f"resize_rows = {opt} vs the launcher's choice")


def test_resize_smoke_known_answer(gpu_stream):  # resize/mod.rs:447-490
    src = np.arange(36, dtype=np.float32).reshape(4, 3, 3)
    got = resize_gpu(gpu_stream, src, 2, 3, "bilinear")[0].reshape(-1)
    want = [2.25, 3.25, 4.25, 6.75, 7.75, 8.75, 14.25, 15.25, 16.25, 18.75, 19.75, 20.75, 26.25, 27.25, 28.25,
            30.75, 31.75, 32.75]
    assert np.abs(got - np.array(want, np.float32)).max() < 1e-4


def test_resize_batch_and_baseline_shape(gpu_stream):
    """configs[1] geometry (1920x1080 -> 224x224, bilinear) on a small batch: every image of the
    batched launch equals its single-image oracle result."""
    n = 3
    src = np.stack([img(1920, 1080, 3, seed=31 * k) for k in range(n)])
    got = resize_gpu(gpu_stream, src, 224, 224, "bilinear", batch=n)
    for k in range(n):
        assert_same_bits(got[k], O.resize(src[k], 224, 224), f"frame {k}")


def warp_gpu(gpu_stream, kind, src, m, dw, dh, mode, batch=1):
    h, w, c = src.shape[-3:]
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, batch * dh * dw * c * 4)
    from kornia_rs import _ffi
    rc = getattr(_ffi.lib, f"kh_warp_{kind}_f32")(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, dw, dh, c,
                                                  fptr(m), O.MODE[mode], batch, h * w * c, dh * dw * c)
    if rc != 0:
        return rc
    return d_dst.to_numpy(np.float32, (batch, dh, dw, c))


def rotation(cx, cy, angle, scale):
    from kornia_rs import _ffi
    out = (C.c_float * 6)()
    _ffi.lib.kh_get_rotation_matrix2d(cx, cy, angle, scale, out)
    return list(out)


AFFINES = {
    "identity": [1, 0, 0, 0, 1, 0],
    "hflip": [-1, 0, 128, 0, 1, 0],
    "shift": [1, 0, 7.25, 0, 1, -3.5],
    "general": [0.9, 0.15, 10.0, -0.1, 1.1, -6.0],
}


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", list(AFFINES) + ["rot30", "rot90", "rot180x2"])
def test_warp_affine_matches_oracle(gpu_stream, mode, name):
    w, h = 129, 97
    src = img(w, h, 3)
    m = AFFINES.get(name) or {"rot30": rotation(64.0, 48.0, 30.0, 1.1), "rot90": rotation(48.0, 48.0, 90.0, 1.0),
                              "rot180x2": rotation(64.5, 48.5, 180.0, 2.0)}[name]
    for dw, dh in [(w, h), (80, 120)]:
        got = warp_gpu(gpu_stream, "affine", src, m, dw, dh, mode)[0]
        assert_same_bits(got, O.warp_affine(src, m, dw, dh, mode), f"affine {name} {mode} -> {dw}x{dh}")


@pytest.mark.parametrize("name", ["identity", "rot30", "shear", "rot180x2"])
@pytest.mark.parametrize("c", [1, 3, 4])
def test_warp_affine_two_pixels_per_lane(gpu_stream, dev_option, name, c):
    """Bilinear warp_affine: the one-pixel kernel, the two-pixels-per-lane kernel (forced; the launcher takes it for nearly horizontal
    source runs only) and the launcher's own choice all give the oracle's bits; widths that leave partial 128-pixel tiles; a batch."""
    w, h = 129, 97
    m = {"identity": [1, 0, 0, 0, 1, 0], "shear": [1.0, 0.3, -4.0, 0.02, 0.95, 3.0], "rot30": rotation(64.0, 48.0, 30.0, 1.1),
         "rot180x2": rotation(64.5, 48.5, 180.0, 2.0)}[name]
    n = 3
    src = np.stack([img(w, h, c, seed=31 * k) for k in range(n)])
    for dw, dh in [(w, h), (64, 5), (65, 7), (200, 30)]:
        want = np.stack([O.warp_affine(src[k], m, dw, dh, "bilinear") for k in range(n)])
        for opt in (-1, 1, 2):
            dev_option("warp_f32_px", opt)
            assert_same_bits(warp_gpu(gpu_stream, "affine", src, m, dw, dh, "bilinear", batch=n), want, f"affine {name} c{c} -> {dw}x{dh} option {opt}")


def test_warp_affine_known_answers(gpu_stream):  # warp/affine.rs:471-640
    src = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.float32)[:, :, None]
    got = warp_gpu(gpu_stream, "affine", src, [-1, 0, 3, 0, 1, 0], 4, 2, "nearest")[0]
    assert got.reshape(-1).tolist() == [4, 3, 2, 1, 8, 7, 6, 5]
    src = np.array([[0, 1], [2, 3]], np.float32)[:, :, None]
    got = warp_gpu(gpu_stream, "affine", src, rotation(0.5, 0.5, 90.0, 1.0), 2, 2, "nearest")[0]
    assert got.reshape(-1).tolist() == [1, 3, 0, 2]
    assert rotation(0.5, 0.5, 90.0, 1.0) == [float(v) for v in __import__("test_oracle_geom_filter").rotation_matrix2d((0.5, 0.5), 90.0, 1.0)]


HOMOGRAPHIES = {
    "projective": (129, 97, [1.03, 0.05, -3.0, -0.02, 0.97, 4.0, 2.0 / (97.0 * 129.0), 1.5 / (129.0 * 97.0), 1.0]),
    "affine_equiv": (320, 240, [0.9, 0.15, 10.0, -0.1, 1.1, -6.0, 0.0, 0.0, 1.0]),
    "identity": (65, 33, [1, 0, 0, 0, 1, 0, 0, 0, 1]),
    "strong": (129, 97, [0.7, -0.2, 30.0, 0.25, 0.8, -10.0, 0.002, -0.001, 1.0]),
}


def test_one_channel_bilinear_taps_at_the_image_edges(gpu_stream):
    """One-channel bilinear taps at the image edges (written for a round-6 experiment that fetched a tap row with one 8-byte load — no
    faster, not kept; the cases stay): translations that put taps on the last column / row, half-pixel shifts, the narrowest sources
    (1 and 2 pixels wide / tall), through warp_affine, warp_perspective, remap and the gather resize."""
    for (w, h) in [(1, 1), (2, 1), (1, 2), (2, 2), (3, 5), (65, 9), (130, 4)]:
        src = img(w, h, 1, seed=w + h)
        for (tx, ty) in [(0.0, 0.0), (0.5, 0.5), (-0.25, 0.75), (1.0, 0.0), (-1.5, -0.5), (0.999, 0.001)]:
            m = [1.0, 0.0, tx, 0.0, 1.0, ty]
            assert_same_bits(warp_gpu(gpu_stream, "affine", src, m, w + 2, h + 1, "bilinear")[0], O.warp_affine(src, m, w + 2, h + 1, "bilinear"), f"affine {w}x{h} shift {tx},{ty}")
            hm = [1.0, 0.0, tx, 0.0, 1.0, ty, 0.0, 0.0, 1.0]
            assert_same_bits(warp_gpu(gpu_stream, "perspective", src, hm, w + 2, h + 1, "bilinear")[0], O.warp_perspective(src, hm, w + 2, h + 1, "bilinear"), f"perspective {w}x{h} shift {tx},{ty}")
        dw, dh = w + 3, h + 2
        mx = np.ascontiguousarray(np.tile(np.linspace(-0.5, w - 0.5, dw, dtype=np.float32), (dh, 1)))
        my = np.ascontiguousarray(np.tile(np.linspace(-0.5, h - 0.5, dh, dtype=np.float32)[:, None], (1, dw)))
        d_src, d_mx, d_my, d_dst = dev(gpu_stream, src), dev(gpu_stream, mx), dev(gpu_stream, my), out_buf(gpu_stream, dw * dh * 4)
        call(gpu_stream, "kh_remap_f32", d_src.ptr, d_mx.ptr, d_my.ptr, d_dst.ptr, w, h, dw, dh, 1, O.MODE["bilinear"], 1, 0, 0)
        assert_same_bits(d_dst.to_numpy(np.float32, (dh, dw, 1)), O.remap(src, mx, my, "bilinear"), f"remap {w}x{h}")
        assert_same_bits(resize_gpu(gpu_stream, src, 2 * w + 1, 2 * h + 1, "bilinear")[0], O.resize(src, 2 * w + 1, 2 * h + 1, "bilinear"), f"resize {w}x{h}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", list(HOMOGRAPHIES))
@pytest.mark.parametrize("c", [1, 3])
def test_warp_perspective_matches_oracle(gpu_stream, mode, name, c):
    w, h, m = HOMOGRAPHIES[name]
    src = img(w, h, c)
    got = warp_gpu(gpu_stream, "perspective", src, m, w, h, mode)[0]
    assert_same_bits(got, O.warp_perspective(src, m, w, h, mode), f"perspective {name} {mode}")
    if name == "identity" and mode not in ("bicubic", "lanczos"):
        assert np.array_equal(got, src)


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("case", ["near_affine", "identity", "strong", "shrink", "all_outside", "upscale"])
def test_warp_perspective_two_pixels_per_lane(gpu_stream, dev_option, case, c):
    """Bilinear warp_perspective runs two pixels of a row per lane (128 x 4 tiles): the oracle's bits for mild and strong
    homographies, shrinks and upscales, all-outside tiles, batches, destination widths that leave a partial half-tile; test option
    warp_f32_px = 1 routes the same call to the one-pixel kernel."""
    sw, sh, dw, dh = 132, 70, 150, 61
    m = {"near_affine": [1.03, 0.05, -3.0, -0.02, 0.97, 4.0, 2e-5, 1.5e-5, 1.0], "identity": [1, 0, 0, 0, 1, 0, 0, 0, 1],
         "strong": [0.9, 0.2, 5.0, -0.1, 1.1, -3.0, 2e-3, 1e-3, 1.0], "shrink": [3.0, 0, 0, 0, 3.0, 0, 0, 0, 1.0],
         "all_outside": [1, 0, 1000.0, 0, 1, 0, 0, 0, 1], "upscale": [0.4, 0.01, 1.0, 0.0, 0.45, 2.0, 0, 0, 1.0]}[case]
    for n, (w_, h_) in ((1, (dw, dh)), (5, (64, 9)), (2, (65, 4)), (2, (129, 5))):
        src = np.stack([img(sw, sh, c, seed=31 * k) for k in range(n)])
        want = np.stack([O.warp_perspective(src[k], m, w_, h_, "bilinear") for k in range(n)])
        dev_option("warp_f32_px", 2)   # (the launcher itself takes the two-pixel kernel only for nearly horizontal source runs)
        assert_same_bits(warp_gpu(gpu_stream, "perspective", src, m, w_, h_, "bilinear", batch=n), want, f"two pixels per lane {case} c{c} n{n}")
        dev_option("warp_f32_px", -1)
        assert_same_bits(warp_gpu(gpu_stream, "perspective", src, m, w_, h_, "bilinear", batch=n), want, f"launcher's choice {case} c{c} n{n}")
        dev_option("warp_f32_px", 1)
        assert_same_bits(warp_gpu(gpu_stream, "perspective", src, m, w_, h_, "bilinear", batch=n), want, f"one pixel per lane {case} c{c} n{n}")
        dev_option("warp_f32_px", -1)


def test_warp_perspective_known_and_singular(gpu_stream):  # warp/perspective.rs:497-590
    from kornia_rs import _ffi
    src = np.arange(6, dtype=np.float32).reshape(3, 2, 1)
    got = warp_gpu(gpu_stream, "perspective", src, [-1, 0, 1, 0, 1, 0, 0, 0, 1], 2, 3, "bilinear")[0]
    assert got.reshape(-1).tolist() == [1, 0, 3, 2, 5, 4]
    src = np.arange(16, dtype=np.float32).reshape(4, 4, 1)
    got = warp_gpu(gpu_stream, "perspective", src, [0.5, 0, -0.25, 0, 0.5, -0.25, 0, 0, 1], 2, 2, "bilinear")[0]
    assert got.reshape(-1).tolist() == [2.5, 4.5, 10.5, 12.5]
    rc = warp_gpu(gpu_stream, "perspective", src, [1, 2, 3, 2, 4, 6, 3, 6, 9], 2, 2, "bilinear")
    assert rc == _ffi.KH_ERR_SINGULAR and "singular" in _ffi.last_error()


def test_unsupported_channels_is_an_error_not_a_fallback(gpu_stream):  # P/warp/cuda.rs:369-400
    from kornia_rs import _ffi
    src = img(8, 8, 2)
    rc = warp_gpu(gpu_stream, "perspective", src, [1, 0, 0, 0, 1, 0, 0, 0, 1], 8, 8, "bilinear")
    assert rc == _ffi.KH_ERR_UNSUPPORTED and "2 channels" in _ffi.last_error()


@pytest.mark.parametrize("mode", MODES)
def test_remap_matches_oracle(gpu_stream, mode):
    w, h, c = 129, 97, 3
    src = img(w, h, c)
    rng = np.random.default_rng(3)
    xs, ys = np.meshgrid(np.arange(80, dtype=np.float32), np.arange(60, dtype=np.float32))
    mx = (xs * np.float32(1.6) + rng.uniform(-2, 2, xs.shape).astype(np.float32)).astype(np.float32)
    my = (ys * np.float32(1.6) + rng.uniform(-2, 2, ys.shape).astype(np.float32)).astype(np.float32)
    mx[0, 0], my[1, 1], mx[2, 2], mx[3, 3] = -0.25, 97.0, np.nan, 128.99  # OOB, OOB, NaN, last-pixel band
    d_src, d_mx, d_my = dev(gpu_stream, src), dev(gpu_stream, mx), dev(gpu_stream, my)
    d_dst = out_buf(gpu_stream, 60 * 80 * c * 4)
    call(gpu_stream, "kh_remap_f32", d_src.ptr, d_mx.ptr, d_my.ptr, d_dst.ptr, w, h, 80, 60, c, O.MODE[mode], 1, 0, 0)
    got = d_dst.to_numpy(np.float32, (60, 80, c))
    assert_same_bits(got, O.remap(src, mx, my, mode), f"remap {mode}")
    assert got[0, 0].tolist() == [0, 0, 0] and got[2, 2].tolist() == [0, 0, 0]


def test_remap_identity_is_exact_copy(gpu_stream):  # P/cuda/remap.rs:770-790
    w, h = 65, 33
    src = img(w, h, 3)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    d_src, d_xs, d_ys = dev(gpu_stream, src), dev(gpu_stream, xs), dev(gpu_stream, ys)  # keep alive
    for mode in ("bilinear", "nearest"):
        d_dst = out_buf(gpu_stream, src.nbytes)
        call(gpu_stream, "kh_remap_f32", d_src.ptr, d_xs.ptr, d_ys.ptr, d_dst.ptr, w, h, w, h, 3, O.MODE[mode], 1, 0, 0)
        assert np.array_equal(d_dst.to_numpy(np.float32, src.shape), src)


OAK_D = dict(intr=(577.48583984375, 652.8748779296875, 577.48583984375, 386.1428833007812),
             dist=(1.7547749280929563, 0.0097926277667284, -0.027250492945313457, 2.1092164516448975,
                   0.462927520275116, -0.08215277642011642, -0.00005535508171073161, 0.00003768636770639569))


def test_undistort_maps_and_pipeline(gpu_stream):
    """config[4] composition at small scale: Brown-Conrady maps (f64 on device) -> remap -> warp."""
    w, h = 160, 96
    intr = (300.0, 300.0, 80.0, 48.0)
    dist = OAK_D["dist"]
    d_mx, d_my = out_buf(gpu_stream, w * h * 4), out_buf(gpu_stream, w * h * 4)
    call(gpu_stream, "kh_correction_map_polynomial_f32", d_mx.ptr, d_my.ptr, w, h, (C.c_double * 4)(*intr), (C.c_double * 8)(*dist))
    mx, my = d_mx.to_numpy(np.float32, (h, w)), d_my.to_numpy(np.float32, (h, w))
    wx, wy = O.correction_map(intr, dist, w, h)
    assert_same_bits(mx, wx, "map_x")
    assert_same_bits(my, wy, "map_y")
    src = img(w, h, 3)
    d_src, d_tmp, d_out = dev(gpu_stream, src), out_buf(gpu_stream, src.nbytes), out_buf(gpu_stream, src.nbytes)
    call(gpu_stream, "kh_remap_f32", d_src.ptr, d_mx.ptr, d_my.ptr, d_tmp.ptr, w, h, w, h, 3, 1, 1, 0, 0)
    hm = [1.03, 0.05, -3.0, -0.02, 0.97, 4.0, 2.0 / (h * w), 1.5 / (w * h), 1.0]
    from kornia_rs import _ffi
    _ffi.check(_ffi.lib.kh_warp_perspective_f32(gpu_stream.cuda_stream_ptr, d_tmp.ptr, d_out.ptr, w, h, w, h, 3, fptr(hm), 1, 1, 0, 0))
    want = O.warp_perspective(O.remap(src, wx, wy), hm, w, h)
    assert_same_bits(d_out.to_numpy(np.float32, src.shape), want, "undistort+warp")


def test_host_matrix_helpers(gpu_stream):
    from kornia_rs import _ffi
    out6, out9 = (C.c_float * 6)(), (C.c_float * 9)()
    _ffi.lib.kh_invert_affine_transform(fptr([0.9, 0.15, 10.0, -0.1, 1.1, -6.0]), out6)
    assert np.array_equal(np.array(list(out6), np.float32), O.invert_affine([0.9, 0.15, 10.0, -0.1, 1.1, -6.0]))
    h = [1.02, 0.03, -5.0, -0.01, 0.99, 2.0, 0.00005, 0.00003, 1.0]
    assert _ffi.lib.kh_invert_homography(fptr(h), out9) == 0
    assert np.array_equal(np.array(list(out9), np.float32), O.invert_homography(h))
    assert _ffi.lib.kh_invert_homography(fptr([0] * 9), out9) == _ffi.KH_ERR_SINGULAR
