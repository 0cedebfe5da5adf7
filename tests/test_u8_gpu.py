"""GPU parity for the u8 fixed-point twins: Q8 blur / binomial, Q10 remap, warp_affine_u8,
warp_perspective_u8 — byte-exact against the CPU oracle.  Shapes and matrices follow the reference's
device==host tests (P/filter/cuda.rs:282-330, P/warp/cuda.rs:174-300, P/interpolation/remap.rs:833-870)."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits, dev, fptr, out_buf
from test_oracle_u8 import hash_image

pytestmark = pytest.mark.gpu


def pat(w, h, c, seed=0):
    return np.roll(O.pattern_u8(w * h * c + seed), -seed)[: w * h * c].reshape(h, w, c).copy()


def blur_gpu(gpu_stream, kind, src, ksize, sigma=None, batch=1):
    from kornia_rs import _ffi
    h, w, c = src.shape[-3:]
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, batch * h * w * c)
    if kind == "gaussian":
        rc = _ffi.lib.kh_gaussian_blur_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, c, ksize[0], ksize[1],
                                          sigma[0], sigma[1], batch, h * w * c, h * w * c)
    else:
        rc = _ffi.lib.kh_box_blur_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, c, ksize[0], ksize[1], batch,
                                     h * w * c, h * w * c)
    if rc != 0:
        return rc
    return d_dst.to_numpy(np.uint8, (batch, h, w, c))


GAUSS = [((3, 3), (1.0, 1.0)), ((3, 3), (0.5, 0.5)), ((5, 5), (1.0, 1.0)), ((7, 7), (2.0, 2.0)), ((7, 7), (1.5, 1.5)),
         ((3, 7), (1.0, 2.0)), ((9, 5), (2.5, 1.2)), ((15, 15), (3.0, 3.0)), ((13, 11), (2.0, 4.0)), ((0, 0), (1.1, 0.0)),
         ((17, 17), (3.0, 3.0)), ((5, 21), (1.0, 4.0))]


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("ksize,sigma", GAUSS)
def test_gaussian_blur_u8_matches_oracle(gpu_stream, c, ksize, sigma):
    for w, h in [(83, 37), (300, 41)]:
        src = hash_image(h, w, c)
        got = blur_gpu(gpu_stream, "gaussian", src, ksize, sigma)[0]
        assert_same_bits(got, O.gaussian_blur_u8(src, ksize, sigma)[0], f"gaussian_u8 {ksize} {sigma} c{c} {w}x{h}")


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("ksize", [(3, 3), (5, 3), (1, 1), (7, 7), (15, 9), (31, 5)])
def test_box_blur_u8_matches_oracle(gpu_stream, c, ksize):
    src = hash_image(45, 131, c)
    got = blur_gpu(gpu_stream, "box", src, ksize)[0]
    assert_same_bits(got, O.box_blur_u8(src, ksize), f"box_u8 {ksize} c{c}")


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("ksize,sigma", [((17, 17), (3.0, 3.0)), ((21, 21), (3.5, 3.5)), ((31, 31), (5.0, 5.0)), ((5, 21), (1.0, 4.0)), ((31, 5), (6.0, 1.0)), ((63, 63), (10.0, 10.0))])
def test_blur_u8_wide_kernels_four_bytes_per_thread(gpu_stream, dev_option, c, ksize, sigma):
    """Kernels beyond the rolling kernel's 15 taps: the two-pass fallback with four bytes per thread (one dword load per tap, byte pairs
    in 16-bit lanes; round 6), its per-byte multiply-add form (u8_blur_swar = 0) and the one-byte-per-thread kernel (2) give the
    oracle's bytes — images narrower than the kernel (every thread on the replicated border), rows that are / are not whole dwords, a
    batch; the box blur of the same size (taps that do not sum to 256 exactly)."""
    for w, h in [(83, 37), (300, 41), (12, 9), (128, 70), (5, 40)]:
        src = hash_image(h, w, c)
        want = O.gaussian_blur_u8(src, ksize, sigma)[0]
        for opt in (-1, 0, 2, 3):   # 3: the vertical pass with one row per thread (the launcher's choice: eight)
            dev_option("u8_blur_swar", opt)
            assert_same_bits(blur_gpu(gpu_stream, "gaussian", src, ksize, sigma)[0], want, f"gaussian_u8 {ksize} c{c} {w}x{h} option {opt}")
    dev_option("u8_blur_swar", -1)
    both = np.stack([hash_image(33, 64, c), hash_image(33, 64, c)[::-1].copy()])
    got = blur_gpu(gpu_stream, "gaussian", both, ksize, sigma, batch=2)
    for k in range(2):
        assert_same_bits(got[k], O.gaussian_blur_u8(both[k], ksize, sigma)[0], f"batch image {k}")
    if ksize[0] <= 31:
        src = hash_image(45, 132, c)
        assert_same_bits(blur_gpu(gpu_stream, "box", src, ksize)[0], O.box_blur_u8(src, ksize), f"box_u8 {ksize} c{c}")


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 5, 1), (5, 1, 1), (1, 1, 3), (2, 1, 3), (3, 2, 1), (4, 1, 1), (257, 3, 1),
                                   (1024, 2, 1), (1025, 2, 3), (342, 5, 3), (85, 400, 3), (64, 33, 4), (1, 700, 4)])
def test_blur_u8_edge_shapes(gpu_stream, shape):
    """1-pixel axes, rows shorter than a dword (fallback kernel), widths around the 256/1024-byte
    wave/tile boundaries, strips taller than the image."""
    w, h, c = shape
    src = pat(w, h, c, seed=7)
    for ksize, sigma in [((3, 3), (1.0, 1.0)), ((7, 7), (2.0, 2.0)), ((15, 3), (3.0, 0.5))]:
        got = blur_gpu(gpu_stream, "gaussian", src, ksize, sigma)[0]
        assert_same_bits(got, O.gaussian_blur_u8(src, ksize, sigma)[0], f"{shape} {ksize}")


@pytest.mark.parametrize("k", [3, 5, 7, 9])
def test_blur_u8_rgb_wave_and_block_seams(gpu_stream, k):
    """The planar RGB kernel: a wave owns 256 pixels (all 64 lanes store), a block 1024; the quads either side of a wave come from
    its halo load, clamped and re-indexed at the image's edges.  Widths either side of every seam, the narrowest rows the kernel
    takes (4 pixels), a last quad of 1 / 2 / 3 pixels, fewer rows than taps."""
    for w, h in [(4, 5), (5, 1), (6, 2), (7, 11), (252, 3), (253, 4), (255, 2), (256, 9), (257, 3), (259, 5), (260, 4), (261, 2), (511, 3), (513, 6),
                 (1023, 2), (1024, 5), (1025, 3), (1027, 4), (1028, 2), (1029, 7), (1281, 3), (2050, 2), (130, 300)]:
        src = pat(w, h, 3, seed=w * 7 + h)
        got = blur_gpu(gpu_stream, "gaussian", src, (k, k), (0.3 * k, 0.2 * k + 0.5))[0]
        assert_same_bits(got, O.gaussian_blur_u8(src, (k, k), (0.3 * k, 0.2 * k + 0.5))[0], f"gaussian {k} {w}x{h}")
    src = pat(1030, 40, 3, seed=5)
    assert_same_bits(blur_gpu(gpu_stream, "box", src, (k, k))[0], O.box_blur_u8(src, (k, k)), f"box {k}")


def test_blur_u8_binomial_planar_kernel(gpu_stream, dev_option):
    """The 3 x 3 binomial (a 3-tap gaussian with sigma in [0.6, 1.2], the default sigma included) on RGB: the planar kernel's rounding
    halving adds on four pixels per dword (round 6) against the interleaved kernel (test option u8_blur_rgb = 0) and the oracle, on the
    widths either side of every wave / block seam, rows fewer than taps, a batch; sigmas just outside the band stay Q8 gaussians."""
    for w, h in [(4, 5), (5, 1), (6, 2), (7, 11), (253, 4), (256, 9), (257, 3), (260, 4), (511, 3), (1023, 2), (1024, 5), (1025, 3), (1029, 7), (2050, 2), (130, 300)]:
        src = pat(w, h, 3, seed=w * 3 + h)
        for sig in ((1.0, 1.0), (0.6, 1.2), (0.8, 0.8), (0.59, 1.0), (1.21, 1.21)):
            want = O.gaussian_blur_u8(src, (3, 3), sig)[0]
            for opt in (-1, 0):
                dev_option("u8_blur_rgb", opt)
                assert_same_bits(blur_gpu(gpu_stream, "gaussian", src, (3, 3), sig)[0], want, f"3x3 sigma {sig} {w}x{h} u8_blur_rgb={opt}")
    dev_option("u8_blur_rgb", -1)
    n = 3
    src = np.stack([pat(1920, 270, 3, seed=31 * k) for k in range(n)])
    got = blur_gpu(gpu_stream, "gaussian", src, (3, 3), (0.8, 0.8), batch=n)
    for k in range(n):
        assert_same_bits(got[k], O.gaussian_blur_u8(src[k], (3, 3), (0.8, 0.8))[0], f"frame {k}")


@pytest.mark.parametrize("k", [3, 5, 7, 9, 11, 13, 15])
def test_blur_u8_gray_rolling_kernel(gpu_stream, dev_option, k):
    """Single-channel images take the rolling gray kernel for 3..15 taps (sixteen pixels per lane, 1024 per wave, 4096 per block; round 6;
    beyond 9 taps with two neighbour dwords and an eight-byte halo on each side),
    widths that are not whole lanes — or destinations off a dword — its RAGGED instantiation: the oracle's bytes for every residue of
    the width mod 16, either side of the lane / wave / block seams, rows fewer than taps, the binomial band and sigmas just outside it,
    box kernels, unequal tap counts, a batch, a destination at an odd address; u8_blur_rgb = 0 keeps the interleaved kernel."""
    from kornia_rs import _ffi
    sizes = [(16 + r, 5) for r in range(0, 16)] + [(1024 + r, 3) for r in (-16, -3, -1, 0, 1, 2, 5, 8, 13, 16)] + [(4096 + r, 3) for r in (-5, 0, 3)] + [(1000, 41), (37, 90), (3840, 9), (64, 1), (48, 2)]
    sigmas = ((0.3 * k, 0.2 * k + 0.5),) if k > 3 else ((1.0, 1.0), (0.6, 1.2), (0.59, 1.3), (0.9, 0.45))
    for w, h in sizes:
        src = pat(w, h, 1, seed=w * 7 + h)
        for sig in sigmas:
            want = O.gaussian_blur_u8(src, (k, k), sig)[0]
            for opt in ((-1, 0) if w in (1000, 37, 1025, 16, 4096) else (-1,)):
                dev_option("u8_blur_rgb", opt)
                assert_same_bits(blur_gpu(gpu_stream, "gaussian", src, (k, k), sig)[0], want, f"gaussian {k} sigma {sig} gray {w}x{h} u8_blur_rgb={opt}")
        dev_option("u8_blur_rgb", -1)
    for w, h in [(1030, 40), (1040, 7)]:
        src = pat(w, h, 1, seed=5)
        assert_same_bits(blur_gpu(gpu_stream, "box", src, (k, k))[0], O.box_blur_u8(src, (k, k)), f"box {k} gray {w}")
        assert_same_bits(blur_gpu(gpu_stream, "gaussian", src, (k, 3), (1.5, 0.8))[0], O.gaussian_blur_u8(src, (k, 3), (1.5, 0.8))[0], f"gaussian {k}x3 gray {w}")
    for (w, h, n) in [(1024, 6, 1), (1001, 7, 3)]:
        src = np.stack([pat(w, h, 1, seed=s_) for s_ in range(n)])
        d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, n * w * h + 8)
        _ffi.check(_ffi.lib.kh_gaussian_blur_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr + 3, w, h, 1, k, k, 1.1, 1.1, n, w * h, w * h))
        got = d_dst.to_numpy(np.uint8, (n * w * h + 8,))
        assert got[:3].tolist() == [255] * 3 and got[3 + n * w * h:3 + n * w * h + 5].tolist() == [255] * 5, "bytes outside the destination were written"
        for i in range(n):
            assert_same_bits(got[3 + i * w * h:3 + (i + 1) * w * h].reshape(h, w, 1), O.gaussian_blur_u8(src[i], (k, k), (1.1, 1.1))[0], f"offset destination {w}x{h} frame {i}")


@pytest.mark.parametrize("k", [3, 5, 7, 9])
def test_blur_u8_rgba_planar_kernel(gpu_stream, dev_option, k):
    """Four-channel images on the planar rolling kernel (round 6: a 16-byte quad per lane, a 4 x 4 byte transpose either side of the RGB
    kernel's per-channel code): the oracle's bytes either side of the wave (256 pixels) / block (1024) seams, the narrowest rows,
    partial last quads, fewer rows than taps, the binomial band, a box, unequal taps, a batch, a destination off a dword;
    u8_blur_rgb = 0 keeps the interleaved kernel."""
    from kornia_rs import _ffi
    sigmas = ((0.3 * k, 0.2 * k + 0.5),) if k > 3 else ((1.0, 1.0), (0.6, 1.2), (0.59, 1.3))
    for w, h in [(4, 5), (5, 1), (6, 2), (7, 11), (253, 4), (255, 2), (256, 9), (257, 3), (260, 4), (1023, 2), (1024, 5), (1025, 3), (1029, 7), (2050, 2), (130, 300), (1000, 9)]:
        src = pat(w, h, 4, seed=w * 7 + h)
        for sig in sigmas:
            want = O.gaussian_blur_u8(src, (k, k), sig)[0]
            for opt in ((-1, 0) if w in (7, 257, 1025, 1000) else (-1,)):
                dev_option("u8_blur_rgb", opt)
                assert_same_bits(blur_gpu(gpu_stream, "gaussian", src, (k, k), sig)[0], want, f"gaussian {k} sigma {sig} rgba {w}x{h} u8_blur_rgb={opt}")
        dev_option("u8_blur_rgb", -1)
    src = pat(1030, 40, 4, seed=5)
    assert_same_bits(blur_gpu(gpu_stream, "box", src, (k, k))[0], O.box_blur_u8(src, (k, k)), f"box {k} rgba")
    assert_same_bits(blur_gpu(gpu_stream, "gaussian", src, (k, 3), (1.5, 0.8))[0], O.gaussian_blur_u8(src, (k, 3), (1.5, 0.8))[0], f"gaussian {k}x3 rgba")
    for (w, h, n) in [(256, 6, 1), (301, 7, 3)]:
        src = np.stack([pat(w, h, 4, seed=s_) for s_ in range(n)])
        d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, n * w * h * 4 + 8)
        _ffi.check(_ffi.lib.kh_gaussian_blur_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr + 3, w, h, 4, k, k, 1.1, 1.1, n, w * h * 4, w * h * 4))
        got = d_dst.to_numpy(np.uint8, (n * w * h * 4 + 8,))
        assert got[:3].tolist() == [255] * 3 and got[3 + n * w * h * 4:3 + n * w * h * 4 + 5].tolist() == [255] * 5, "bytes outside the destination were written"
        for i in range(n):
            assert_same_bits(got[3 + i * w * h * 4:3 + (i + 1) * w * h * 4].reshape(h, w, 4), O.gaussian_blur_u8(src[i], (k, k), (1.1, 1.1))[0], f"offset destination {w}x{h} frame {i}")


def test_blur_u8_batch_4k_strip_and_errors(gpu_stream):
    from kornia_rs import _ffi
    n = 3
    src = np.stack([pat(1920, 1080, 3, seed=31 * k) for k in range(n)])
    got = blur_gpu(gpu_stream, "gaussian", src, (7, 7), (1.5, 1.5), batch=n)
    for k in range(n):
        assert_same_bits(got[k], O.gaussian_blur_u8(src[k], (7, 7), (1.5, 1.5))[0], f"frame {k}")
    assert blur_gpu(gpu_stream, "box", src[0], (4, 3)) == _ffi.KH_ERR_INVALID_ARG
    assert blur_gpu(gpu_stream, "gaussian", src[0], (4, 3), (1.0, 1.0)) == _ffi.KH_ERR_INVALID_ARG
    assert blur_gpu(gpu_stream, "gaussian", pat(8, 8, 2), (3, 3), (1.0, 1.0)) == _ffi.KH_ERR_UNSUPPORTED


def test_quantize_kernel_256_host_helper():
    from kornia_rs import _ffi
    for n, s in [(3, 0.85), (5, 1.0), (7, 2.0), (15, 3.0)]:
        k = (C.c_float * n)()
        _ffi.lib.kh_gaussian_kernel_1d(n, s, k)
        q = (C.c_uint8 * n)()
        _ffi.lib.kh_quantize_kernel_256(k, n, q)
        assert list(q) == O.quantize_kernel_256(np.array(list(k), np.float32)).tolist()


# ---- Q10 gathers ---------------------------------------------------------------------------------------

def warp_u8_gpu(gpu_stream, kind, src, m, dw, dh, batch=1):
    from kornia_rs import _ffi
    h, w, c = src.shape[-3:]
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, batch * dh * dw * c)
    rc = getattr(_ffi.lib, f"kh_warp_{kind}_u8")(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, dw, dh, c, fptr(m),
                                                 batch, h * w * c, dh * dw * c)
    if rc != 0:
        return rc
    return d_dst.to_numpy(np.uint8, (batch, dh, dw, c))


def rotation(cx, cy, angle, scale):
    from kornia_rs import _ffi
    out = (C.c_float * 6)()
    _ffi.lib.kh_get_rotation_matrix2d(cx, cy, angle, scale, out)
    return list(out)


AFFINES = {"flip": [-1.0, 0.0, 128.0, 0.0, 1.0, 0.0], "half": [1.0, 0.0, 0.5, 0.0, 1.0, 0.5], "identity": [1, 0, 0, 0, 1, 0],
           "general": [0.9, 0.15, 10.0, -0.1, 1.1, -6.0], "vflip_shear": [1.0, 0.3, -5.0, 0.0, -1.0, 96.0],
           "degenerate_x": [0.0, 1.0, 3.0, 1.0, 1e-13, 0.0], "zoom": [3.7, 0.0, -100.0, 0.0, 3.7, -80.0]}


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("name", list(AFFINES) + ["rot37", "rot90"])
def test_warp_affine_u8_matches_oracle(gpu_stream, c, name):  # P/warp/cuda.rs:174-222
    m = AFFINES.get(name) or {"rot37": rotation(64.0, 48.0, 37.0, 1.3), "rot90": rotation(64.0, 48.0, 90.0, 1.0)}[name]
    for (w, h), (dw, dh) in [((129, 97), (129, 97)), ((33, 21), (33, 21)), ((129, 97), (80, 150))]:
        src = pat(w, h, c)
        got = warp_u8_gpu(gpu_stream, "affine", src, m, dw, dh)[0]
        assert_same_bits(got, O.warp_affine_u8(src, m, dw, dh), f"affine_u8 {name} c{c} {w}x{h}->{dw}x{dh}")


STAGED_CASES = {  # name -> (forward matrix builder, (w, h), (dw, dh)); sizes span several 64 x 16 tiles and ragged tile edges
    "rot12_wide": (lambda: rotation(210.0, 75.0, 12.0, 0.9), (421, 150), (421, 150)),
    "rot77_tall": (lambda: rotation(60.0, 170.0, 77.0, 1.1), (120, 341), (200, 130)),
    "minify7": (lambda: [1 / 7.0, 0.0, 3.0, 0.0, 1 / 7.0, 2.0], (900, 500), (140, 75)),      # boxes too large for LDS: global fallback
    "magnify5": (lambda: [5.0, 0.3, -20.0, -0.2, 5.0, 7.0], (70, 45), (330, 215)),
    "shear_out": (lambda: [1.0, 0.9, -150.0, 0.05, 1.0, 40.0], (300, 120), (257, 129)),           # tiles fully / partly outside the source
    "tiny_src": (lambda: [1.0, 0.0, 0.4, 0.0, 1.0, 0.3], (3, 5), (70, 20)),                       # source narrower than a quad
    "last_row": (lambda: [1.0, 0.0, 0.0, 0.0, 1.0, -0.5], (67, 33), (67, 40)),                    # taps on the last source row / column
}


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("name", list(STAGED_CASES))
def test_warp_affine_u8_staged_tiles_match_oracle(gpu_stream, c, name):
    """The LDS-staged affine kernel: multi-tile images, ragged edges, boxes that do not fit (block-uniform fallback), quads at the
    end of the last source row — byte for byte the restatement (and therefore the per-pixel kernel)."""
    build, (w, h), (dw, dh) = STAGED_CASES[name]
    m = build()
    src = pat(w, h, c)
    got = warp_u8_gpu(gpu_stream, "affine", src, m, dw, dh)[0]
    assert_same_bits(got, O.warp_affine_u8(src, m, dw, dh), f"staged affine_u8 {name} c{c}")


def test_warp_affine_u8_both_kernels_agree(gpu_stream, dev_option):
    """The test option warp_u8_direct = 1 selects the per-pixel kernel (what images beyond the staged gather's 16-bit box fields
    take): same inputs, the bytes must equal the LDS-staged result."""
    src = pat(421, 150, 3)
    m = rotation(210.0, 75.0, 12.0, 0.9)
    batch = np.stack([src, src[::-1].copy()])
    staged = warp_u8_gpu(gpu_stream, "affine", batch, m, 421, 150, batch=2)
    dev_option("warp_u8_direct", 1)
    direct = warp_u8_gpu(gpu_stream, "affine", batch, m, 421, 150, batch=2)
    assert np.array_equal(direct, staged)
    assert np.array_equal(direct[0], O.warp_affine_u8(src, m, 421, 150))


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("name", list(STAGED_CASES))
def test_warp_affine_u8_whole_box_option(gpu_stream, dev_option, name, c):
    """Production stages only the quads inside each box row's span (affine); the test option warp_u8_spans = 0 keeps the whole box.
    Both are the restatement's bytes."""
    build, (w, h), (dw, dh) = STAGED_CASES[name]
    m = build()
    src = pat(w, h, c)
    want = O.warp_affine_u8(src, m, dw, dh)
    assert_same_bits(warp_u8_gpu(gpu_stream, "affine", src, m, dw, dh)[0], want, f"staged affine_u8 {name} c{c} spans")
    dev_option("warp_u8_spans", 0)
    assert_same_bits(warp_u8_gpu(gpu_stream, "affine", src, m, dw, dh)[0], want, f"staged affine_u8 {name} c{c} whole box")
    dev_option("warp_u8_spans", -1)
    for rows in (8, 16, 32):   # production picks the tile height from the matrix (kh_u8.hip::stage_rows); both, forced
        dev_option("warp_u8_rows", rows)
        assert_same_bits(warp_u8_gpu(gpu_stream, "affine", src, m, dw, dh)[0], want, f"staged affine_u8 {name} c{c} 64 x {rows} tiles")


PROJ = [0.9, 0.12, 4.0, -0.08, 1.05, -2.0, 6.0e-4, -4.5e-4, 1.0]
HOMOGRAPHIES = {"proj": PROJ, "affine_h": [1.1, 0.1, -3.0, -0.05, 0.95, 2.0, 0.0, 0.0, 1.0], "neg": [-v for v in PROJ],
                "flip": [-1, 0, 128, 0, 1, 0, 0, 0, 1], "identity": [1, 0, 0, 0, 1, 0, 0, 0, 1],
                "strong": [0.7, -0.2, 30.0, 0.25, 0.8, -10.0, 0.002, -0.001, 1.0],
                "horizon": [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.02, 0.0, -1.0]}  # denominator changes sign inside rows


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("name", list(HOMOGRAPHIES))
def test_warp_perspective_u8_matches_oracle(gpu_stream, c, name):  # P/warp/cuda.rs:224-300
    m = HOMOGRAPHIES[name]
    for w, h in [(129, 97), (33, 21)]:
        src = pat(w, h, c)
        got = warp_u8_gpu(gpu_stream, "perspective", src, m, w, h)[0]
        assert_same_bits(got, O.warp_perspective_u8(src, m, w, h), f"perspective_u8 {name} c{c} {w}x{h}")


def test_warp_u8_known_answers_batch_and_errors(gpu_stream):
    from kornia_rs import _ffi
    src = np.array([[10, 20, 30, 40], [50, 60, 70, 80]], np.uint8)[:, :, None]
    flip = [-1, 0, 3, 0, 1, 0, 0, 0, 1]
    assert warp_u8_gpu(gpu_stream, "perspective", src, flip, 4, 2)[0].reshape(-1).tolist() == [40, 30, 20, 10, 80, 70, 60, 50]
    assert warp_u8_gpu(gpu_stream, "affine", src, flip[:6], 4, 2)[0].reshape(-1).tolist() == [40, 30, 20, 10, 80, 70, 60, 50]
    rc = warp_u8_gpu(gpu_stream, "perspective", src, [1, 2, 3, 2, 4, 6, 3, 6, 9], 4, 2)
    assert rc == _ffi.KH_ERR_SINGULAR
    assert warp_u8_gpu(gpu_stream, "affine", pat(8, 8, 5), flip[:6], 8, 8) == _ffi.KH_ERR_UNSUPPORTED
    n = 11  # one full group of kStageNB = 8 images + a partial one
    batch = np.stack([pat(640, 360, 3, seed=31 * k) for k in range(n)])
    m = rotation(320.0, 180.0, 12.0, 0.9)
    got = warp_u8_gpu(gpu_stream, "affine", batch, m, 640, 360, batch=n)
    gotp = warp_u8_gpu(gpu_stream, "perspective", batch, PROJ, 640, 360, batch=n)
    for k in range(n):
        assert_same_bits(got[k], O.warp_affine_u8(batch[k], m, 640, 360), f"affine frame {k}")
        assert_same_bits(gotp[k], O.warp_perspective_u8(batch[k], PROJ, 640, 360), f"perspective frame {k}")


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("mode", ["nearest", "bilinear"])
def test_remap_u8_matches_oracle(gpu_stream, c, mode):  # P/interpolation/remap.rs:833-870
    from kornia_rs import _ffi
    w, h, dw, dh, n = 129, 97, 80, 60, 6  # 6 images: one full group of 4 + a partial one
    src = np.stack([pat(w, h, c, seed=31 * k) for k in range(n)])
    rng = np.random.default_rng(5)
    xs, ys = np.meshgrid(np.arange(dw, dtype=np.float32), np.arange(dh, dtype=np.float32))
    mx = (xs * np.float32(1.6) + rng.uniform(-2, 2, xs.shape).astype(np.float32)).astype(np.float32)
    my = (ys * np.float32(1.6) + rng.uniform(-2, 2, ys.shape).astype(np.float32)).astype(np.float32)
    mx[0, 0], my[1, 1], mx[2, 2], mx[3, 3], my[4, 4], mx[5, 5] = -0.25, 97.0, np.nan, 128.99, np.inf, 3.0e9
    d_src, d_mx, d_my = dev(gpu_stream, src), dev(gpu_stream, mx), dev(gpu_stream, my)
    d_dst = out_buf(gpu_stream, n * dh * dw * c)
    _ffi.check(_ffi.lib.kh_remap_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_mx.ptr, d_my.ptr, d_dst.ptr, w, h, dw, dh, c,
                                    O.MODE[mode], n, h * w * c, dh * dw * c))
    got = d_dst.to_numpy(np.uint8, (n, dh, dw, c))
    for k in range(n):
        assert_same_bits(got[k], O.remap_u8(src[k], mx, my, mode), f"remap_u8 {mode} c{c} frame {k}")
    rc = _ffi.lib.kh_remap_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_mx.ptr, d_my.ptr, d_dst.ptr, w, h, dw, dh, c,
                              O.MODE["bicubic"], 1, 0, 0)
    assert rc == _ffi.KH_ERR_UNSUPPORTED


def test_remap_u8_known_answers_and_identity(gpu_stream):  # remap.rs:552-672
    from kornia_rs import _ffi

    def run(src, mx, my, mode):
        h, w, c = src.shape
        d_src, d_mx, d_my = dev(gpu_stream, src), dev(gpu_stream, mx), dev(gpu_stream, my)
        d_dst = out_buf(gpu_stream, mx.size * c)
        _ffi.check(_ffi.lib.kh_remap_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_mx.ptr, d_my.ptr, d_dst.ptr, w, h,
                                        mx.shape[1], mx.shape[0], c, O.MODE[mode], 1, 0, 0))
        return d_dst.to_numpy(np.uint8, mx.shape + (c,))

    two = np.array([[0, 255]], np.uint8)[:, :, None]
    assert run(two, np.array([[0.1]], np.float32), np.array([[0.0]], np.float32), "bilinear").reshape(-1).tolist() == [25]
    sq = np.array([[10, 20], [30, 40]], np.uint8)[:, :, None]
    mx = np.array([[0.49, 1.49], [-1.0, 0.5]], np.float32)
    my = np.array([[0.49, 0.49], [0.5, 2.0]], np.float32)
    assert run(sq, mx, my, "nearest").reshape(-1).tolist() == [10, 20, 0, 0]
    img = pat(65, 33, 3)
    xs, ys = np.meshgrid(np.arange(65, dtype=np.float32), np.arange(33, dtype=np.float32))
    for mode in ("bilinear", "nearest"):
        assert np.array_equal(run(img, xs, ys, mode), img)


# ---- the staged gather (round 3) on the other two operators: multi-tile images, partial image groups, tiles whose box does not fit ----
@pytest.mark.parametrize("rows", [-1, 8, 16, 32])   # -1: the launcher's choice (kh_u8.hip::stage_rows); 64 x 16 / 64 x 32 tiles forced
@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("name", ["proj", "strong", "horizon", "neg"])
def test_warp_perspective_u8_staged_tiles_match_oracle(gpu_stream, dev_option, c, name, rows):
    m = HOMOGRAPHIES[name]
    dev_option("warp_u8_rows", rows)
    for (w, h), (dw, dh), n in [((421, 150), (421, 150), 9), ((900, 500), (140, 75), 2), ((70, 45), (330, 215), 1)]:
        src = np.stack([pat(w, h, c, seed=31 * k) for k in range(n)])
        got = warp_u8_gpu(gpu_stream, "perspective", src, m, dw, dh, batch=n)
        for k in range(n):
            assert_same_bits(got[k], O.warp_perspective_u8(src[k], m, dw, dh), f"staged perspective_u8 {name} c{c} {w}x{h}->{dw}x{dh} frame {k}")


@pytest.mark.parametrize("rows", [-1, 8, 32])   # production: 64 x 16 tiles; 64 x 32 forced
@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("kind", ["smooth", "magnify", "minify", "wild", "all_outside"])
def test_remap_u8_staged_tiles_match_oracle(gpu_stream, dev_option, c, kind, rows):
    """Bilinear remap_u8 through the staged gather: smooth maps (boxes fit), magnification, strong minification and random maps (boxes do
    not fit: block-uniform global fallback), maps that leave the image everywhere (zero tiles); 5 images = one full group + 1."""
    from kornia_rs import _ffi
    dev_option("warp_u8_rows", rows)
    w, h, dw, dh, n = 300, 170, 257, 131, 10   # 10 images = one full group of kStageNB = 8 + a partial one
    src = np.stack([pat(w, h, c, seed=31 * k) for k in range(n)])
    rng = np.random.default_rng(11)
    xs, ys = np.meshgrid(np.arange(dw, dtype=np.float32), np.arange(dh, dtype=np.float32))
    if kind == "smooth":
        mx = xs * np.float32(1.1) + np.float32(4.0) * np.sin(ys / np.float32(17.0)) + np.float32(2.5)
        my = ys * np.float32(1.2) + np.float32(3.0) * np.cos(xs / np.float32(23.0)) - np.float32(1.25)
    elif kind == "magnify":
        mx, my = xs * np.float32(0.21) + np.float32(240.3), ys * np.float32(0.19) + np.float32(140.7)   # reaches the last column / row
    elif kind == "minify":
        mx, my = xs * np.float32(9.0) - np.float32(900.0), ys * np.float32(7.0) - np.float32(300.0)
    elif kind == "wild":
        mx, my = rng.uniform(-20, w + 20, xs.shape), rng.uniform(-20, h + 20, ys.shape)
    else:
        mx, my = xs + np.float32(1000.0), ys - np.float32(1000.0)
    mx, my = np.ascontiguousarray(mx, np.float32), np.ascontiguousarray(my, np.float32)
    d_src, d_mx, d_my = dev(gpu_stream, src), dev(gpu_stream, mx), dev(gpu_stream, my)
    d_dst = out_buf(gpu_stream, n * dh * dw * c)
    _ffi.check(_ffi.lib.kh_remap_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_mx.ptr, d_my.ptr, d_dst.ptr, w, h, dw, dh, c,
                                    O.MODE["bilinear"], n, h * w * c, dh * dw * c))
    got = d_dst.to_numpy(np.uint8, (n, dh, dw, c))
    for k in range(n):
        assert_same_bits(got[k], O.remap_u8(src[k], mx, my, "bilinear"), f"staged remap_u8 {kind} c{c} frame {k}")


@pytest.mark.parametrize("rows", [8, 16, 32])
@pytest.mark.parametrize("kind", ["affine", "perspective"])
def test_staged_gather_sixteen_images_per_block(gpu_stream, dev_option, kind, rows):
    """Batches of 128 and more put 16 consecutive images in a block (8 below that).  130 images = eight full groups + a group of
    two; a mild rotation (boxes of ~3 000 pixels, two staging rounds), 30 degrees (~5 200, three) and 45 degrees at magnification
    1 / 0.6 (~12 800: no staging, the block-uniform global fallback)."""
    dev_option("warp_u8_rows", rows)
    n, w, h, c = 130, 150, 70, 3
    src = np.stack([pat(w, h, c, seed=7 * k + 1) for k in range(n)])
    for ang, scale in [(9.0, 0.95), (30.0, 0.9), (45.0, 0.6)]:
        if kind == "affine":
            m = O.rotation_matrix(w / 2.0, h / 2.0, ang, scale) if hasattr(O, "rotation_matrix") else None
            if m is None:
                a, b = scale * np.cos(np.deg2rad(ang)), scale * np.sin(np.deg2rad(ang))
                m = [float(a), float(b), float((1 - a) * w / 2 - b * h / 2), float(-b), float(a), float(b * w / 2 + (1 - a) * h / 2)]
            want = [O.warp_affine_u8(src[k], m, w, h) for k in (0, 1, 15, 16, 127, 128, 129)]
        else:
            a, b = scale * np.cos(np.deg2rad(ang)), scale * np.sin(np.deg2rad(ang))
            m = [float(a), float(b), float((1 - a) * w / 2 - b * h / 2), float(-b), float(a), float(b * w / 2 + (1 - a) * h / 2), 1e-4, -2e-4, 1.0]
            want = [O.warp_perspective_u8(src[k], m, w, h) for k in (0, 1, 15, 16, 127, 128, 129)]
        got = warp_u8_gpu(gpu_stream, kind, src, m, w, h, batch=n)
        for i, k in enumerate((0, 1, 15, 16, 127, 128, 129)):
            assert_same_bits(got[k], want[i], f"{kind} {ang} deg frame {k} of {n}")
