"""GPU parity for the Gaussian pyramid (f32 / u8) and u8 morphology: bit-exact against the CPU oracle
(shapes after P/cuda/pyramid.rs and P/morphology/cuda.rs device==host tests)."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits, dev, out_buf

pytestmark = pytest.mark.gpu
SHAPES = [(129, 97), (64, 48), (63, 41), (2, 2), (1, 1), (1, 7), (7, 1), (3, 2), (300, 5)]


def make(w, h, c, dtype, seed=0):
    if dtype == np.uint8:
        return np.roll(O.pattern_u8(w * h * c + seed), -seed)[: w * h * c].reshape(h, w, c).copy()
    return np.roll(O.pattern_f32(w * h * c + seed), -seed)[: w * h * c].reshape(h, w, c).copy()


def pyr_gpu(gpu_stream, src, up, batch=1):
    from kornia_rs import _ffi
    h, w, c = src.shape[-3:]
    dh, dw = (2 * h, 2 * w) if up else ((h + 1) // 2, (w + 1) // 2)
    es = src.dtype.itemsize
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, batch * dh * dw * c * es)
    fn = getattr(_ffi.lib, f"kh_{'pyrup' if up else 'pyrdown'}_{'u8' if src.dtype == np.uint8 else 'f32'}")
    _ffi.check(fn(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, c, batch, h * w * c, dh * dw * c))
    return d_dst.to_numpy(src.dtype, (batch, dh, dw, c))


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("up", [False, True])
def test_pyramid_matches_oracle(gpu_stream, dtype, c, up):
    for w, h in SHAPES:
        src = make(w, h, c, dtype)
        got = pyr_gpu(gpu_stream, src, up)[0]
        assert_same_bits(got, O.pyrup(src) if up else O.pyrdown(src), f"{'pyrup' if up else 'pyrdown'} {dtype.__name__} c{c} {w}x{h}")


@pytest.mark.parametrize("c", [1, 3, 4])
def test_pyrdown_f32_wave_seams(gpu_stream, c):
    """Destination widths either side of one / two / four waves (a wave = 64 destination pixels of one row), odd and even source
    widths (the last destination pixel has one source pixel or two), one- to five-row images and a batch."""
    for w, h in [(1, 1), (2, 3), (3, 1), (4, 2), (5, 5), (126, 4), (127, 3), (128, 5), (129, 2), (130, 7), (131, 3), (254, 3), (255, 2), (256, 4), (257, 3), (258, 2),
                 (259, 5), (511, 2), (512, 3), (513, 2), (514, 9), (700, 33)]:
        src = make(w, h, c, np.float32, seed=3 * w + h)
        assert_same_bits(pyr_gpu(gpu_stream, src, False)[0], O.pyrdown(src), f"pyrdown f32 c{c} {w}x{h}")
    batch = np.stack([make(261, 19, c, np.float32, seed=k) for k in range(3)])
    got = pyr_gpu(gpu_stream, batch, False, batch=3)
    for k in range(3):
        assert_same_bits(got[k], O.pyrdown(batch[k]), f"pyrdown f32 batch frame {k}")


@pytest.mark.parametrize("c", [1, 3, 4])
def test_pyrdown_u8_tiled_interior_and_edge_tiles(gpu_stream, c):
    """Sizes that give the tiled pyrdown_u8 kernel (64 x 16 destination pixels per block) interior tiles (dword window loads),
    interior tiles whose last window ends exactly on the image's last bytes, and ragged right / bottom tiles."""
    for w, h in [(520, 140), (260, 65), (259, 65), (261, 66), (262, 67), (265, 66), (267, 65), (513, 33), (390, 130), (2101, 21), (1990, 37), (8, 9), (9, 8), (497, 12), (503, 75), (511, 9), (512, 7), (514, 8), (519, 5), (520, 6), (521, 4), (2047, 5), (2049, 6), (2056, 4), (2057, 3)]:  # + widths around the rolling RGB kernel's 512-source-pixel waves and 2048-pixel blocks
        src = make(w, h, c, np.uint8, seed=w)
        assert_same_bits(pyr_gpu(gpu_stream, src, False)[0], O.pyrdown(src), f"pyrdown u8 c{c} {w}x{h}")
    n = 3
    batch = np.stack([make(300, 70, c, np.uint8, seed=k) for k in range(n)])
    got = pyr_gpu(gpu_stream, batch, False, batch=n)
    for k in range(n):
        assert_same_bits(got[k], O.pyrdown(batch[k]), f"pyrdown u8 batch frame {k}")


def test_pyrdown_u8_gray_rolling_kernel(gpu_stream, dev_option):
    """Single-channel sources whose rows are whole 16-pixel groups take the rolling gray kernel (sixteen source pixels per lane, 1024
    per wave, 4096 per block; round 6): the oracle's bytes on widths either side of those seams, the narrowest rows, one- to
    five-row images, odd heights, strips of a few rows, a batch; pyr_roll = 0 keeps the tile kernel; other widths never leave it."""
    for w, h in [(16, 1), (16, 2), (32, 3), (48, 5), (64, 40), (1008, 5), (1024, 4), (1040, 7), (2032, 3), (2048, 6), (2064, 4), (4080, 3), (4096, 5), (4112, 4),
                 (3840, 31), (128, 401), (100, 9), (1030, 4)]:
        src = make(w, h, 1, np.uint8, seed=w + h)
        want = O.pyrdown(src)
        for opt in (-1, 0):
            dev_option("pyr_roll", opt)
            assert_same_bits(pyr_gpu(gpu_stream, src, False)[0], want, f"pyrdown u8 gray {w}x{h} pyr_roll={opt}")
    dev_option("pyr_roll", -1)
    batch = np.stack([make(1056, 75, 1, np.uint8, seed=k) for k in range(3)])
    got = pyr_gpu(gpu_stream, batch, False, batch=3)
    for k in range(3):
        assert_same_bits(got[k], O.pyrdown(batch[k]), f"pyrdown u8 gray batch frame {k}")


def test_pyrup_u8_gray_rolling_kernel(gpu_stream, dev_option):
    """Single-channel sources whose rows are whole 8-pixel groups take the rolling gray pyrup kernel (eight source pixels per lane, 512
    per wave, 2048 per block; round 6): the oracle's bytes on widths either side of those seams, the narrowest rows, one- to
    four-row images, strips of a few rows, a batch; pyr_roll = 0 keeps the pair kernel; other widths never leave it."""
    for w, h in [(8, 1), (8, 2), (16, 3), (24, 4), (64, 40), (504, 5), (512, 4), (520, 7), (1016, 3), (1024, 6), (1032, 4), (2040, 3), (2048, 5), (2056, 4),
                 (1920, 31), (128, 201), (100, 9), (518, 4)]:
        src = make(w, h, 1, np.uint8, seed=w + h)
        want = O.pyrup(src)
        for opt in (-1, 0):
            dev_option("pyr_roll", opt)
            assert_same_bits(pyr_gpu(gpu_stream, src, True)[0], want, f"pyrup u8 gray {w}x{h} pyr_roll={opt}")
    dev_option("pyr_roll", -1)
    batch = np.stack([make(528, 75, 1, np.uint8, seed=k) for k in range(3)])
    got = pyr_gpu(gpu_stream, batch, True, batch=3)
    for k in range(3):
        assert_same_bits(got[k], O.pyrup(batch[k]), f"pyrup u8 gray batch frame {k}")


def test_pyramid_u8_gray_ragged_widths(gpu_stream, dev_option):
    """Single-channel source widths that are not whole lanes (16 pixels for pyrdown, 8 for pyrup) run on the RAGGED instantiations of the
    rolling gray kernels (round 6): every residue of the width next to a lane, wave and block seam, odd widths (pyrdown's last
    destination pixel has one source pixel), the narrowest rows, destinations off a dword, a batch; pyr_roll = 0 keeps the old kernels."""
    from kornia_rs import _ffi
    down = [(16 + r, 5) for r in range(1, 16)] + [(1024 + r, 4) for r in (-3, -1, 1, 2, 5, 8, 13)] + [(4096 + r, 3) for r in (-5, 3)] + [(1000, 41), (37, 90), (251, 7)]
    up = [(8 + r, 5) for r in range(1, 8)] + [(512 + r, 4) for r in (-3, -1, 1, 2, 5)] + [(2048 + r, 3) for r in (-5, 3)] + [(500, 41), (19, 90), (125, 7)]
    for is_up, sizes in ((False, down), (True, up)):
        for (w, h) in sizes:
            src = make(w, h, 1, np.uint8, seed=w + h)
            want = O.pyrup(src) if is_up else O.pyrdown(src)
            for opt in ((-1, 0) if w in (1000, 37, 500, 19, 1025, 513) else (-1,)):
                dev_option("pyr_roll", opt)
                assert_same_bits(pyr_gpu(gpu_stream, src, is_up)[0], want, f"{'pyrup' if is_up else 'pyrdown'} u8 gray {w}x{h} pyr_roll={opt}")
    dev_option("pyr_roll", -1)
    for is_up, (w, h, n) in ((False, (1024, 6, 1)), (False, (1001, 7, 3)), (True, (512, 6, 1)), (True, (501, 7, 3))):
        src = np.stack([make(w, h, 1, np.uint8, seed=s_) for s_ in range(n)])
        dh, dw = (2 * h, 2 * w) if is_up else ((h + 1) // 2, (w + 1) // 2)
        d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, n * dw * dh + 8)
        fn = _ffi.lib.kh_pyrup_u8 if is_up else _ffi.lib.kh_pyrdown_u8
        _ffi.check(fn(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr + 3, w, h, 1, n, w * h, dw * dh))   # a destination off a dword
        got = d_dst.to_numpy(np.uint8, (n * dw * dh + 8,))
        assert got[:3].tolist() == [255] * 3 and got[3 + n * dw * dh:3 + n * dw * dh + 5].tolist() == [255] * 5, "bytes outside the destination were written"
        for i in range(n):
            assert_same_bits(got[3 + i * dw * dh:3 + (i + 1) * dw * dh].reshape(dh, dw, 1), O.pyrup(src[i]) if is_up else O.pyrdown(src[i]), f"offset destination {'up' if is_up else 'down'} {w}x{h} frame {i}")


def test_pyrdown_u8_rgba_rolling_kernel(gpu_stream, dev_option):
    """Four-channel sources on the rolling planar pyrdown kernel (round 6: two 16-byte loads and one 16-byte store per lane, 4 x 4 byte
    transposes around the RGB kernel's per-channel code): the oracle's bytes on widths either side of the wave (512 source pixels) and
    block (2048) seams, partial last quads, odd widths and heights, the narrowest images, a batch, a destination off a dword;
    pyr_roll = 0 keeps the tile kernel."""
    from kornia_rs import _ffi
    for w, h in [(8, 9), (9, 8), (10, 3), (11, 5), (15, 1), (497, 12), (503, 7), (511, 9), (512, 7), (513, 6), (514, 8), (519, 5), (520, 6), (521, 4), (2047, 5), (2049, 6), (2056, 4), (2057, 3), (300, 131)]:
        src = make(w, h, 4, np.uint8, seed=w + h)
        want = O.pyrdown(src)
        for opt in ((-1, 0) if w in (9, 513, 2057, 300) else (-1,)):
            dev_option("pyr_roll", opt)
            assert_same_bits(pyr_gpu(gpu_stream, src, False)[0], want, f"pyrdown u8 rgba {w}x{h} pyr_roll={opt}")
    dev_option("pyr_roll", -1)
    batch = np.stack([make(301, 70, 4, np.uint8, seed=k) for k in range(3)])
    got = pyr_gpu(gpu_stream, batch, False, batch=3)
    for k in range(3):
        assert_same_bits(got[k], O.pyrdown(batch[k]), f"pyrdown u8 rgba batch frame {k}")
    w, h, n = 301, 7, 2
    src = np.stack([make(w, h, 4, np.uint8, seed=s_) for s_ in range(n)])
    dw, dh = (w + 1) // 2, (h + 1) // 2
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, n * dw * dh * 4 + 8)
    _ffi.check(_ffi.lib.kh_pyrdown_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr + 3, w, h, 4, n, w * h * 4, dw * dh * 4))
    got = d_dst.to_numpy(np.uint8, (n * dw * dh * 4 + 8,))
    assert got[:3].tolist() == [255] * 3 and got[3 + n * dw * dh * 4:3 + n * dw * dh * 4 + 5].tolist() == [255] * 5, "bytes outside the destination were written"
    for i in range(n):
        assert_same_bits(got[3 + i * dw * dh * 4:3 + (i + 1) * dw * dh * 4].reshape(dh, dw, 4), O.pyrdown(src[i]), f"offset destination frame {i}")


def test_pyrup_u8_rgba_rolling_kernel(gpu_stream, dev_option):
    """Four-channel sources on the rolling planar pyrup kernel (round 6: a 16-byte quad per lane, 32 destination bytes per lane and row
    through the wave's LDS transposition): the oracle's bytes on widths either side of the wave (256 source pixels) and block (1024)
    seams, partial last quads, the narrowest images, strips of a few rows, a batch, a destination off a dword; pyr_roll = 0 keeps the
    pair kernel."""
    from kornia_rs import _ffi
    for w, h in [(4, 3), (5, 2), (7, 9), (247, 5), (248, 17), (249, 3), (251, 4), (253, 6), (255, 4), (256, 6), (257, 5), (259, 3), (260, 7), (261, 2), (992, 3), (1003, 8), (1023, 3), (1024, 4),
                 (1025, 5), (1028, 2), (1029, 3), (3, 40), (500, 47)]:
        src = make(w, h, 4, np.uint8, seed=w + h)
        want = O.pyrup(src)
        for opt in ((-1, 0) if w in (5, 257, 1029, 500) else (-1,)):
            dev_option("pyr_roll", opt)
            assert_same_bits(pyr_gpu(gpu_stream, src, True)[0], want, f"pyrup u8 rgba {w}x{h} pyr_roll={opt}")
    dev_option("pyr_roll", -1)
    batch = np.stack([make(301, 70, 4, np.uint8, seed=k) for k in range(3)])
    got = pyr_gpu(gpu_stream, batch, True, batch=3)
    for k in range(3):
        assert_same_bits(got[k], O.pyrup(batch[k]), f"pyrup u8 rgba batch frame {k}")
    w, h, n = 301, 7, 2
    src = np.stack([make(w, h, 4, np.uint8, seed=s_) for s_ in range(n)])
    dw, dh = 2 * w, 2 * h
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, n * dw * dh * 4 + 8)
    _ffi.check(_ffi.lib.kh_pyrup_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr + 3, w, h, 4, n, w * h * 4, dw * dh * 4))
    got = d_dst.to_numpy(np.uint8, (n * dw * dh * 4 + 8,))
    assert got[:3].tolist() == [255] * 3 and got[3 + n * dw * dh * 4:3 + n * dw * dh * 4 + 5].tolist() == [255] * 5, "bytes outside the destination were written"
    for i in range(n):
        assert_same_bits(got[3 + i * dw * dh * 4:3 + (i + 1) * dw * dh * 4].reshape(dh, dw, 4), O.pyrup(src[i]), f"offset destination frame {i}")


def test_pyramid_batch_and_host_api(gpu_stream):
    from kornia_rs import Image, ImageError, imgproc
    n = 3
    for dtype in (np.float32, np.uint8):
        src = np.stack([make(640, 360, 3, dtype, seed=31 * k) for k in range(n)])
        down, up = pyr_gpu(gpu_stream, src, False, batch=n), pyr_gpu(gpu_stream, src, True, batch=n)
        for k in range(n):
            assert_same_bits(down[k], O.pyrdown(src[k]), f"pyrdown frame {k}")
            assert_same_bits(up[k], O.pyrup(src[k]), f"pyrup frame {k}")
    img = Image.from_numpy(make(37, 29, 1, np.float32)).to_hip(gpu_stream)
    levels = imgproc.build_pyramid(img, 3)  # pyramid.rs:845-885
    assert [(l.height, l.width) for l in levels] == [(29, 37), (15, 19), (8, 10), (4, 5)]
    ref = make(37, 29, 1, np.float32)
    for l in levels[1:]:
        ref = O.pyrdown(ref)
        assert_same_bits(l.cpu().numpy(), ref, "build_pyramid level")
    with pytest.raises(ImageError) as e:
        imgproc.pyrup(img, dst=Image.uninit(10, 10, 1, "float32", gpu_stream))
    assert e.value.kind == "InvalidImageSize"


def morph_gpu(gpu_stream, src, op, mask, border, cval, batch=1):
    from kornia_rs import _ffi
    h, w, c = src.shape[-3:]
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, batch * h * w * c)
    m = np.ascontiguousarray(mask, np.uint8)
    cv = (C.c_uint8 * 4)(*([int(v) for v in cval] + [0] * (4 - len(cval))))
    rc = _ffi.lib.kh_morphology_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, c, {"dilate": 0, "erode": 1}[op],
                                   m.ctypes.data_as(C.POINTER(C.c_uint8)), m.shape[1], m.shape[0], O.BORDER[border], cv, batch,
                                   h * w * c, h * w * c)
    if rc != 0:
        return rc
    return d_dst.to_numpy(np.uint8, (batch, h, w, c))


@pytest.mark.parametrize("border", ["constant", "replicate", "reflect101", "reflect", "wrap"])
@pytest.mark.parametrize("kshape", [("box", 3, 3), ("cross", 5, 5), ("ellipse", 7, 5), ("box", 4, 4), ("ellipse", 2, 6), ("box", 15, 15), ("box", 1, 1)])
def test_morphology_matches_oracle(gpu_stream, border, kshape):
    mask = O.morph_kernel(*kshape)
    for (w, h), c in [((129, 97), 3), ((33, 21), 1), ((5, 3), 4), ((1, 1), 1)]:
        src = make(w, h, c, np.uint8, seed=5)
        for op, cval in (("dilate", [7] * c), ("erode", [200] * c)):
            got = morph_gpu(gpu_stream, src, op, mask, border, cval)[0]
            assert_same_bits(got, O.morphology_u8(src, op, mask, border, cval), f"{op} {kshape} {border} {w}x{h} c{c}")


@pytest.mark.parametrize("kshape", [("box", 3, 3), ("box", 5, 5), ("box", 7, 7), ("ellipse", 7, 5), ("box", 2, 2), ("box", 1, 1), ("cross", 3, 9), ("box", 31, 3)])
def test_morphology_tiled_interior_and_ragged_tiles(gpu_stream, kshape):
    """Sizes with tiles that take the tiled kernel's interior staging path (384 flat bytes x 32 rows per tile), a ragged last tile
    column / row, and - 2x2 and 1x1 masks have no right / bottom halo - an interior tile that ends on the image's last byte."""
    mask = O.morph_kernel(*kshape)
    for (w, h), c, border in [((400, 100), 3, "reflect101"), ((1200, 70), 1, "constant"), ((300, 97), 4, "wrap"), ((257, 64), 3, "replicate"),
                              ((768, 96), 1, "reflect")]:
        src = make(w, h, c, np.uint8, seed=11)
        for op, cval in (("dilate", [3] * c), ("erode", [250] * c)):
            got = morph_gpu(gpu_stream, src, op, mask, border, cval)[0]
            assert_same_bits(got, O.morphology_u8(src, op, mask, border, cval), f"{op} {kshape} {border} {w}x{h} c{c}")


@pytest.mark.parametrize("border", ["constant", "replicate", "reflect101", "reflect"])
@pytest.mark.parametrize("k", [3, 5, 7])
def test_morphology_rgb_rolling_wave_boundaries(gpu_stream, k, border):
    """The rolling planar RGB kernel (square boxes of 3 / 5 / 7): a wave owns 256 pixels and a block 1024, the image is cut into
    row strips, edge waves re-index their clamped quads and halo quads — widths either side of every one of those seams, the narrowest
    images the kernel takes (4 .. 7 pixels; narrower ones stay on the tile kernel), heights below the mask's, per-channel
    border values and a batch."""
    mask = O.morph_kernel("box", k, k)
    cval = [9, 130, 251]
    for (w, h) in [(4, 9), (5, 2), (6, 1), (7, 40), (3, 8), (247, 5), (248, 3), (249, 11), (251, 7), (252, 4), (496, 6), (991, 3), (992, 9), (993, 4),
                   (996, 5), (1241, 3), (1988, 2), (131, 400), (253, 6), (255, 3), (256, 8), (257, 5), (259, 4), (260, 3), (261, 9), (512, 4), (1023, 3), (1024, 6),
                   (1025, 4), (1027, 3), (1028, 5), (1029, 2), (1285, 4)]:
        src = make(w, h, 3, np.uint8, seed=w + h)
        for op in ("dilate", "erode"):
            got = morph_gpu(gpu_stream, src, op, mask, border, cval)[0]
            assert_same_bits(got, O.morphology_u8(src, op, mask, border, cval), f"{op} box{k} {border} {w}x{h}")
    src = np.stack([make(1000, 75, 3, np.uint8, seed=s) for s in (4, 5, 6)])
    got = morph_gpu(gpu_stream, src, "erode", mask, border, cval, batch=3)
    for i in range(3):
        assert_same_bits(got[i], O.morphology_u8(src[i], "erode", mask, border, cval), f"batch frame {i}")


@pytest.mark.parametrize("border", ["constant", "replicate", "reflect101", "reflect"])
@pytest.mark.parametrize("k", [9, 11, 13, 15, 21, 31])
def test_morphology_rgb_large_boxes_as_a_chain(gpu_stream, dev_option, k, border):
    """Square RGB boxes of 9 .. 31 run as a chain of rolling-kernel passes (boxes of 7 and one of 3 / 5 / 7 through one scratch image;
    round 6): max / min compose exactly and every border mode commutes with the composition — the oracle's bytes on images smaller
    than the mask, narrower than a wave, with per-channel border values that win (dilate, 251) or lose, and a batch; test option
    morph_roll = 2 keeps the tile kernel."""
    mask = O.morph_kernel("box", k, k)
    cval = [9, 130, 251]
    for (w, h) in [(4, 9), (7, 40), (40, 7), (131, 97), (260, 33), (1029, 12), (64, 200)]:
        src = make(w, h, 3, np.uint8, seed=w + h + k)
        for op in ("dilate", "erode"):
            want = O.morphology_u8(src, op, mask, border, cval)
            for opt in (-1, 2):
                dev_option("morph_roll", opt)
                assert_same_bits(morph_gpu(gpu_stream, src, op, mask, border, cval)[0], want, f"{op} box{k} {border} {w}x{h} morph_roll={opt}")
    dev_option("morph_roll", -1)
    src = np.stack([make(300, 75, 3, np.uint8, seed=s) for s in (4, 5, 6)])
    got = morph_gpu(gpu_stream, src, "dilate", mask, border, cval, batch=3)
    for i in range(3):
        assert_same_bits(got[i], O.morphology_u8(src[i], "dilate", mask, border, cval), f"batch frame {i}")


@pytest.mark.parametrize("border", ["constant", "replicate", "reflect101", "reflect"])
@pytest.mark.parametrize("k", [3, 5, 7, 9, 15, 31])
def test_morphology_gray_rolling_kernel(gpu_stream, dev_option, k, border):
    """Single-channel images whose rows are whole 16-pixel groups take the rolling gray kernel (sixteen pixels per lane, 1024 per wave;
    round 6) for square boxes of 3 / 5 / 7 and the chain of such passes for 9 .. 31: the oracle's bytes on widths either side of the
    wave (1024) and block (4096) seams, the narrowest rows (16 pixels), heights below the mask's, a border value that wins / loses, a
    batch; morph_roll = 2 keeps the tile kernel; widths that are not multiples of 16 never leave it."""
    mask = O.morph_kernel("box", k, k)
    for (w, h) in [(16, 9), (32, 2), (48, 1), (64, 40), (1008, 5), (1024, 4), (1040, 6), (2048, 3), (2064, 4), (4096, 3), (4112, 5), (3840, 12), (128, 400), (100, 9), (1030, 4)]:
        src = make(w, h, 1, np.uint8, seed=w + h + k)
        for op, cval in (("dilate", [250]), ("erode", [9]), ("dilate", [3])):
            want = O.morphology_u8(src, op, mask, border, cval)
            for opt in (-1, 2):
                dev_option("morph_roll", opt)
                assert_same_bits(morph_gpu(gpu_stream, src, op, mask, border, cval)[0], want, f"{op} gray box{k} {border} {w}x{h} cval {cval} morph_roll={opt}")
    dev_option("morph_roll", -1)
    src = np.stack([make(1056, 75, 1, np.uint8, seed=s_) for s_ in (4, 5, 6)])
    got = morph_gpu(gpu_stream, src, "erode", mask, border, [200], batch=3)
    for i in range(3):
        assert_same_bits(got[i], O.morphology_u8(src[i], "erode", mask, border, [200]), f"batch frame {i}")


@pytest.mark.parametrize("border", ["constant", "replicate", "reflect101", "reflect"])
@pytest.mark.parametrize("kshape", [("cross", 3), ("cross", 5), ("cross", 7), ("ellipse", 3), ("ellipse", 5), ("ellipse", 7)])
def test_morphology_cross_and_ellipse_on_the_rolling_kernels(gpu_stream, dev_option, kshape, border):
    """kh_morph_kernel's cross and ellipse of 3 / 5 / 7 are unions of at most two rectangles of the window (the ellipse sits half a
    pixel down / right of the anchor: a 2 x 2 / 4 x 4 block, two overlapping 4 x 6 blocks) and run on the rolling RGB and gray kernels
    (round 6): the oracle's bytes either side of the wave / block seams, on the narrowest images, heights below the mask's, border
    values that win or lose, a batch; morph_roll = 2 keeps the tile kernel's mask scan."""
    shape, k = kshape
    mask = O.morph_kernel(shape, k, k)
    for c, sizes in ((3, [(4, 9), (5, 2), (6, 1), (7, 40), (3, 8), (248, 3), (251, 7), (255, 3), (256, 8), (257, 5), (260, 3), (1023, 3), (1024, 6), (1025, 4),
                          (1029, 2), (131, 200)]),
                     (1, [(16, 9), (32, 2), (48, 1), (64, 40), (1008, 5), (1024, 4), (1040, 6), (4096, 3), (4112, 5), (128, 200), (100, 9)])):
        for (w, h) in sizes:
            src = make(w, h, c, np.uint8, seed=w + h + k)
            for op, cval in (("dilate", [250, 3, 130][:c]), ("erode", [9, 251, 130][:c])):
                want = O.morphology_u8(src, op, mask, border, cval)
                for opt in (-1, 2):
                    dev_option("morph_roll", opt)
                    assert_same_bits(morph_gpu(gpu_stream, src, op, mask, border, cval)[0], want, f"{op} c{c} {shape}{k} {border} {w}x{h} morph_roll={opt}")
    dev_option("morph_roll", -1)
    for c, w in ((3, 1000), (1, 1056)):
        src = np.stack([make(w, 75, c, np.uint8, seed=s_) for s_ in (4, 5, 6)])
        got = morph_gpu(gpu_stream, src, "erode", mask, border, [200] * c, batch=3)
        for i in range(3):
            assert_same_bits(got[i], O.morphology_u8(src[i], "erode", mask, border, [200] * c), f"batch frame {i} c{c}")


@pytest.mark.parametrize("border", ["constant", "replicate", "reflect101", "reflect"])
@pytest.mark.parametrize("kshape", [("box", 3), ("box", 5), ("box", 7), ("cross", 5), ("ellipse", 7), ("box", 9)])
def test_morphology_gray_ragged_widths(gpu_stream, dev_option, kshape, border):
    """Single-channel widths that are not whole 16-pixel lanes run on the RAGGED instantiation of the rolling gray kernel (round 6): the
    lane a row ends in re-indexes its sixteen loaded bytes, stores its first 1 .. 15, and rows off a dword leave through unaligned
    stores.  The oracle's bytes for every residue of the width mod 16 next to a lane, wave and block seam, the narrowest rows, borders
    that win or lose, destinations off a dword, a batch with an odd image stride; morph_roll = 2 keeps the tile kernel."""
    shape, k = kshape
    mask = O.morph_kernel(shape, k, k)
    sizes = [(16 + r, 5) for r in range(1, 16)] + [(1024 + r, 3) for r in (-3, -1, 1, 2, 5, 8, 13)] + [(4096 + r, 3) for r in (-5, 3)] + [(1000, 40), (37, 90), (250, 7)]
    for (w, h) in sizes:
        src = make(w, h, 1, np.uint8, seed=w + h + k)
        for op, cval in (("dilate", [250]), ("erode", [9])):
            want = O.morphology_u8(src, op, mask, border, cval)
            for opt in ((-1, 2) if w in (1000, 37, 1025) else (-1,)):
                dev_option("morph_roll", opt)
                assert_same_bits(morph_gpu(gpu_stream, src, op, mask, border, cval)[0], want, f"{op} gray {shape}{k} {border} {w}x{h} morph_roll={opt}")
    dev_option("morph_roll", -1)
    # a destination off a dword (the aligned 1024-pixel rows then take the unaligned stores too) and a batch whose images are 1001 * 7 bytes apart
    from kornia_rs import _ffi
    for (w, h, n) in [(1024, 6, 1), (1001, 7, 3)]:
        src = np.stack([make(w, h, 1, np.uint8, seed=s_) for s_ in range(n)])
        d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, n * w * h + 8)
        m = np.ascontiguousarray(mask, np.uint8)
        cv = (C.c_uint8 * 4)(77, 0, 0, 0)
        _ffi.check(_ffi.lib.kh_morphology_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr + 3, w, h, 1, 0, m.ctypes.data_as(C.POINTER(C.c_uint8)), k, k,
                                             O.BORDER[border], cv, n, w * h, w * h))
        got = d_dst.to_numpy(np.uint8, (n * w * h + 8,))
        assert got[:3].tolist() == [255] * 3 and got[3 + n * w * h:3 + n * w * h + 5].tolist() == [255] * 5, "bytes outside the destination were written"
        for i in range(n):
            assert_same_bits(got[3 + i * w * h:3 + (i + 1) * w * h].reshape(h, w, 1), O.morphology_u8(src[i], "dilate", mask, border, [77]), f"offset destination {w}x{h} frame {i}")


@pytest.mark.parametrize("border", ["constant", "replicate", "reflect101", "reflect"])
@pytest.mark.parametrize("kshape", [("box", 3), ("box", 5), ("box", 7), ("cross", 5), ("ellipse", 7), ("box", 9), ("box", 15)])
def test_morphology_rgba_rolling_kernel(gpu_stream, dev_option, kshape, border):
    """Four-channel images on the rolling planar kernel (round 6: a 16-byte quad per lane, a 4 x 4 byte transpose either side of the RGB
    kernel's per-channel code): the oracle's bytes either side of the wave (256 pixels) and block (1024) seams, the narrowest images,
    partial last quads, heights below the mask's, per-channel border values that win or lose, the chain for boxes of 9 / 15, a batch, a
    destination off a dword; morph_roll = 2 keeps the tile kernel."""
    from kornia_rs import _ffi
    shape, k = kshape
    mask = O.morph_kernel(shape, k, k)
    cval = [9, 130, 251, 77]
    for (w, h) in [(4, 9), (5, 2), (6, 1), (7, 40), (3, 8), (248, 3), (251, 7), (255, 3), (256, 8), (257, 5), (260, 3), (1023, 3), (1024, 6), (1025, 4), (1029, 2), (131, 200)]:
        src = make(w, h, 4, np.uint8, seed=w + h + k)
        for op in ("dilate", "erode"):
            want = O.morphology_u8(src, op, mask, border, cval)
            for opt in ((-1, 2) if w in (7, 257, 1025, 131) else (-1,)):
                dev_option("morph_roll", opt)
                assert_same_bits(morph_gpu(gpu_stream, src, op, mask, border, cval)[0], want, f"{op} c4 {shape}{k} {border} {w}x{h} morph_roll={opt}")
    dev_option("morph_roll", -1)
    src = np.stack([make(1000, 75, 4, np.uint8, seed=s_) for s_ in (4, 5, 6)])
    got = morph_gpu(gpu_stream, src, "erode", mask, border, cval, batch=3)
    for i in range(3):
        assert_same_bits(got[i], O.morphology_u8(src[i], "erode", mask, border, cval), f"batch frame {i}")
    w, h, n = 301, 7, 2
    src = np.stack([make(w, h, 4, np.uint8, seed=s_) for s_ in range(n)])
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, n * w * h * 4 + 8)
    m = np.ascontiguousarray(mask, np.uint8)
    cv = (C.c_uint8 * 4)(*cval)
    _ffi.check(_ffi.lib.kh_morphology_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr + 3, w, h, 4, 0, m.ctypes.data_as(C.POINTER(C.c_uint8)), k, k,
                                         O.BORDER[border], cv, n, w * h * 4, w * h * 4))
    got = d_dst.to_numpy(np.uint8, (n * w * h * 4 + 8,))
    assert got[:3].tolist() == [255] * 3 and got[3 + n * w * h * 4:3 + n * w * h * 4 + 5].tolist() == [255] * 5, "bytes outside the destination were written"
    for i in range(n):
        assert_same_bits(got[3 + i * w * h * 4:3 + (i + 1) * w * h * 4].reshape(h, w, 4), O.morphology_u8(src[i], "dilate", mask, border, cval), f"offset destination frame {i}")


def test_morphology_both_kernels_agree(gpu_stream, tmp_path):
    """KH_MORPH_DIRECT=1 selects the per-pixel kernel (read once per process): a child process runs it on the same inputs and the
    bytes must equal this process's tiled result."""
    import os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    src = np.stack([make(421, 75, 3, np.uint8, seed=s) for s in (1, 2)])
    ell = O.morph_kernel("ellipse", 7, 5)
    tiled = morph_gpu(gpu_stream, src, "dilate", ell, "reflect101", [0, 0, 0], batch=2)
    np.save(tmp_path / "src.npy", src)
    code = (f"import sys, numpy as np; sys.path[:0] = [{str(root / 'kornia-rs_amd')!r}, {str(root / 'tests')!r}, {str(root / 'oracle')!r}]\n"
            "import conftest, test_pyramid_morph_gpu as T\nfrom kornia_rs import hip\n"
            f"src = np.load({str(tmp_path / 'src.npy')!r}); st = hip.Stream.new(0)\n"
            "out = T.morph_gpu(st, src, 'dilate', T.O.morph_kernel('ellipse', 7, 5), 'reflect101', [0, 0, 0], batch=2)\n"
            f"np.save({str(tmp_path / 'direct.npy')!r}, out)\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KH_MORPH_DIRECT="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(tmp_path / "direct.npy"), tiled)


def test_morphology_unit_tests_batch_and_errors(gpu_stream):  # ops.rs:326-400
    from kornia_rs import Image, _ffi, imgproc
    box3 = O.morph_kernel("box", 3)
    src = np.zeros((3, 3, 1), np.uint8); src[1, 1] = 255
    assert (morph_gpu(gpu_stream, src, "dilate", box3, "constant", [0]) == 255).all()
    er = morph_gpu(gpu_stream, np.full((3, 3, 1), 255, np.uint8), "erode", box3, "constant", [0])[0].reshape(3, 3)
    assert er[1, 1] == 255 and er[0, 0] == 0
    empty = np.zeros((3, 3), np.uint8)  # no active tap: dilate -> 0, erode -> unwrap_or_default = 0
    img = make(20, 10, 3, np.uint8)
    for op in ("dilate", "erode"):
        assert (morph_gpu(gpu_stream, img, op, empty, "replicate", [0, 0, 0]) == 0).all()
    n = 4
    batch = np.stack([make(320, 180, 3, np.uint8, seed=31 * k) for k in range(n)])
    ell = O.morph_kernel("ellipse", 9, 9)
    got = morph_gpu(gpu_stream, batch, "erode", ell, "reflect101", [0, 0, 0], batch=n)
    for k in range(n):
        assert_same_bits(got[k], O.morphology_u8(batch[k], "erode", ell, "reflect101"), f"frame {k}")
    assert morph_gpu(gpu_stream, img, "dilate", np.ones((33, 3), np.uint8), "constant", [0, 0, 0]) == _ffi.KH_ERR_UNSUPPORTED
    # host API: open removes an isolated pixel, close fills a hole (ops.rs:345-400)
    noise = np.zeros((5, 5, 1), np.uint8); noise[1, 1] = 255
    k3 = imgproc.Kernel("box", 3)
    assert np.array_equal(k3.data, box3) and k3.pad() == (1, 1)
    opened = imgproc.morph_open(Image.from_numpy(noise).to_hip(gpu_stream), k3).cpu().numpy()
    assert (opened == 0).all()
    hole = np.zeros((5, 5, 1), np.uint8); hole[1:4, 1:4] = 255; hole[2, 2] = 0
    closed = imgproc.morph_close(Image.from_numpy(hole).to_hip(gpu_stream), k3).cpu().numpy().reshape(5, 5)
    assert closed[2, 2] == 255 and closed[1, 1] == 255 and closed[3, 3] == 255
    assert np.array_equal(imgproc.Kernel("ellipse", (7, 5)).data, O.morph_kernel("ellipse", 7, 5))


@pytest.mark.parametrize("c", [1, 3, 4])
def test_pyrup_u8_rolling_wave_boundaries(gpu_stream, c):
    """Source widths around the rolling RGB kernel's 256-source-pixel waves and 1024-pixel blocks, partial last quads, rows shorter than
    a quad (the pair kernel), strips of a few rows, a batch; the other channel counts take the pair kernel on the same shapes."""
    for w, h in [(4, 3), (5, 2), (7, 9), (247, 5), (248, 17), (249, 33), (251, 4), (253, 6), (992, 3), (993, 5), (1003, 18), (3, 40), (500, 47), (255, 4), (256, 6), (257, 5), (259, 3), (260, 7), (261, 2), (1023, 3), (1024, 4), (1025, 5), (1028, 2), (1029, 3)]:
        src = make(w, h, c, np.uint8, seed=w + h)
        assert_same_bits(pyr_gpu(gpu_stream, src, True)[0], O.pyrup(src), f"pyrup u8 c{c} {w}x{h}")
    n = 3
    batch = np.stack([make(301, 70, c, np.uint8, seed=k) for k in range(n)])
    got = pyr_gpu(gpu_stream, batch, True, batch=n)
    for k in range(n):
        assert_same_bits(got[k], O.pyrup(batch[k]), f"pyrup u8 batch frame {k}")
