"""CPU pins for the pyramid / morphology oracle (oracle/ko_pyramid_morph.c): the reference's unit tests
(P/pyramid.rs:842-1100, P/morphology/ops.rs:276-410) and independent numpy / scipy forms."""
import numpy as np
import pytest
from scipy import ndimage

import oracle_ffi as O

K5 = np.array([1, 4, 6, 4, 1], np.float64) / 16.0


def test_pyramid_sizes_and_constants():  # pyramid.rs:845-885 (7x5 -> 4x3 -> 2x2 -> 1x1, constant stays constant)
    img = np.ones((7, 5, 1), np.float32)
    sizes = [img.shape[:2]]
    for _ in range(3):
        img = O.pyrdown(img)
        sizes.append(img.shape[:2])
        assert np.abs(img - 1.0).max() < 1e-6
    assert sizes == [(7, 5), (4, 3), (2, 2), (1, 1)]
    assert np.array_equal(O.pyrup(np.full((3, 4, 3), 0.5, np.float32)), np.full((6, 8, 3), 0.5, np.float32))
    assert np.array_equal(O.pyrdown(np.full((9, 7, 3), 200, np.uint8)), np.full((5, 4, 3), 200, np.uint8))
    assert np.array_equal(O.pyrup(np.full((3, 4, 3), 200, np.uint8)), np.full((6, 8, 3), 200, np.uint8))


@pytest.mark.parametrize("shape", [(37, 53, 1), (16, 16, 3), (5, 2, 4), (1, 9, 1), (9, 1, 3), (1, 1, 1)])
def test_pyrdown_against_scipy(shape):
    rng = np.random.default_rng(1)
    f = rng.random(shape).astype(np.float32)
    want = ndimage.correlate1d(ndimage.correlate1d(f.astype(np.float64), K5, axis=0, mode="mirror"), K5, axis=1, mode="mirror")[::2, ::2]
    assert np.abs(O.pyrdown(f) - want).max() < 1e-6
    u = rng.integers(0, 256, shape, dtype=np.uint8)
    h = ndimage.correlate1d(u.astype(np.int64), [1, 4, 6, 4, 1], axis=1, mode="mirror")[:, ::2]
    v = ndimage.correlate1d(h, [1, 4, 6, 4, 1], axis=0, mode="mirror")[::2]
    assert np.array_equal(O.pyrdown(u), np.minimum((v + 128) >> 8, 255).astype(np.uint8))


def pyrup_axis_u8(a, axis):
    a = np.moveaxis(a.astype(np.int64), axis, 0)
    n = a.shape[0]
    idx = np.arange(n)
    refl = lambda i: np.abs(i) if n == 1 else np.where(np.abs(i) % (2 * (n - 1)) >= n, 2 * (n - 1) - np.abs(i) % (2 * (n - 1)), np.abs(i) % (2 * (n - 1)))
    prev, nxt = a[refl(idx - 1) if n > 1 else idx * 0], a[refl(idx + 1) if n > 1 else idx * 0]
    out = np.empty((2 * n,) + a.shape[1:], np.int64)
    out[0::2] = (prev + 6 * a + nxt + 4) >> 3
    out[1::2] = (a + nxt + 1) >> 1
    return np.moveaxis(out, 0, axis).astype(np.uint8)


@pytest.mark.parametrize("shape", [(13, 17, 1), (8, 8, 3), (2, 5, 4), (1, 6, 1), (6, 1, 3), (1, 1, 1)])
def test_pyrup_u8_against_numpy(shape):
    u = np.random.default_rng(2).integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(O.pyrup(u), pyrup_axis_u8(pyrup_axis_u8(u, 1), 0))


def pyrup_axis_f32(a, axis):
    a = np.moveaxis(a, axis, 0)
    n = a.shape[0]
    out = np.empty((2 * n,) + a.shape[1:], np.float32)
    if n == 1:
        out[0] = out[1] = a[0]
    else:
        f = np.float32
        out[0] = (f(6) * a[0] + f(2) * a[1]) * f(0.125)
        out[1] = (a[0] + a[1]) * f(0.5)
        out[2:-2:2] = (a[:-2] + f(6) * a[1:-1] + a[2:]) * f(0.125)
        out[3:-2:2] = (a[1:-1] + a[2:]) * f(0.5)
        out[-2] = (a[-2] + f(7) * a[-1]) * f(0.125)
        out[-1] = a[-1]
    return np.moveaxis(out, 0, axis)


@pytest.mark.parametrize("shape", [(13, 17, 1), (8, 8, 3), (2, 5, 4), (1, 6, 1), (6, 1, 3), (1, 1, 1), (3, 3, 1)])
def test_pyrup_f32_against_numpy(shape):
    f = np.random.default_rng(3).random(shape).astype(np.float32)
    assert np.array_equal(O.pyrup(f), pyrup_axis_f32(pyrup_axis_f32(f, 1), 0))  # same f32 expressions, bit for bit


def test_morph_kernels():  # ops.rs:280-325, kernels.rs:113-185
    assert O.morph_kernel("box", 3).all()
    cross = O.morph_kernel("cross", 3)
    assert cross.reshape(-1).tolist() == [0, 1, 0, 1, 1, 1, 0, 1, 0]
    ell = O.morph_kernel("ellipse", 5, 5)
    assert ell[2, 2] == 1 and ell.shape == (5, 5) and ell[0, 0] == 0
    assert O.morph_kernel("ellipse", 7, 3).shape == (3, 7)


def test_morphology_reference_unit_tests():  # ops.rs:326-400
    box3 = O.morph_kernel("box", 3)
    src = np.zeros((3, 3), np.uint8); src[1, 1] = 255
    assert (O.morphology_u8(src, "dilate", box3) == 255).all()
    full = np.full((3, 3), 255, np.uint8)
    er = O.morphology_u8(full, "erode", box3, "constant", [0]).reshape(3, 3)
    assert er[1, 1] == 255 and er[0, 0] == 0
    noise = np.zeros((5, 5), np.uint8); noise[1, 1] = 255
    opened = O.morphology_u8(O.morphology_u8(noise, "erode", box3), "dilate", box3)
    assert (opened == 0).all()
    hole = np.zeros((5, 5), np.uint8); hole[1:4, 1:4] = 255; hole[2, 2] = 0
    closed = O.morphology_u8(O.morphology_u8(hole, "dilate", box3), "erode", box3).reshape(5, 5)
    assert closed[2, 2] == 255 and closed[1, 1] == 255 and closed[3, 3] == 255


@pytest.mark.parametrize("border,mode", [("replicate", "nearest"), ("reflect101", "mirror"), ("reflect", "reflect"), ("wrap", "wrap"), ("constant", "constant")])
@pytest.mark.parametrize("kshape", [("box", 3, 3), ("cross", 5, 5), ("ellipse", 7, 5), ("box", 4, 4), ("ellipse", 2, 6)])
def test_morphology_against_scipy(border, mode, kshape):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (19, 23, 3), dtype=np.uint8)
    mask = O.morph_kernel(*kshape)
    kh, kw = mask.shape
    # scipy centres an even footprint at size//2 as well once the origin is left at 0 and the footprint is
    # applied as a correlation; the reference's tap (ky, kx) reads (y + ky - kh//2, x + kx - kw//2)
    for op, fn, cval in (("dilate", ndimage.maximum_filter, 7), ("erode", ndimage.minimum_filter, 200)):
        want = np.stack([fn(img[:, :, c], footprint=mask.astype(bool), mode=mode, cval=cval,
                            origin=(-(1 - kh % 2) * 0, -(1 - kw % 2) * 0)) for c in range(3)], axis=2)
        got = O.morphology_u8(img, op, mask, border, [cval] * 3)
        if kh % 2 and kw % 2:
            assert np.array_equal(got, want), (op, border, kshape)
        else:  # even footprints: compare against an explicit padded-window evaluation instead
            pad = np.pad(img, ((kh // 2, kh // 2), (kw // 2, kw // 2), (0, 0)),
                         mode={"replicate": "edge", "reflect101": "reflect", "reflect": "symmetric", "wrap": "wrap", "constant": "constant"}[border],
                         **({"constant_values": cval} if border == "constant" else {}))
            wins = [pad[ky:ky + 19, kx:kx + 23] for ky in range(kh) for kx in range(kw) if mask[ky, kx]]
            ref = np.max(wins, axis=0) if op == "dilate" else np.min(wins, axis=0)
            assert np.array_equal(got, ref), (op, border, kshape)


def test_pyramid_reference_known_answers():
    """P/pyramid.rs:885-1290 — the reference's own vectors ("verified with opencv")."""
    ramp = np.arange(16, dtype=np.float32).reshape(4, 4, 1)
    assert np.abs(O.pyrdown(ramp).reshape(-1) - [3.75, 4.875, 8.25, 9.375]).max() < 1e-4            # test_pyrdown
    ramp3 = np.arange(48, dtype=np.float32).reshape(4, 4, 3)
    want = [11.25, 12.25, 13.25, 14.625, 15.625, 16.625, 24.75, 25.75, 26.75, 28.125, 29.125, 30.125]
    assert np.abs(O.pyrdown(ramp3).reshape(-1) - want).max() < 1e-4                                  # test_pyrdown_3c
    up = O.pyrup(np.array([[0.0, 1.0], [2.0, 3.0]], np.float32))                                     # test_pyrup
    assert up.shape == (4, 4, 1) and np.isfinite(up).all()
    odd = O.pyrdown(np.arange(35, dtype=np.float32).reshape(7, 5, 1))                                # test_pyrdown_odd_dims
    assert odd.shape == (4, 3, 1) and np.isfinite(odd).all()
    for img in (np.array([[42.0]], np.float32), np.array([[1.0], [2.0]], np.float32), np.array([[1.0, 2.0]], np.float32)):
        out = O.pyrdown(img)                                                                         # test_pyrdown_min_sizes
        assert out.shape == (1, 1, 1) and np.isfinite(out).all()
    assert O.pyrdown(np.array([[42.0]], np.float32))[0, 0, 0] == 42.0
    big = O.pyrdown(np.full((4, 4, 1), 1e9, np.float32))                                             # test_pyrdown_numeric_extremes
    assert np.isfinite(big).all() and (np.abs(big) <= 1e9).all()
    small = O.pyrdown(np.arange(16, dtype=np.uint8).reshape(4, 4, 1))                                # test_pyrdown_u8_smoke
    assert small.shape == (2, 2, 1) and (small < 255).all()
    assert np.array_equal(small.reshape(-1), np.floor(np.array([3.75, 4.875, 8.25, 9.375]) + 0.5).astype(np.uint8))
    up8 = O.pyrup(np.array([[0, 10], [20, 30]], np.uint8))                                           # test_pyrup_u8_smoke
    assert up8.shape == (4, 4, 1) and up8.min() >= 0 and up8.max() <= 30
    for val in (0, 1, 127, 200, 255):                                                                # test_pyrdown_u8_flat
        assert np.array_equal(O.pyrdown(np.full((16, 16, 1), val, np.uint8)), np.full((8, 8, 1), val, np.uint8))
