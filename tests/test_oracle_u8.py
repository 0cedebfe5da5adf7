"""CPU pins for the u8 fixed-point oracle (oracle/ko_u8.c) against the reference's own tests and an
independent numpy restatement (P/filter/ops.rs:1985-2157, P/interpolation/remap.rs:552-672,
P/warp/perspective.rs:335-360, P/warp/cuda.rs:174-300)."""
import numpy as np
import pytest

import oracle_ffi as O


def hash_image(rows, cols, c=1):  # ops.rs:1991-1994: ((i * 2654435761) >> 24) as u8
    i = np.arange(rows * cols * c, dtype=np.uint64)
    return ((i * np.uint64(2654435761)) >> np.uint64(24)).astype(np.uint8).reshape(rows, cols, c)


def q8_pass_numpy(img, q, axis):
    half = len(q) // 2
    pad = [(0, 0)] * 3
    pad[axis] = (half, half)
    p = np.pad(img.astype(np.uint32), pad, mode="edge")
    acc = np.zeros(img.shape, np.uint32)
    for t, k in enumerate(q):
        sl = [slice(None)] * 3
        sl[axis] = slice(t, t + img.shape[axis])
        acc += p[tuple(sl)] * np.uint32(k)
    return ((acc + 128) >> 8).astype(np.uint8)


def gaussian_taps(n, sigma):
    x = np.arange(n, dtype=np.float32) - np.float32(n // 2)
    g = np.exp(-(x * x) / (np.float32(2.0) * np.float32(sigma) * np.float32(sigma))).astype(np.float32)
    return g / g.sum(dtype=np.float32)


def test_quantize_kernel_256_known_values():  # ops.rs:748-760
    assert O.quantize_kernel_256(np.full(3, 1 / 3, np.float32)).tolist() == [85, 86, 85]
    assert O.quantize_kernel_256(np.full(5, 0.2, np.float32)).tolist() == [51, 51, 52, 51, 51]
    assert O.quantize_kernel_256(np.array([1.0], np.float32)).tolist() == [255]  # 256 saturates, centre clamps
    for n, s in [(3, 0.85), (5, 1.0), (7, 2.0), (7, 1.5), (15, 3.0)]:
        q = O.quantize_kernel_256(gaussian_taps(n, s))
        assert int(q.astype(np.int32).sum()) == 256 and np.array_equal(q, q[::-1])


@pytest.mark.parametrize("c", [1, 3, 4])
def test_q8_separable_matches_numpy_restatement(c):
    img = hash_image(37, 83, c)
    qx, qy = O.quantize_kernel_256(gaussian_taps(7, 2.0)), O.quantize_kernel_256(gaussian_taps(5, 1.0))
    want = q8_pass_numpy(q8_pass_numpy(img, qx, 1), qy, 0)
    assert np.array_equal(O.separable_blur_u8(img, qx, qy), want)


def test_gaussian_blur_u8_5x5_takes_general_path():  # ops.rs:2020-2058
    img = hash_image(37, 83)
    got, path = O.gaussian_blur_u8(img, (5, 5), (1.0, 1.0))
    q = O.quantize_kernel_256(gaussian_taps(5, 1.0))
    assert path == 2 and np.array_equal(got, O.separable_blur_u8(img, q, q))


def test_binomial_path_selection_and_closeness():  # ops.rs:21-27, 2063-2103
    for rows, cols, c in [(37, 83, 1), (17, 45, 3)]:
        img = hash_image(rows, cols, c)
        got, path = O.gaussian_blur_u8(img, (3, 3), (1.0, 1.0))
        assert path == 1 and np.array_equal(got, O.binomial3_u8(img))
        q = O.quantize_kernel_256(gaussian_taps(3, 0.85))
        diff = np.abs(got.astype(np.int16) - O.separable_blur_u8(img, q, q).astype(np.int16)).max()
        assert diff <= 2
        # nested halving adds == ((a + 2b + c) rounded up twice): independent numpy form
        p = np.pad(img.astype(np.uint32), ((0, 0), (1, 1), (0, 0)), mode="edge")
        rh = lambda a, b: (a + b + 1) >> 1
        h = rh(rh(p[:, :-2], p[:, 1:-1]), rh(p[:, 1:-1], p[:, 2:]))
        p = np.pad(h, ((1, 1), (0, 0), (0, 0)), mode="edge")
        v = rh(rh(p[:-2], p[1:-1]), rh(p[1:-1], p[2:]))
        assert np.array_equal(got, v.astype(np.uint8))
    assert O.gaussian_blur_u8(hash_image(9, 9), (3, 3), (1.3, 1.0))[1] == 2  # sigma outside [0.6, 1.2]
    assert O.gaussian_blur_u8(hash_image(9, 9), (3, 5), (1.0, 1.0))[1] == 2


def test_blur_u8_degenerate_shapes_and_errors():  # ops.rs:2107-2157, 66-75
    col = (np.arange(5) * 50).astype(np.uint8).reshape(5, 1, 1)
    assert O.gaussian_blur_u8(col, (3, 3), (1.0, 1.0))[0].shape == (5, 1, 1)
    assert O.gaussian_blur_u8(col.reshape(1, 5, 1), (3, 3), (1.0, 1.0))[0].shape == (1, 5, 1)
    const = np.full((6, 7, 3), 77, np.uint8)
    assert np.array_equal(O.gaussian_blur_u8(const, (7, 7), (1.5, 1.5))[0], const)
    assert np.array_equal(O.box_blur_u8(const, (5, 3)), const)
    with pytest.raises(ValueError):
        O.box_blur_u8(const, (4, 3))
    with pytest.raises(ValueError):
        O.gaussian_blur_u8(const, (4, 3), (1.0, 1.0))


def test_remap_u8_known_answers():  # remap.rs:552-672
    img = O.pattern_u8(9 * 7 * 3).reshape(7, 9, 3)
    xs, ys = np.meshgrid(np.arange(9, dtype=np.float32), np.arange(7, dtype=np.float32))
    for mode in ("bilinear", "nearest"):
        assert np.array_equal(O.remap_u8(img, xs, ys, mode), img)
    two = np.array([[0, 255]], np.uint8)
    assert O.remap_u8(two, np.array([[0.1]], np.float32), np.array([[0.0]], np.float32)).reshape(-1).tolist() == [25]
    sq = np.array([[10, 20], [30, 40]], np.uint8)
    mx = np.array([[0.49, 1.49], [-1.0, 0.5]], np.float32)
    my = np.array([[0.49, 0.49], [0.5, 2.0]], np.float32)
    assert O.remap_u8(sq, mx, my, "nearest").reshape(-1).tolist() == [10, 20, 0, 0]
    nan = np.array([[np.nan, np.inf]], np.float32)
    assert O.remap_u8(sq, nan, np.zeros((1, 2), np.float32)).reshape(-1).tolist() == [0, 0]


def test_warp_u8_edge_columns_and_identity():  # perspective.rs:335-360, affine.rs:471-495
    src = np.array([[10, 20, 30, 40], [50, 60, 70, 80]], np.uint8)
    flip_h = [-1, 0, 3, 0, 1, 0, 0, 0, 1]
    assert O.warp_perspective_u8(src, flip_h, 4, 2).reshape(-1).tolist() == [40, 30, 20, 10, 80, 70, 60, 50]
    assert O.warp_affine_u8(src, flip_h[:6], 4, 2).reshape(-1).tolist() == [40, 30, 20, 10, 80, 70, 60, 50]
    img = O.pattern_u8(33 * 21 * 3).reshape(21, 33, 3)
    assert np.array_equal(O.warp_affine_u8(img, [1, 0, 0, 0, 1, 0], 33, 21), img)
    assert np.array_equal(O.warp_perspective_u8(img, [1, 0, 0, 0, 1, 0, 0, 0, 1], 33, 21), img)
    neg = [-v for v in [0.9, 0.12, 4.0, -0.08, 1.05, -2.0, 6.0e-4, -4.5e-4, 1.0]]
    pos = [-v for v in neg]
    assert np.array_equal(O.warp_perspective_u8(img, neg, 33, 21), O.warp_perspective_u8(img, pos, 33, 21))
    with pytest.raises(ValueError):
        O.warp_perspective_u8(img, [1, 2, 3, 2, 4, 6, 3, 6, 9], 33, 21)


def test_warp_affine_u8_tracks_the_f32_warp():
    """Q16 coordinates + Q10 weights stay within a few grey levels of the f32 bilinear warp."""
    img = O.pattern_u8(129 * 97).reshape(97, 129, 1)
    smooth = O.gaussian_blur_u8(img, (7, 7), (2.0, 2.0))[0]
    m = [1.2 * np.cos(0.3), 1.2 * np.sin(0.3), -8.0, -1.2 * np.sin(0.3), 1.2 * np.cos(0.3), 20.0]
    got = O.warp_affine_u8(smooth, m, 129, 97).astype(np.float32)
    ref = O.warp_affine(smooth.astype(np.float32), m, 129, 97, "bilinear")
    inner = (slice(2, -2), slice(2, -2))
    assert np.abs(got[inner] - ref[inner]).max() <= 3.0


def test_two_channel_gathers_are_per_channel_independent():
    """C = 2 (P/warp/cuda.rs:458-494): the Q10 sampler treats channels independently, so a 2-channel warp equals
    the two 1-channel warps interleaved — pins the restatement's channel indexing for the odd channel count."""
    src = O.pattern_u8(37 * 29 * 2).reshape(29, 37, 2)
    m = np.array([0.9, 0.1, 2.0, -0.05, 1.05, -1.0], np.float32)
    hm = np.array([1.02, 0.04, -1.5, -0.03, 0.97, 2.0, 2e-4, -1e-4, 1.0], np.float32)
    ys, xs = np.mgrid[0:23, 0:31].astype(np.float32)
    mx, my = (xs * 1.17 - 0.4).astype(np.float32), (ys * 1.21 + 0.3).astype(np.float32)
    for got, one in [(O.warp_affine_u8(src, m, 31, 23), lambda c: O.warp_affine_u8(src[:, :, c:c + 1].copy(), m, 31, 23)),
                     (O.warp_perspective_u8(src, hm, 31, 23), lambda c: O.warp_perspective_u8(src[:, :, c:c + 1].copy(), hm, 31, 23)),
                     (O.remap_u8(src, mx, my), lambda c: O.remap_u8(src[:, :, c:c + 1].copy(), mx, my))]:
        assert got.shape == (23, 31, 2) and got.any()
        for c in range(2):
            assert np.array_equal(got[:, :, c], one(c)[:, :, 0])
