"""Parity of the rest of the filter module (spatial gradients, box_blur_fast, median_blur, bilateral_filter) through the
C ABI against the CPU restatement — bit-exact (f32) / byte-exact (u8), on the reference's own test shapes
(median.rs:1100-1140, bilateral.rs:447-468) plus ragged, 1-pixel-wide and batched cases."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from gpu_util import assert_same_bits, dev, out_buf

pytestmark = pytest.mark.gpu


def _lib():
    from kornia_rs import _ffi
    return _ffi


SHAPES_F32 = [(5, 5, 2), (1, 1, 1), (1, 9, 3), (7, 1, 1), (2, 2, 4), (37, 131, 3), (64, 200, 1), (19, 70, 5),
              # row length a multiple of 4 (with the 4-aligned batch stride below: the four-elements-per-thread kernel)
              (5, 8, 1), (7, 4, 1), (12, 6, 2), (9, 16, 3), (33, 100, 4), (3, 2, 2), (6, 1, 4), (17, 268, 3)]


@pytest.mark.parametrize("kind", ["sobel", "scharr"])
@pytest.mark.parametrize("shape", SHAPES_F32)
def test_spatial_gradient_bit_exact(gpu_stream, kind, shape):
    F = _lib()
    h, w, c = shape
    batch = 3
    imgs = O.pattern_f32(batch * h * w * c).reshape(batch, h, w, c)
    imgs[0].reshape(-1)[::7] *= -3.5  # mixed signs
    n = h * w * c
    stride = n + (8 if (w * c) % 4 == 0 else 5)  # padded batch stride; 4-aligned where the vector kernel can run
    src = np.zeros(batch * stride, np.float32)
    for k in range(batch):
        src[k * stride:k * stride + n] = imgs[k].reshape(-1)
    d_src, d_gx, d_gy = dev(gpu_stream, src), out_buf(gpu_stream, 4 * batch * stride), out_buf(gpu_stream, 4 * batch * stride)
    F.check(F.lib.kh_spatial_gradient_f32(gpu_stream.cuda_stream_ptr, d_src.ptr, d_gx.ptr, d_gy.ptr, w, h, c, O.GRADIENT_KINDS[kind], batch,
                                          stride, stride))
    gx, gy = d_gx.to_numpy(np.float32, (batch * stride,)), d_gy.to_numpy(np.float32, (batch * stride,))
    for k in range(batch):
        wx, wy = O.spatial_gradient(imgs[k], kind)
        assert_same_bits(gx[k * stride:k * stride + n].reshape(h, w, c), wx, f"dx {kind} {shape} image {k}")
        assert_same_bits(gy[k * stride:k * stride + n].reshape(h, w, c), wy, f"dy {kind} {shape} image {k}")
        pad = gx[k * stride + n:(k + 1) * stride].view(np.uint32)
        assert (pad == 0xFFFFFFFF).all()  # nothing written between images


def test_spatial_gradient_non_finite_inputs_follow_the_expression(gpu_stream):
    """Zero taps are multiplied too (inf * 0 = NaN in the reference): the kernel keeps them."""
    from kornia_rs import Image, imgproc
    img = O.pattern_f32(9 * 11).reshape(9, 11, 1)
    img[4, 5, 0] = np.inf
    img[2, 2, 0] = -0.0
    gx, gy = imgproc.spatial_gradient_float(Image.from_numpy(img).to_hip(gpu_stream))
    wx, wy = O.spatial_gradient(img, "sobel")
    assert_same_bits(gx.numpy(), wx, "dx")
    assert_same_bits(gy.numpy(), wy, "dy")
    assert np.isnan(wx[4, 5, 0]) and np.isnan(wy[4, 5, 0])


def test_spatial_gradient_python_mirror_and_known_answer(gpu_stream):
    from kornia_rs import Image, ImageError, imgproc
    ramp = np.stack([np.arange(25, dtype=np.float32), np.arange(25, dtype=np.float32) + 25.0], -1).reshape(5, 5, 2)
    src = Image.from_numpy(ramp).to_hip(gpu_stream)
    gx, gy = imgproc.spatial_gradient_float(src)
    assert np.array_equal(gx.numpy()[..., 0], np.tile(np.array([0.5, 1, 1, 1, 0.5], np.float32), (5, 1)))  # test_spatial_gradient
    assert np.array_equal(gy.numpy()[..., 1], np.repeat(np.array([2.5, 5, 5, 5, 2.5], np.float32), 5).reshape(5, 5))
    sx, sy = imgproc.scharr_spatial_gradient_float(src)
    assert sx.numpy()[2, 2, 0] == 1.0 and sy.numpy()[2, 2, 0] == 5.0  # test_scharr_spatial_gradient
    with pytest.raises(ImageError):  # InvalidImageSize
        imgproc.spatial_gradient_float(src, Image.zeros(4, 5, 2, "float32", gpu_stream))
    with pytest.raises(ImageError):
        imgproc.spatial_gradient_float(Image.from_numpy(ramp))  # host image: no silent CPU path


@pytest.mark.parametrize("shape,half", [((9, 14, 3), 0), ((9, 14, 3), 1), ((9, 14, 3), 13), ((1, 1, 1), 0), ((70, 3, 1), 2), ((5, 300, 4), 17),
                                        ((130, 33, 2), 5), ((300, 200, 3), 4), ((20, 120, 3), 30), ((4, 200, 3), 80), ((260, 70, 1), 9)])
def test_fast_horizontal_filter_bit_exact(gpu_stream, shape, half):
    F = _lib()
    h, w, c = shape
    batch = 2
    imgs = O.pattern_f32(batch * h * w * c).reshape(batch, h, w, c)
    n = h * w * c
    d_src, d_dst = dev(gpu_stream, imgs), out_buf(gpu_stream, 4 * batch * n)
    F.check(F.lib.kh_fast_horizontal_filter_f32(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, c, half, batch, n, n))
    got = d_dst.to_numpy(np.float32, (batch, w, h, c))
    for k in range(batch):
        assert_same_bits(got[k], O.fast_horizontal_filter(imgs[k], half), f"{shape} half {half} image {k}")


@pytest.mark.parametrize("shape,sigma", [((5, 5, 1), (0.5, 0.5)), ((24, 31, 3), (1.0, 2.0)), ((40, 17, 2), (3.0, 0.7)), ((64, 64, 1), (4.5, 4.5))])
def test_box_blur_fast_bit_exact(gpu_stream, shape, sigma):
    from kornia_rs import Image, imgproc
    h, w, c = shape
    img = np.arange(25, dtype=np.float32).reshape(5, 5, 1) if shape == (5, 5, 1) else O.pattern_f32(h * w * c).reshape(shape)
    got = imgproc.box_blur_fast(Image.from_numpy(img).to_hip(gpu_stream), sigma).numpy()
    assert_same_bits(got, O.box_blur_fast(img, sigma), f"{shape} {sigma}")
    if shape == (5, 5, 1):  # test_box_blur_fast: the reference's 25 exact floats
        assert got[0, 0, 0] == np.float32(4.444444) and got[2, 2, 0] == np.float32(12.0) and got[4, 4, 0] == np.float32(19.555555)


def test_box_blur_fast_rejects_what_the_reference_cannot_index(gpu_stream):
    from kornia_rs import Image, ImageError, imgproc
    assert imgproc.box_blur_fast_kernels_1d(0.5, 3) == [1, 1, 1] and imgproc.box_blur_fast_kernels_1d(1.0, 5) == [1, 1, 1, 1, 3]
    img = Image.from_numpy(O.pattern_f32(4 * 6).reshape(4, 6, 1)).to_hip(gpu_stream)
    with pytest.raises(ImageError):  # sigma 3 -> half widths 5 / 7: wider than the 4-row image
        imgproc.box_blur_fast(img, (0.5, 3.0))


MEDIAN_SHAPES = [(48, 64, 1), (43, 67, 1), (4, 5, 1), (1, 1, 1), (9, 2, 1), (21, 33, 3), (7, 6, 4), (5, 9, 2), (3, 129, 3), (2, 1, 1)]


@pytest.mark.parametrize("ksize", [3, 5])
@pytest.mark.parametrize("shape", MEDIAN_SHAPES)
def test_median_blur_byte_exact(gpu_stream, ksize, shape):
    F = _lib()
    h, w, c = shape
    batch = 2
    imgs = O.pattern_u8(batch * h * w * c).reshape(batch, h, w, c)
    n = h * w * c
    stride = n + 3
    src = np.zeros(batch * stride, np.uint8)
    for k in range(batch):
        src[k * stride:k * stride + n] = imgs[k].reshape(-1)
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, batch * stride)
    F.check(F.lib.kh_median_blur_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, c, ksize, batch, stride, stride))
    got = d_dst.to_numpy(np.uint8, (batch * stride,))
    for k in range(batch):
        assert_same_bits(got[k * stride:k * stride + n].reshape(h, w, c), O.median_blur(imgs[k], ksize), f"{shape} k={ksize} image {k}")
        assert (got[k * stride + n:(k + 1) * stride] == 0xFF).all()


def test_median_blur_python_mirror(gpu_stream):
    from kornia_rs import Image, ImageError, imgproc
    const = Image.from_numpy(np.full((12, 16, 1), 200, np.uint8)).to_hip(gpu_stream)
    assert (imgproc.median_blur(const, 3).numpy() == 200).all() and (imgproc.median_blur(const, 5).numpy() == 200).all()  # constant_image_unchanged
    for k in (4, 7, 1):  # rejects_bad_ksize_and_size_mismatch
        with pytest.raises(ImageError):
            imgproc.median_blur(const, k)
    with pytest.raises(ImageError):
        imgproc.median_blur(const, 3, Image.zeros(15, 12, 1, "uint8", gpu_stream))
    extremes = np.zeros((6, 8, 3), np.uint8)
    extremes[::2] = 255
    extremes[1, 3] = (7, 9, 250)
    assert np.array_equal(imgproc.median_blur(Image.from_numpy(extremes).to_hip(gpu_stream), 5).numpy(), O.median_blur(extremes, 5))


@pytest.mark.parametrize("d,sc,ss", [(5, 50.0, 50.0), (3, 25.0, 10.0), (9, 75.0, 75.0), (0, 30.0, 3.0)])
@pytest.mark.parametrize("shape", [(48, 64), (43, 67), (5, 9), (1, 1), (40, 16), (3, 15)])
def test_bilateral_byte_exact(gpu_stream, d, sc, ss, shape):
    from kornia_rs import Image, imgproc
    h, w = shape
    img = O.pattern_u8(h * w).reshape(h, w, 1)
    got = imgproc.bilateral_filter(Image.from_numpy(img).to_hip(gpu_stream), d, sc, ss).numpy()
    assert_same_bits(got, O.bilateral_filter(img, d, sc, ss), f"{shape} d={d} sc={sc} ss={ss}")


def test_bilateral_batch_tables_and_degenerate_sigma(gpu_stream):
    F = _lib()
    from kornia_rs import Image, ImageError, imgproc
    h, w, batch = 21, 37, 3
    imgs = O.pattern_u8(batch * h * w).reshape(batch, h, w, 1)
    n = h * w
    d_src, d_dst = dev(gpu_stream, imgs), out_buf(gpu_stream, batch * n)
    F.check(F.lib.kh_bilateral_filter_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, 5, 40.0, 12.0, batch, n, n))
    got = d_dst.to_numpy(np.uint8, (batch, h, w, 1))
    for k in range(batch):
        assert_same_bits(got[k], O.bilateral_filter(imgs[k], 5, 40.0, 12.0), f"image {k}")
    F.check(F.lib.kh_bilateral_filter_u8(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, w, h, 5, 0.0, 12.0, batch, n, n))  # copy-through
    assert np.array_equal(d_dst.to_numpy(np.uint8, (batch, h, w, 1)), imgs)
    src = Image.from_numpy(imgs[0]).to_hip(gpu_stream)
    assert np.array_equal(imgproc.bilateral_filter(src, 5, 50.0, 1e-7).numpy(), imgs[0])  # degenerate_sigma_copies_through
    const = Image.from_numpy(np.full((12, 16, 1), 200, np.uint8)).to_hip(gpu_stream)
    assert (imgproc.bilateral_filter(const, 5, 50.0, 50.0).numpy() == 200).all()  # constant_image_unchanged
    with pytest.raises(ImageError):  # three channels: no kernel, no fallback
        imgproc.bilateral_filter(Image.zeros(8, 8, 3, "uint8", gpu_stream))
    # the product's host tables equal the restatement's bit for bit (radius_rule_matches_cv2 and the table contents)
    for (d, sc, ss) in [(5, 50.0, 50.0), (3, 50.0, 50.0), (0, 50.0, 2.0), (-1, 50.0, 0.1), (9, 75.0, 75.0), (0, 30.0, 3.0), (7, 12.5, 4.0)]:
        t, want = imgproc.bilateral_tables(d, sc, ss), O.bilateral_tables(d, sc, ss)
        assert t["radius"] == want["radius"] and t["taps"] == list(zip(want["dy"].tolist(), want["dx"].tolist()))
        assert t["simd_order"] == want["simd_order"].tolist()
        assert_same_bits(t["space_weight"], want["space_weight"], "space")
        assert_same_bits(t["color_weight"], want["color_weight"], "color")
    assert imgproc.bilateral_tables(5, 50.0, 50.0)["radius"] == 2 and len(imgproc.bilateral_tables(5, 50.0, 50.0)["taps"]) == 13


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_median_width_sweep_exact_allocations(gpu_stream, c):
    """Every width 1..41 with tightly sized device images: the wide (dword) window loads must hand over to byte loads exactly
    where a dword would leave the row (the host simulator's AddressSanitizer build turns a 1-byte overrun into a failure)."""
    from kornia_rs import Image, imgproc
    for w in list(range(1, 42)) + [127, 128, 129]:
        img = O.pattern_u8(3 * w * c + w).reshape(-1)[: 3 * w * c].reshape(3, w, c)
        src = Image.from_numpy(img).to_hip(gpu_stream)
        for k in (3, 5):
            assert np.array_equal(imgproc.median_blur(src, k).numpy(), O.median_blur(img, k)), (w, c, k)
