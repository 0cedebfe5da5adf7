"""Host conveniences on a device — ``kornia_rs.calibration`` (typed camera parameters -> correction maps -> remap;
P/calibration/distortion.rs:135-152, examples/undistort) and ``color_spaces.convert`` (the ConvertColor trait,
P/color/convert.rs).  Sorted last on purpose: both chain entry points the earlier files already pin individually."""
from dataclasses import astuple

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def test_undistort_image_matches_maps_then_remap(gpu_stream):
    from kornia_rs import Image, ImageError, calibration, imgproc
    intr = calibration.CameraIntrinsic(300.0, 300.0, 64.0, 48.0)
    dist = calibration.PolynomialDistortion(k1=0.1, k2=0.01, p1=1e-4, p2=1e-4)
    host = O.pattern_f32(129 * 97 * 3).reshape(97, 129, 3)
    src = Image.from_numpy(host).to_hip(gpu_stream)
    wx, wy = O.correction_map(astuple(intr), astuple(dist), 129, 97)
    want = O.remap(host, wx, wy)
    got = calibration.undistort_image(src, intr, dist)
    assert got.is_device and np.array_equal(got.numpy(), want)
    # cached maps + nearest, and the u8 twin through the same maps
    maps = calibration.generate_correction_map_polynomial(intr, dist, (129, 97), gpu_stream)
    assert np.array_equal(maps[0].numpy()[:, :, 0], wx) and np.array_equal(maps[1].numpy()[:, :, 0], wy)
    near = calibration.undistort_image(src, intr, dist, "nearest", maps=maps)
    assert np.array_equal(near.numpy(), O.remap(host, wx, wy, "nearest"))
    rgb = O.pattern_u8(129 * 97 * 3).reshape(97, 129, 3)
    und8 = calibration.undistort_image(Image.from_numpy(rgb).to_hip(gpu_stream), intr, dist, maps=maps)
    assert np.array_equal(und8.numpy(), O.remap_u8(rgb, wx, wy))
    with pytest.raises(ImageError) as e:
        calibration.undistort_image(Image.from_numpy(host), intr, dist)
    assert e.value.kind == "HostPathUnavailable"
    assert imgproc.crop is imgproc.crop_image


def test_convert_color_reference_cases(gpu_stream):  # P/color/convert.rs:280-560
    from kornia_rs import color_spaces as cs
    CS = cs.ColorSpace
    up = lambda img: img.to_hip(gpu_stream)
    # test_bgr_from_rgb / test_rgb_from_bgr
    bgr = cs.convert(up(cs.Rgb8(np.array([[[255, 128, 64]]], np.uint8))), cs.Bgr8)
    assert bgr.color_space is CS.BGR and bgr.numpy().reshape(-1).tolist() == [64, 128, 255]
    assert cs.convert(bgr, CS.RGB).numpy().reshape(-1).tolist() == [255, 128, 64]
    # test_gray_from_rgb_{f32,u8}, test_rgb_from_gray: shapes + values against the restatement
    rgbf = np.array([1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.5, 0.5, 0.5], np.float32).reshape(2, 2, 3)
    gray = cs.convert(up(cs.Rgbf32(rgbf)), cs.Grayf32)
    assert gray.shape == (2, 2, 1) and np.array_equal(gray.numpy().reshape(-1), O.color_map("gray_from_rgb_f32", rgbf, 1))
    g8 = cs.convert(up(cs.Rgb8(np.array([[[255, 0, 0], [0, 255, 0]]], np.uint8))), cs.Gray8)
    assert g8.shape == (1, 2, 1) and g8.numpy().reshape(-1).tolist() == [76, 150]
    back = cs.convert(up(cs.Grayf32(np.array([[0.0, 0.5], [1.0, 0.25]], np.float32))), cs.Rgbf32)
    assert back.shape == (2, 2, 3) and np.array_equal(back.numpy()[:, :, 1], np.array([[0.0, 0.5], [1.0, 0.25]], np.float32))
    # test_ycbcr_and_yuv_round_trip_u8
    data = ((np.arange(4 * 2 * 3) * 7 + 11) % 256).astype(np.uint8).reshape(2, 4, 3)
    rgb = up(cs.Rgb8(data))
    ycc = cs.convert(rgb, cs.YCbCr8)
    assert np.abs(cs.convert(ycc, cs.Rgb8).numpy().astype(int) - data.astype(int)).max() <= 3
    yuv = cs.convert(rgb, cs.Yuv8)
    assert yuv.color_space is CS.YUV and ycc.numpy()[0, 0, 0] == yuv.numpy()[0, 0, 0]
    assert np.abs(cs.convert(yuv, cs.Rgb8).numpy().astype(int) - data.astype(int)).max() <= 3
    # test_yuyv_decode_to_rgb: Y=16, U=V=128 -> black (limited range)
    black = cs.convert(cs.Yuyv8(2, 1, np.array([16, 128, 16, 128], np.uint8)).to_hip(gpu_stream), cs.Rgb8)
    assert black.color_space is CS.RGB and black.numpy().reshape(-1).tolist() == [0] * 6
    # test_hsv_from_rgb + float-only CIE spaces resolve; test_rgb_from_rgba{,_with_background}, test_rgba_from_rgb
    hsv = cs.convert(up(cs.Rgbf32(np.array([[[255.0, 0.0, 0.0]]], np.float32))), cs.Hsvf32)
    assert hsv.shape == (1, 1, 3) and hsv.color_space is CS.HSV
    lab = cs.convert(up(cs.Rgbf32(rgbf)), CS.LAB)
    assert np.abs(cs.convert(lab, CS.RGB).numpy() - rgbf).max() < 1e-4
    rgba = up(cs.Rgba8(np.array([[[255, 0, 0, 128], [0, 255, 0, 255]]], np.uint8)))
    assert cs.convert(rgba, cs.Rgb8).numpy().reshape(-1).tolist() == [255, 0, 0, 0, 255, 0]
    blended = cs.convert(rgba, cs.Rgb8, background=(100, 100, 100)).numpy().reshape(-1)
    assert blended.tolist() == [178, 50, 50, 0, 255, 0]  # convert.rs:498-520: 50% red over (100, 100, 100)
    again = cs.convert(up(cs.Rgb8(np.array([[[1, 2, 3]]], np.uint8))), cs.Rgba8)
    assert again.numpy().reshape(-1).tolist() == [1, 2, 3, 255]


def test_u8_gathers_two_channels(gpu_stream):  # warp_u8_c2_device_matches_cpu, P/warp/cuda.rs:458-494
    from kornia_rs import Image, imgproc
    src = O.pattern_u8(37 * 29 * 2).reshape(29, 37, 2)
    dev = Image.from_numpy(src).to_hip(gpu_stream)
    m = [0.9, 0.1, 2.0, -0.05, 1.05, -1.0]
    got = imgproc.warp_affine(dev, m, (23, 31), "bilinear")
    assert got.shape == (23, 31, 2) and np.array_equal(got.numpy(), O.warp_affine_u8(src, np.array(m, np.float32), 31, 23))
    hm = [1.02, 0.04, -1.5, -0.03, 0.97, 2.0, 2e-4, -1e-4, 1.0]
    got = imgproc.warp_perspective(dev, hm, (23, 31), "bilinear")
    assert np.array_equal(got.numpy(), O.warp_perspective_u8(src, np.array(hm, np.float32), 31, 23))
    ys, xs = np.mgrid[0:23, 0:31].astype(np.float32)
    mx, my = (xs * 1.17 - 0.4).astype(np.float32), (ys * 1.21 + 0.3).astype(np.float32)
    dmx, dmy = Image.from_numpy(mx).to_hip(gpu_stream), Image.from_numpy(my).to_hip(gpu_stream)
    for mode in ("bilinear", "nearest"):
        assert np.array_equal(imgproc.remap(dev, dmx, dmy, mode).numpy(), O.remap_u8(src, mx, my, mode)), mode
    # wide rows too: full-wave store path, odd width tail
    wide = O.pattern_u8(131 * 17 * 2).reshape(17, 131, 2)
    got = imgproc.warp_affine(Image.from_numpy(wide).to_hip(gpu_stream), [1.0, 0.02, 0.5, -0.01, 1.0, 0.25], (17, 131))
    assert np.array_equal(got.numpy(), O.warp_affine_u8(wide, np.array([1.0, 0.02, 0.5, -0.01, 1.0, 0.25], np.float32), 131, 17))


def test_apply_colormap_by_name(gpu_stream):  # P/color/colormap.rs:252-300
    from kornia_rs import ColormapType, Image, colormap, imgproc
    gray = O.pattern_u8(131 * 17).reshape(17, 131, 1)
    dev = Image.from_numpy(gray).to_hip(gpu_stream)
    for name in ("viridis", ColormapType.TURBO, "Winter"):
        table = colormap.lut(name)
        got = imgproc.apply_colormap(dev, name).numpy()
        assert got.shape == (17, 131, 3) and np.array_equal(got, table.T[gray[:, :, 0]])
    ramp = Image.from_numpy(np.arange(256, dtype=np.uint8).reshape(1, 256, 1)).to_hip(gpu_stream)
    assert np.array_equal(imgproc.apply_colormap(ramp, "autumn").numpy()[0], colormap.lut("autumn").T)


# ---- f64 colour family (P/color/cuda_dispatch.rs:48-61, 111-135) ---------------------------------------------

def _f64_samples(name):
    from test_color_f64 import samples
    return samples(name, n=20000, seed=3)


@pytest.mark.parametrize("name", sorted(O.F64_CONV, key=O.F64_CONV.get))
def test_color_f64_matches_cpu_arithmetic(gpu_stream, name):
    """The +,-,*,/ conversions (gray, hsv, hls, YCbCr, YUV, XYZ) must equal the CPU f64 path bit for bit; the ones through
    pow / cbrt (sRGB transfer, Lab, Luv) within 1e-9 of it — the reference holds its own f64 device twins to 1e-3 / 1e-4
    (P/cuda/color/cie.rs:554-620, hsv_hls.rs:466)."""
    from kornia_rs import _ffi
    from kornia_rs.hip import DeviceBuffer
    x = _f64_samples(name)
    want = O.color_f64(name, x)
    d_src = DeviceBuffer.from_numpy(x.reshape(-1), gpu_stream)
    d_dst = DeviceBuffer(want.nbytes, gpu_stream, zeroed=False)
    _ffi.check(_ffi.lib.kh_color_convert_f64(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, x.shape[0], O.F64_CONV[name]))
    got = d_dst.to_numpy(np.float64, want.shape)
    libm = name in ("linear_rgb_from_rgb", "rgb_from_linear_rgb", "lab_from_rgb", "rgb_from_lab", "luv_from_rgb", "rgb_from_luv")
    if libm:
        ok = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), ok)
        err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
        assert err.max() < 1e-9, f"{name}: {err.max()}"
    else:
        assert got.tobytes() == want.tobytes(), f"{name}: max |diff| {np.nanmax(np.abs(got - want))}"


def test_color_f64_through_the_host_api(gpu_stream):
    from kornia_rs import Image, ImageError, color_spaces as cs, imgproc
    rgb = (O.pattern_u8(3 * 33 * 17).astype(np.float64) / 255.0).reshape(17, 33, 3)
    dev = cs.Rgbf64(rgb).to_hip(gpu_stream)
    gray = cs.convert(dev, cs.Grayf64)
    assert gray.dtype == "float64" and gray.shape == (17, 33, 1)
    assert np.array_equal(gray.numpy(), O.color_f64("gray_from_rgb", rgb))
    assert np.array_equal(cs.convert(gray, cs.Rgbf64).numpy(), np.repeat(gray.numpy(), 3, axis=2))
    hsv = imgproc.hsv_from_rgb(Image.from_numpy(rgb * 255.0).to_hip(gpu_stream))
    assert np.array_equal(hsv.numpy(), O.color_f64("hsv_from_rgb", rgb * 255.0))
    yuv = cs.convert(dev, cs.Yuvf64)
    assert np.array_equal(yuv.numpy(), O.color_f64("yuv_from_rgb", rgb))
    lab = cs.convert(dev, cs.ColorSpace.LAB)
    assert lab.dtype == "float64" and np.abs(lab.numpy() - O.color_f64("lab_from_rgb", rgb)).max() < 1e-9
    with pytest.raises(ImageError) as e:  # no f64 BGR swizzle in the reference either (convert.rs:124-131)
        imgproc.bgr_from_rgb(dev)
    assert e.value.kind == "NoDeviceKernel"


# ---- YUYV mode decode (P/cuda/color/video.rs:128-190, yuyv_mode_decode_bit_exact_vs_cpu) --------------------------

@pytest.mark.parametrize("mode", sorted(O.YUV_MODE))
def test_yuyv_mode_decode_bit_exact(gpu_stream, mode):
    from kornia_rs import Image, ImageError, imgproc
    from kornia_rs.hip import DeviceBuffer
    for w, h in [(640, 37), (38, 7), (2, 1), (5, 3)]:  # (5, 3): odd width, last column untouched
        buf = O.pattern_u8(w * h * 2)
        dst = Image.from_numpy(np.full((h, w, 3), 77, np.uint8)).to_hip(gpu_stream)
        got = imgproc.convert_yuyv_to_rgb_u8(DeviceBuffer.from_numpy(buf, gpu_stream), w, h, mode, dst)
        assert got is dst and np.array_equal(got.numpy(), O.yuyv_to_rgb_mode(buf, w, h, mode, fill=77)), (mode, w, h)
    # every (Y, U, V) once: 256 rows of (u, v) pairs for each y would be 2^24 pairs = 64 MiB in, 96 MiB out
    yy, uu, vv = np.meshgrid(np.arange(0, 256, 1, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    buf = np.stack([yy, uu, yy, vv], -1).reshape(-1)
    got = imgproc.convert_yuyv_to_rgb_u8(DeviceBuffer.from_numpy(buf, gpu_stream), 512, 65536, mode)
    assert np.array_equal(got.numpy(), O.yuyv_to_rgb_mode(buf, 512, 65536, mode))
    with pytest.raises(ImageError):
        imgproc.convert_yuyv_to_rgb_u8(DeviceBuffer.from_numpy(buf[:16], gpu_stream), 4, 2, "bt2020")
    with pytest.raises(ImageError):  # buffer shorter than width * height * 2
        imgproc.convert_yuyv_to_rgb_u8(DeviceBuffer.from_numpy(buf[:15], gpu_stream), 4, 2, mode)


# ---- PixelMapping launchers + fused resize / normalise (P/cuda/resize.rs:184-236, 433-473, 580-650) ------------------

@pytest.mark.parametrize("mapping", ["half_pixel", "align_corners"])
def test_resize_mapped_matches_cpu_arithmetic(gpu_stream, mapping):
    from kornia_rs import Image, imgproc
    for c in (1, 3, 4):
        src = O.pattern_f32(131 * 77 * c).reshape(77, 131, c)
        dev = Image.from_numpy(src).to_hip(gpu_stream)
        for mode in ("nearest", "bilinear", "bicubic", "lanczos"):
            for dw, dh in [(64, 48), (200, 150), (131, 77), (1, 1)]:
                got = imgproc.resize_mapped(dev, (dh, dw), mode, mapping).numpy()
                assert np.array_equal(got, O.resize_mapped(src, dw, dh, mode, mapping)), (c, mode, mapping, dw, dh)
    src = O.pattern_f32(23 * 17 * 3).reshape(17, 23, 3)
    ac = imgproc.resize_mapped(Image.from_numpy(src).to_hip(gpu_stream), (9, 12), "bicubic", "align_corners").numpy()
    assert np.array_equal(ac, src[::2, ::2])  # every sample lands on a source pixel


def test_resize_bilinear_normalize_fused(gpu_stream):
    from kornia_rs import Image, ImageError, imgproc
    src = O.pattern_f32(1920 * 270 * 3).reshape(270, 1920, 3)
    dev = Image.from_numpy(src).to_hip(gpu_stream)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    for mapping in ("half_pixel", "align_corners"):
        for dw, dh in [(224, 224), (960, 135), (37, 11)]:
            got = imgproc.resize_bilinear_normalize(dev, (dh, dw), mean, std, mapping).numpy()
            assert np.array_equal(got, O.resize_bilinear_normalize(src, dw, dh, mean, std, mapping)), (mapping, dw, dh)
    # == resize then the separate normalisation pass up to the (x - m) * (1/s) vs (x - m) / s rounding
    two_pass = imgproc.normalize_mean_std(imgproc.resize(dev, (224, 224), "bilinear"), mean, std).numpy()
    fused = imgproc.resize_bilinear_normalize(dev, (224, 224), mean, std).numpy()
    assert np.abs(fused - two_pass).max() <= 1e-6 * np.abs(two_pass).max()
    with pytest.raises(ImageError):
        imgproc.resize_bilinear_normalize(dev, (8, 8), mean, (0.2, 0.0, 0.2))


def test_device_video_frame_convert_matches_cpu(gpu_stream):  # P/cuda/color/video.rs:647-690
    from kornia_rs import Image, ImageError
    from kornia_rs.color_spaces import ColorSpace, DeviceVideoFrame
    from kornia_rs.hip import DeviceBuffer
    w, h = 64, 48
    for fmt, decode, layout in [("yuyv", O.rgb_from_yuyv, 0), ("nv12", O.rgb_from_nv12, 0), ("yv12", O.rgb_from_nv12, 3), ("uyvy", O.rgb_from_yuyv, 1)]:
        raw = O.pattern_u8(DeviceVideoFrame.buffer_len(fmt, w, h) + 5)  # longer than needed: exactly buffer_len bytes go up
        frame = DeviceVideoFrame.from_host(raw, w, h, fmt, gpu_stream)
        assert (frame.width, frame.height, frame.format) == (w, h, fmt)
        rgb = frame.convert()
        assert rgb.color_space is ColorSpace.RGB and np.array_equal(rgb.numpy(), decode(raw[:DeviceVideoFrame.buffer_len(fmt, w, h)], w, h, layout))
        adopted = DeviceVideoFrame.from_device_buffer(DeviceBuffer.from_numpy(raw[:DeviceVideoFrame.buffer_len(fmt, w, h)], gpu_stream), w, h, fmt)
        dst = Image.zeros(w, h, 3, "uint8", gpu_stream)
        assert adopted.to_rgb(dst) is dst and np.array_equal(dst.numpy(), rgb.numpy())
    with pytest.raises(ImageError):
        frame.to_rgb(Image.zeros(w + 2, h, 3, "uint8", gpu_stream))


def test_kornia_py_spellings_on_device(gpu_stream):  # imgproc.pyi:80-160
    from kornia_rs import Image, imgproc
    rgb = O.pattern_u8(3 * 61 * 47).reshape(47, 61, 3)
    dev = Image.from_numpy(rgb).to_hip(gpu_stream)
    # resize(image, new_size, interpolation, antialias): antialias reaches the u8 separable kernels
    for aa in (True, False):
        got = imgproc.resize(dev, (20, 25), "lanczos", aa).numpy()
        assert np.array_equal(got, O.resize_fast_u8(rgb, 25, 20, "lanczos", aa)[0]), aa
    # dilate / erode(image, kernel="box", size=(h, w), border="replicate")
    assert np.array_equal(imgproc.dilate(dev).numpy(), O.morphology_u8(rgb, "dilate", O.morph_kernel("box", 3), "replicate"))
    got = imgproc.erode(dev, "ellipse", size=(3, 5), border="reflect101").numpy()
    assert np.array_equal(got, O.morphology_u8(rgb, "erode", O.morph_kernel("ellipse", 5, 3), "reflect101"))
    # normalize_rgb_u8: x * scale + offset (plain mul-add, the scalar expression)
    scale, offset = np.array([0.5, 0.25, 2.0], np.float32), np.array([-1.0, 0.5, 3.0], np.float32)
    got = imgproc.normalize_rgb_u8(dev, scale, offset).numpy()
    assert got.dtype == np.float32 and np.array_equal(got, (rgb.astype(np.float32) * scale + offset).astype(np.float32))
    f = Image.from_numpy(O.pattern_f32(3 * 61 * 47).reshape(47, 61, 3)).to_hip(gpu_stream)
    assert np.array_equal(imgproc.gray_from_rgb_f32(f).numpy().reshape(-1), O.color_map("gray_from_rgb_f32", f.numpy(), 1))


# ---- kornia_rs.cuda.Graph / mem_get_info (cuda.pyi:64-90, PY/cuda_ext/mod.rs:1684-1790) ------------------------------

def test_graph_capture_and_replay(gpu_stream):
    from kornia_rs import Image, hip, imgproc
    stream = hip.Stream.new(0)  # capture needs a non-default stream; operands live on it
    w, h = 64, 48
    rgb0 = O.pattern_u8(3 * w * h).reshape(h, w, 3)
    src = Image.from_numpy(rgb0).to_hip(stream)
    gray, small, blur = Image.zeros(w, h, 1, "uint8", stream), Image.zeros(32, 24, 3, "uint8", stream), Image.zeros(w, h, 3, "uint8", stream)
    stream.synchronize()

    def frame():  # allocation-free: every op writes into a preallocated destination
        imgproc.gray_from_rgb(src, gray)
        imgproc.resize(src, (24, 32), "nearest", True, small)
        imgproc.gaussian_blur(src, (5, 5), (1.2, 1.2), blur)

    free0, total = hip.mem_get_info()
    assert 0 < free0 <= total
    graph = hip.Graph.capture(frame, [src, gray, small, blur], stream)
    stream.synchronize()
    assert not gray.numpy().any()  # capture records, it does not run
    for k in range(3):  # new pixels in the same device memory, one launch per frame
        rgb = np.roll(rgb0, 7 * k + 1, axis=1).copy()
        _upload(src, rgb)
        graph.replay()
        stream.synchronize()
        assert np.array_equal(gray.numpy().reshape(-1), O.color_map("gray_from_rgb_u8", rgb, 1)), k
        assert np.array_equal(small.numpy(), O.resize_fast_u8(rgb, 32, 24, "nearest", True)[0]), k
        assert np.array_equal(blur.numpy(), O.gaussian_blur_u8(rgb, (5, 5), (1.2, 1.2))[0]), k
    with pytest.raises(ValueError):
        hip.Graph.capture(lambda: None, [], stream)  # nothing enqueued
    with pytest.raises(ZeroDivisionError):  # the callable's own error surfaces, and the stream is usable afterwards
        hip.Graph.capture(lambda: 1 / 0, [], stream)
    imgproc.gray_from_rgb(src, gray)
    stream.synchronize()
    free1, _ = hip.mem_get_info()
    assert abs(free1 - free0) <= 64 << 20  # no per-replay allocations


def _upload(img, array):
    """Overwrite a device image's pixels in place (same allocation, so a captured graph sees the new frame)."""
    from kornia_rs import _ffi
    a = np.ascontiguousarray(array)
    img.stream.synchronize()
    _ffi.check(_ffi.lib.kh_memcpy_h2d_async(img.data_ptr, a.ctypes.data, a.nbytes, img.stream.cuda_stream_ptr))
    img.stream.synchronize()


def test_cpp_mirror_full_surface_on_device(tmp_path_factory):
    """tests/cpp/host_mirror_ops_test.cpp `gpu`: the reference's known answers through every remaining C++ wrapper."""
    import test_cpp_mirror as T
    T._run(T.build(tmp_path_factory, "host_mirror_ops_test"), "gpu")


def test_min_max_ignores_nan_in_any_lane(gpu_stream):
    """find_min_max (P/normalize.rs:123-146): a NaN never wins a comparison.  A NaN in the FIRST lane of a wave whose other
    lanes hold the extremes used to drop that wave's contribution (the `any` flag was exchanged with a divergent shuffle)."""
    from kornia_rs import Image, imgproc
    x = np.linspace(1.0, 2.0, 64 * 5, dtype=np.float32)
    x[64] = np.nan          # lane 0 of wave 1
    x[70], x[100] = -7.5, 99.25  # ... whose other lanes hold the minimum and the maximum
    x[200] = np.nan
    lo, hi = imgproc.find_min_max(Image.from_numpy(x.reshape(1, -1, 1)).to_hip(gpu_stream))
    assert (lo, hi) == (-7.5, 99.25)
    y = x.copy()
    y[0] = np.nan           # a NaN FIRST element poisons both, as the reference loop does
    lo, hi = imgproc.find_min_max(Image.from_numpy(y.reshape(1, -1, 1)).to_hip(gpu_stream))
    assert np.isnan(lo) and np.isnan(hi)


# ---- Bayer demosaic (P/cuda/color/bayer.rs, P/color/bayer/mod.rs) -------------------------------------------------------------

@pytest.mark.parametrize("pattern", sorted(O.BAYER))
def test_bayer_demosaic_bit_exact(gpu_stream, pattern):
    from kornia_rs import Image, ImageError, color_spaces as cs, imgproc
    for (w, h) in [(4, 4), (5, 5), (7, 6), (33, 4), (40, 9), (3, 3), (2, 2), (1, 5), (6, 1), (1, 1), (131, 67), (640, 37)]:
        data = ((np.arange(w * h) * 37 + 11) % 256).astype(np.uint8).reshape(h, w, 1)
        got = imgproc.rgb_from_bayer(Image.from_numpy(data).to_hip(gpu_stream), pattern)
        assert got.shape == (h, w, 3) and np.array_equal(got.numpy(), O.rgb_from_bayer(data, pattern)), (pattern, w, h)
    mosaic = cs.Bayer8(O.pattern_u8(48 * 36).reshape(36, 48), pattern).to_hip(gpu_stream)
    rgb = cs.convert(mosaic, cs.Rgb8)
    assert rgb.color_space is cs.ColorSpace.RGB and np.array_equal(rgb.numpy(), O.rgb_from_bayer(mosaic.cpu().as_image().numpy(), pattern))
    with pytest.raises(ImageError):  # size_mismatch_errors
        imgproc.rgb_from_bayer(mosaic, None, Image.zeros(49, 36, 3, "uint8", gpu_stream))
    with pytest.raises(ImageError):
        imgproc.rgb_from_bayer(mosaic.as_image(), "rgbg")


# ---- the Rust-shaped Preprocessor constructors and _f16 twins (P/preprocess.rs:654-880, 1086-1282) -----------------------

def test_preprocessor_builder_and_f16_twins_on_device(gpu_stream):
    from kornia_rs import Normalize, PreprocessError, Preprocessor, SourceFormat, Tensor
    from kornia_rs.hip import DeviceBuffer
    w, h, dw, dh = 46, 30, 20, 16
    raw = O.pattern_u8(w * h * 3 // 2)
    pre = (Preprocessor.builder().source_format(SourceFormat.from_name("nv12")).normalize(Normalize.imagenet()).pad_value(114)
           .sampling("bilinear").build_hip(gpu_stream))
    src = DeviceBuffer.from_numpy(raw, gpu_stream)
    f32 = Tensor.uninit((1, 3, dh, dw), "float32", gpu_stream)
    f16 = Tensor.uninit((1, 3, dh, dw), "float16", gpu_stream)
    pre.run_raw(src, w, h, f32)
    pre.run_raw_f16(src, w, h, f16)
    want = O.preprocess(raw, w, h, dw, dh, fmt="nv12", mode="letterbox", mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))
    assert np.array_equal(f32.numpy_raw(), want)
    assert np.array_equal(f16.numpy_raw().view(np.uint16), want.astype(np.float16).view(np.uint16))
    two = Tensor.uninit((2, 3, dh, dw), "float16", gpu_stream)
    Preprocessor.builder().source_format("nv12").normalize(Normalize.imagenet()).build_hip(gpu_stream).run_raw_batch_f16([src, src], w, h, two)
    assert np.array_equal(two.numpy_raw()[1].view(np.uint16), f16.numpy_raw()[0].view(np.uint16))
    with pytest.raises(PreprocessError):  # an f32 destination is not an _f16 launch
        pre.run_raw_f16(src, w, h, f32)
    stretch = Preprocessor.stretch(gpu_stream)  # rgb8, unit scale
    rgb = O.pattern_u8(w * h * 3)
    out = Tensor.uninit((1, 3, dh, dw), "float32", gpu_stream)
    stretch.run_surface(DeviceBuffer.from_numpy(rgb, gpu_stream), w, h, w * 3, 3, out)
    assert np.array_equal(out.numpy_raw(), O.preprocess(rgb, w, h, dw, dh, fmt="rgb", mode="stretch"))


def test_rust_api_spellings_on_device(gpu_stream):
    """kornia_rs.rust_api: (src, dst, params) order, typed suffixes — same bytes as the restatement."""
    from kornia_rs import Image, rust_api as R
    w, h = 37, 23
    rgb = O.pattern_u8(w * h * 3).reshape(h, w, 3)
    src = Image.from_numpy(rgb).to_hip(gpu_stream)
    gray = Image.zeros(w, h, 1, "uint8", gpu_stream)
    assert R.gray_from_rgb_u8(src, gray) is None
    assert np.array_equal(gray.numpy().reshape(-1), O.color_map("gray_from_rgb_u8", rgb, 1))
    ycc = Image.zeros(w, h, 3, "uint8", gpu_stream)
    R.ycc_from_rgb_u8(src, ycc, "yuv")
    from kornia_rs import imgproc
    assert np.array_equal(ycc.numpy(), imgproc.yuv_from_rgb(src).numpy())
    small = Image.zeros(16, 12, 3, "uint8", gpu_stream)
    R.resize_fast_rgb_aa(src, small, "lanczos", False)
    assert np.array_equal(small.numpy(), O.resize_fast_u8(rgb, 16, 12, "lanczos", False)[0])
    R.resize_opencv_u8(src, small, "bilinear")
    assert np.array_equal(small.numpy(), O.resize_opencv(rgb, 16, 12, "bilinear"))
    f = Image.from_numpy(O.pattern_f32(w * h * 3).reshape(h, w, 3)).to_hip(gpu_stream)
    gx, gy = Image.zeros(w, h, 3, "float32", gpu_stream), Image.zeros(w, h, 3, "float32", gpu_stream)
    R.spatial_gradient_float_parallel_row(f, gx, gy)
    wx, wy = O.spatial_gradient(f.numpy(), "sobel")
    assert np.array_equal(gx.numpy(), wx) and np.array_equal(gy.numpy(), wy)
    nv12 = O.nv12_from_rgb(O.pattern_u8(32 * 16 * 3).reshape(16, 32, 3))
    out = Image.zeros(32, 16, 3, "uint8", gpu_stream)
    from kornia_rs.hip import DeviceBuffer
    R.rgb_from_planar420(DeviceBuffer.from_numpy(nv12, gpu_stream), 32, 16, out, "nv12")
    assert np.array_equal(out.numpy(), O.rgb_from_nv12(nv12, 32, 16))
