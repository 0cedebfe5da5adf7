"""Host logic above the C ABI, no GPU: residency dispatch, typed errors, DLPack, sharding.

The reference tests the same things with a ``FakeDeviceResource`` that reports
``MemoryDomain::Device`` without a GPU (crates/kornia-tensor/src/storage.rs:493-517); here a
``Tensor`` built around a dummy device pointer plays that role — every check below fires before
any HIP call.  Cf. ``mixed_residency_is_a_typed_error`` / ``unsupported_channels_error_not_fallback``
(crates/kornia-imgproc/src/warp/cuda.rs:369-400) and kornia-py/tests/test_dlpack.py."""
import numpy as np
import pytest

import kornia_rs as K
from kornia_rs import Image, ImageError, Stream, Tensor, imgproc, sharding


def fake_device_image(w, h, c, dtype="float32", device=0, stream_handle=0):
    t = Tensor((h, w, c), dtype, device_ptr=0x10000, device=device, stream=Stream.from_handle(stream_handle, device))
    return Image(t)


def host_image(w, h, c, dtype="float32"):
    return Image.from_numpy(np.zeros((h, w, c), dtype))


def test_mixed_residency_is_a_typed_error():
    dev, host = fake_device_image(8, 8, 3), host_image(8, 8, 3)
    for a, b in ((dev, host), (host, dev)):
        with pytest.raises(ImageError) as e:
            imgproc.resize(a, out=b)
        assert e.value.kind == "MixedResidency"
        with pytest.raises(ImageError) as e:
            imgproc.gaussian_blur(a, (3, 3), (1.0, 1.0), dst=b)
        assert e.value.kind == "MixedResidency"


def test_host_pair_is_not_silently_computed():
    with pytest.raises(ImageError) as e:
        imgproc.gray_from_rgb(host_image(4, 4, 3, "uint8"))
    assert e.value.kind == "HostPathUnavailable"
    with pytest.raises(ImageError) as e:
        imgproc.resize(host_image(4, 4, 3), out=host_image(2, 2, 3))
    assert e.value.kind == "HostPathUnavailable"


def test_device_mismatch_and_unsupported_kernels():
    a, b = fake_device_image(8, 8, 3, device=0), fake_device_image(8, 8, 3, device=1)
    with pytest.raises(ImageError) as e:
        imgproc.resize(a, out=b)
    assert e.value.kind == "DeviceMismatch"
    with pytest.raises(ImageError) as e:  # unsupported_channels_error_not_fallback
        imgproc.warp_perspective(fake_device_image(8, 8, 2), [1, 0, 0, 0, 1, 0, 0, 0, 1], out=fake_device_image(8, 8, 2))
    assert e.value.kind == "NoDeviceKernel"
    with pytest.raises(ImageError) as e:
        imgproc.hsv_from_rgb(fake_device_image(8, 8, 3, "uint8"))
    assert e.value.kind == "NoDeviceKernel"
    with pytest.raises(ImageError) as e:
        imgproc.resize(fake_device_image(8, 8, 3), (4, 4), "area")
    assert e.value.kind == "NoDeviceKernel"
    with pytest.raises(ImageError) as e:  # u8 warps have the Q10 bilinear kernel only
        imgproc.warp_affine(fake_device_image(8, 8, 3, "uint8"), [1, 0, 0, 0, 1, 0], out=fake_device_image(8, 8, 3, "uint8"),
                            interpolation="bicubic")
    assert e.value.kind == "NoDeviceKernel"
    with pytest.raises(ImageError) as e:  # box_blur_u8 rejects even kernels (P/filter/ops.rs:66-75)
        imgproc.box_blur(fake_device_image(8, 8, 3, "uint8"), (4, 3), dst=fake_device_image(8, 8, 3, "uint8"))
    assert e.value.kind == "InvalidKernelLength"
    with pytest.raises(ImageError) as e:  # singular homography rejected on the host, before any launch
        imgproc.warp_perspective(a, [1, 2, 3, 2, 4, 6, 3, 6, 9], out=fake_device_image(8, 8, 3))
    assert e.value.kind == "CannotComputeDeterminant"
    with pytest.raises(ImageError) as e:
        imgproc.gaussian_blur(a, (4, 3), (1.0, 1.0), dst=fake_device_image(8, 8, 3))
    assert e.value.kind == "InvalidSigmaValue"
    with pytest.raises(ImageError) as e:
        imgproc.sobel(a, 7, dst=fake_device_image(8, 8, 3))
    assert e.value.kind == "InvalidKernelLength"
    with pytest.raises(ImageError) as e:
        imgproc.crop(a, 4, 4, 8, 8, dst=fake_device_image(8, 8, 3))
    assert e.value.kind == "PixelIndexOutOfBounds"
    with pytest.raises(ImageError) as e:
        imgproc.gray_from_rgb(fake_device_image(8, 8, 3, "uint8"), fake_device_image(9, 8, 1, "uint8"))
    assert e.value.kind == "InvalidImageSize"


def test_host_access_to_device_memory_is_refused():
    dev = fake_device_image(4, 4, 3)
    with pytest.raises(ImageError) as e:
        dev.as_slice()
    assert e.value.kind == "UnsupportedDevice"
    assert dev.device == "cuda:0" and dev.is_device and host_image(2, 2, 1).device == "cpu"
    with pytest.raises(AttributeError):
        host_image(2, 2, 1).__cuda_array_interface__
    cai = dev.__cuda_array_interface__
    assert cai["shape"] == (4, 4, 3) and cai["typestr"] == "<f4" and cai["data"] == (0x10000, False) and cai["version"] == 3


def test_image_basics_and_matrix_helpers():
    img = Image.from_numpy(np.arange(24, dtype=np.uint8).reshape(2, 4, 3))
    assert (img.width, img.height, img.channels, img.dtype, img.size) == (4, 2, 3, "uint8", (4, 2))
    assert img.numpy().base is not None or img.numpy().flags["C_CONTIGUOUS"]
    z = Image.zeros(5, 3, 1, "float32")
    assert z.shape == (3, 5, 1) and not z.is_device and float(z.numpy().sum()) == 0.0
    assert imgproc.invert_affine_transform([2, 0, 1, 0, 4, -2]) == [0.5, -0.0, -0.5, -0.0, 0.25, 0.5]
    m = imgproc.get_rotation_matrix2d((0.5, 0.5), 90.0, 1.0)
    assert abs(m[1] - 1.0) < 1e-6 and abs(m[0]) < 1e-6


def test_dlpack_host_round_trip_and_keepalive():
    from kornia_rs import dlpack
    a = np.arange(24, dtype=np.float32).reshape(2, 4, 3)
    t = Tensor.from_numpy(a)
    before = (dlpack.export_count, dlpack.release_count)
    b = np.from_dlpack(t)  # numpy consumes our capsule
    assert np.shares_memory(a, b) and np.array_equal(a, b)
    assert t.__dlpack_device__() == (dlpack.kDLCPU, 0)
    del b
    assert dlpack.export_count == before[0] + 1 and dlpack.release_count == before[1] + 1  # deleter ran once
    # import: numpy producer -> our Tensor / Image, zero-copy, producer kept alive
    src = np.arange(12, dtype=np.uint8).reshape(2, 2, 3)
    img = Image.from_dlpack(src)
    assert img.shape == (2, 2, 3) and img.dtype == "uint8" and not img.is_device
    assert img.numpy().ctypes.data == src.ctypes.data
    # non-contiguous producers are rejected (C-contiguous only, T/dlpack.rs:172-265)
    with pytest.raises(ValueError):
        Tensor.from_dlpack(np.arange(12, dtype=np.float32).reshape(3, 4).T)
    # an unconsumed capsule releases its keepalive when dropped
    before = dlpack.release_count
    cap = t.__dlpack__()
    del cap
    assert dlpack.release_count == before + 1


def test_dlpack_torch_host_interop():
    torch = pytest.importorskip("torch")
    t = Tensor.from_numpy(np.arange(6, dtype=np.float32).reshape(2, 3))
    tt = torch.from_dlpack(t)
    tt[0, 0] = 42.0  # shared memory
    assert t.numpy()[0, 0] == 42.0
    back = Tensor.from_dlpack(torch.arange(8, dtype=torch.int32).reshape(2, 4))
    assert back.dtype == "int32" and back.numpy().tolist() == [[0, 1, 2, 3], [4, 5, 6, 7]]


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 1024, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_range(2048, 3, 8) == (768, 1024)  # configs[4]: 256 images per GPU
    with pytest.raises(ValueError):
        sharding.shard_range(8, 8, 8)


def test_preprocessor_host_validation():
    from kornia_rs import PreprocessError, Preprocessor
    with pytest.raises(PreprocessError) as e:
        Preprocessor(std=(0.0, 0.2, 0.2), mean=(0.5, 0.5, 0.5), stream=Stream.default(0))
    assert e.value.kind == "InvalidNormalize"
    with pytest.raises(PreprocessError) as e:
        Preprocessor(sampling="bicubic", stream=Stream.default(0))
    assert e.value.kind == "UnsupportedSampling"
    with pytest.raises(PreprocessError) as e:
        Preprocessor()  # no stream = CPU preprocessor in the reference; not shipped here
    assert e.value.kind == "NotDeviceImage"
    pre = Preprocessor(format="nv12", stream=Stream.default(0))
    dst = Tensor((1, 1, 4, 4), "float32", device_ptr=0x1000, device=0, stream=Stream.default(0))
    with pytest.raises(PreprocessError) as e:
        pre.run_raw(0x2000, 8, 6, dst)
    assert e.value.kind == "BadOutputShape"
    with pytest.raises(PreprocessError) as e:
        pre.run_raw(0x2000, 8, 6, Tensor.zeros((1, 3, 4, 4)))
    assert e.value.kind == "NotDeviceTensor"
    assert K.preprocess.SourceFormat.from_name("NV12").buffer_len(8, 6) == 72
    assert K.preprocess.SourceFormat.from_name("yuyv").buffer_len(8, 6) == 96
    assert K.preprocess.SourceFormat.from_name("nope") is None


def test_color_space_tags_and_typed_constructors():  # I/color_spaces.rs:19-80, 269-620
    from kornia_rs import ColorSpace, ImageError, color_spaces as cs
    assert ColorSpace.GRAY.channels == 1 and ColorSpace.BGRA.channels == 4 and ColorSpace.LAB.channels == 3
    assert ColorSpace.LAB.float_only and not ColorSpace.RGB.float_only
    img = cs.Rgb8(np.zeros((4, 6, 3), np.uint8))
    assert img.color_space is ColorSpace.RGB and (img.width, img.height, img.channels) == (6, 4, 3)
    assert cs.Gray8(np.zeros((4, 6), np.uint8)).channels == 1
    assert cs.Labf32(np.zeros((2, 2, 3), np.float32)).color_space is ColorSpace.LAB
    for bad in (np.zeros((4, 6, 4), np.uint8), np.zeros((4, 6, 3), np.float32)):
        with pytest.raises(ImageError) as e:
            cs.Rgb8(bad)
        assert e.value.kind == "InvalidChannelShape"


def test_video_buffer_types_validate_layout():  # I/color_spaces.rs:630-830
    from kornia_rs import ImageError, color_spaces as cs, imgproc
    nv = cs.Nv12(8, 4, np.arange(48, dtype=np.uint8))
    assert nv.size == (8, 4) and nv.nbytes == 48 and not nv.is_device and nv.as_slice().shape == (48,)
    assert cs.I420.from_size_vec((8, 4), np.zeros(48, np.uint8)).layout == "i420"
    yu = cs.Yuyv8(8, 3, np.zeros(48, np.uint8))  # packed 4:2:2 allows an odd height
    assert yu.nbytes == 48 and yu.layout == "yuyv"
    for ctor, args in [(cs.Nv12, (8, 4, np.zeros(47, np.uint8))), (cs.Nv21, (7, 4, np.zeros(42, np.uint8))),
                       (cs.Yv12, (8, 3, np.zeros(36, np.uint8))), (cs.Uyvy8, (7, 2, np.zeros(28, np.uint8))),
                       (cs.Yvyu8, (8, 2, np.zeros(31, np.uint8)))]:
        with pytest.raises(ImageError) as e:
            ctor(*args)
        assert e.value.kind == "InvalidImageSize"
    with pytest.raises(ImageError) as e:  # host buffers never reach a device decoder implicitly
        imgproc.rgb_from_video(nv)
    assert e.value.kind == "HostPathUnavailable"


# ---- ConvertColor dispatch (P/color/convert.rs) --------------------------------------------------------

def test_convert_color_table_covers_the_reference_impls_and_rejects_the_rest():
    from kornia_rs import ImageError, color_spaces as cs, imgproc
    CS = cs.ColorSpace
    # every impl_convert! of convert.rs:108-238 that has a device arm resolves to an imgproc entry
    for (have, want), (name, dtypes) in cs._CONVERSIONS.items():
        assert callable(getattr(imgproc, name)), name
        assert have.channels in (1, 3, 4) and want.channels in (1, 3, 4) and dtypes
    assert len(cs._CONVERSIONS) == 24
    rgb = cs.Rgb8(np.zeros((4, 6, 3), np.uint8))
    assert rgb.color_space is CS.RGB and rgb.cpu().color_space is CS.RGB  # the tag travels with copies
    # host operands: classified, then refused (device backend only) — the table lookup itself succeeded
    with pytest.raises(ImageError) as e:
        cs.convert(rgb, cs.Gray8)
    assert e.value.kind == "HostPathUnavailable"
    with pytest.raises(ImageError) as e:  # float-only space from a u8 image: no impl (convert.rs:134-190)
        cs.convert(rgb, CS.HSV)
    assert e.value.kind == "NoDeviceKernel"
    with pytest.raises(ImageError) as e:  # no multi-hop routes: HSV -> LAB is not an impl
        cs.convert(cs.Hsvf32(np.zeros((2, 2, 3), np.float32)), CS.LAB)
    assert e.value.kind == "NoDeviceKernel"
    with pytest.raises(ImageError) as e:  # untyped image
        cs.convert(__import__("kornia_rs").Image.from_numpy(np.zeros((2, 2, 3), np.uint8)), CS.GRAY)
    assert e.value.kind == "InvalidChannelShape"
    with pytest.raises(ImageError) as e:  # background only for RGBA / BGRA sources
        cs.convert(rgb, CS.GRAY, background=(1, 2, 3))
    assert e.value.kind == "NoDeviceKernel"
    with pytest.raises(ImageError) as e:  # camera buffers decode to RGB8 only
        cs.convert(cs.Nv12(4, 2, np.zeros(12, np.uint8)), CS.GRAY)
    assert e.value.kind == "NoDeviceKernel"


# ---- named colour maps (P/color/colormap.rs) -----------------------------------------------------------

def test_named_colormaps_match_the_reference_digests():
    """All 21 named tables are bundled (19 rebuilt from public definitions by scripts/gen_colormaps.py; parula and deepgreen are
    literal 3 x 256 constant tables, shipped as data since round 3); their SHA-256 must equal the digests of the reference's
    tables (tests/golden/colormaps/reference_sha256.json, taken from colormap_luts.rs)."""
    import hashlib
    import json
    from pathlib import Path
    from kornia_rs import ColormapType, ImageError, colormap
    digests = json.loads((Path(__file__).parent / "golden" / "colormaps" / "reference_sha256.json").read_text())
    assert sorted(digests) == sorted(k.value for k in ColormapType) and len(digests) == 21  # colormap.rs:49-73
    assert len(colormap.bundled()) == 21 and set(digests) == set(colormap.bundled())
    for name in colormap.bundled():
        table = colormap.lut(name)
        assert table.shape == (3, 256) and table.dtype == np.uint8
        assert hashlib.sha256(table.tobytes()).hexdigest() == digests[name], name
    assert ColormapType.from_name("ViRiDiS") is ColormapType.VIRIDIS and ColormapType.from_name("nope") is None  # :78-84
    assert np.array_equal(colormap.lut(ColormapType.AUTUMN)[:, [0, 255]], [[255, 255], [0, 255], [0, 0]])
    with pytest.raises(ImageError) as e:
        colormap.lut("nope")
    assert "unknown" in str(e.value)


def test_device_video_frame_contract_without_a_device():  # P/cuda/color/video.rs:470-560
    from kornia_rs import ImageError
    from kornia_rs.color_spaces import DeviceVideoFrame, Nv12
    assert DeviceVideoFrame.buffer_len("yuyv", 64, 48) == 64 * 48 * 2 and DeviceVideoFrame.buffer_len("NV12", 64, 48) == 64 * 48 * 3 // 2
    with pytest.raises(ImageError) as e:
        DeviceVideoFrame.buffer_len("h264", 64, 48)
    assert e.value.kind == "InvalidArgument"
    with pytest.raises(ImageError) as e:  # short source: check_len before any upload
        DeviceVideoFrame.from_host(np.zeros(10, np.uint8), 4, 2, "nv12", None)
    assert e.value.kind == "InvalidImageSize"
    with pytest.raises(ImageError) as e:  # a host-resident buffer is not a device frame
        DeviceVideoFrame(Nv12(4, 2, np.zeros(12, np.uint8)))
    assert e.value.kind == "UnsupportedDevice"


def test_kornia_py_spellings_resolve_before_the_residency_check():
    """imgproc.pyi:80-160: resize(image, new_size, interpolation, antialias, out), dilate / erode(image, kernel="box",
    size=(h, w), border=...), gray_from_rgb_f32, apply_colormap(image, name).  On host images each call must get past its
    argument handling and stop at the typed residency error."""
    from kornia_rs import Image, ImageError, imgproc
    u8 = Image.from_numpy(np.zeros((8, 10, 3), np.uint8))
    f32 = Image.from_numpy(np.zeros((8, 10, 3), np.float32))
    calls = [lambda: imgproc.resize(u8, (4, 5), "lanczos", False), lambda: imgproc.resize(f32, (4, 5), "bicubic", True, None),
             lambda: imgproc.dilate(u8, "cross", size=(5, 5), border="reflect101"), lambda: imgproc.erode(u8, kernel="ellipse", size=(5, 5)),
             lambda: imgproc.dilate(u8, imgproc.Kernel("box", 3), "replicate"), lambda: imgproc.gray_from_rgb_f32(f32),
             lambda: imgproc.normalize_rgb_u8(u8, (1, 1, 1), (0, 0, 0)),
             lambda: imgproc.apply_colormap(Image.from_numpy(np.zeros((4, 4, 1), np.uint8)), "viridis"),
             lambda: imgproc.resize_mapped(f32, (4, 5), "bilinear", "align_corners"),
             lambda: imgproc.resize_bilinear_normalize(f32, (4, 5), (0, 0, 0), (1, 1, 1))]
    for call in calls:
        with pytest.raises(ImageError) as e:
            call()
        assert e.value.kind == "HostPathUnavailable", e.value
    k = imgproc._kernel_arg("ellipse", (3, 5))  # size is (height, width) in the Python API
    assert (k.height, k.width) == (3, 5)
    with pytest.raises(ImageError) as e:  # parse_kernel (kornia-py/src/morphology.rs:9-27): box / cross are square
        imgproc.dilate(u8, "cross", size=(3, 5))
    assert "square" in str(e.value)
    assert imgproc._border_arg("box", None, None) == "replicate" and imgproc._border_arg(k, None, None) == "constant"
    assert imgproc._border_arg("box", None, "wrap") == "wrap" and imgproc._border_arg(k, "reflect", None) == "reflect"
    with pytest.raises(ImageError) as e:
        imgproc.gray_from_rgb_f32(u8)
    assert e.value.kind == "NoDeviceKernel"
    with pytest.raises(ImageError) as e:
        imgproc.dilate(u8, "star")
    assert e.value.kind == "InvalidKernelShape"
    with pytest.raises(ImageError) as e:
        imgproc.resize_mapped(f32, (4, 5), "bilinear", "corner")
    assert e.value.kind in ("InvalidArgument", "HostPathUnavailable")


def test_graph_and_mem_info_fail_loudly_without_a_device():  # cuda.pyi:64-90
    from kornia_rs import _ffi, hip
    if hip.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_ffi.KorniaHipError):
        hip.mem_get_info()
    with pytest.raises(_ffi.KorniaHipError) as e:  # the default stream cannot be captured
        hip.Graph.capture(lambda: None, [], None)
    assert e.value.code == _ffi.KH_ERR_INVALID_ARG


def test_preprocessor_builder_mirrors_the_rust_api():
    """PreprocessorBuilder / Normalize / Preprocessor::{builder, letterbox, stretch, with_mode} (P/preprocess.rs:654-880): defaults,
    chaining and the validation `build_cuda` performs, all before any device work."""
    from kornia_rs import IMAGENET_MEAN, IMAGENET_STD, Normalize, PreprocessError, Preprocessor, PreprocessorBuilder, ResizeMode, SourceFormat
    s = Stream.default(0)
    pre = PreprocessorBuilder.new().build_hip(s)  # defaults: letterbox, unit scale, pad 114, bilinear, rgb8
    assert (pre.mode, pre.sampling, pre.source_format.name, float(pre.pad_value)) == (ResizeMode.LETTERBOX, "bilinear", "rgb8", 114.0)
    assert pre.mean.tolist() == [0, 0, 0] and pre.inv_std.tolist() == [1, 1, 1]
    pre = (Preprocessor.builder().source_format(SourceFormat.from_name("nv12")).mode(ResizeMode.STRETCH).normalize(Normalize.imagenet())
           .pad_value(0).sampling("lanczos").build_hip(s))
    assert (pre.mode, pre.sampling, pre.source_format.name, float(pre.pad_value)) == (ResizeMode.STRETCH, "lanczos", "nv12", 0.0)
    assert np.array_equal(pre.mean, np.asarray(IMAGENET_MEAN, np.float32))
    assert np.array_equal(pre.inv_std, (np.float32(1.0) / np.asarray(IMAGENET_STD, np.float32)).astype(np.float32))
    assert Normalize.imagenet() == Normalize.mean_std(IMAGENET_MEAN, IMAGENET_STD) and Normalize.unit_scale() != Normalize.imagenet()
    assert Preprocessor.letterbox(s).mode == ResizeMode.LETTERBOX and Preprocessor.stretch(s).mode == ResizeMode.STRETCH
    assert Preprocessor.with_mode(s, ResizeMode.STRETCH).mode == ResizeMode.STRETCH
    with pytest.raises(PreprocessError) as e:  # UnsupportedSampling (preprocess.rs:717-722)
        PreprocessorBuilder().sampling("bicubic").build_hip(s)
    assert e.value.kind == "UnsupportedSampling"
    with pytest.raises(PreprocessError) as e:  # InvalidNormalize (preprocess.rs:110-116)
        PreprocessorBuilder().normalize(Normalize.mean_std((0.5, 0.5, 0.5), (0.2, 0.0, 0.2))).build_hip(s)
    assert e.value.kind == "InvalidNormalize"
    with pytest.raises(PreprocessError):  # the CPU preprocessor is the reference crate's, not this backend's
        PreprocessorBuilder().build()
    with pytest.raises(ValueError):
        PreprocessorBuilder().pad_value(300)
    with pytest.raises(PreprocessError) as e:  # the _f16 twins take float16 destinations only
        Preprocessor.letterbox(s).run_raw_f16(0x1000, 8, 8, object())
    assert e.value.kind == "BadOutputShape"


def test_rust_api_names_tables_and_type_checks():
    """kornia_rs.rust_api: the Rust crate's free-function names.  Host-side tables equal the reference's
    (filter/kernels.rs tests :180-226, morphology/kernels.rs) and a wrong element type is a typed error before any device work."""
    from kornia_rs import Image, ImageError, rust_api as R
    assert R.sobel_kernel_1d(3) == ([-1.0, 0.0, 1.0], [1.0, 2.0, 1.0])  # test_sobel_kernel_1d
    assert R.sobel_kernel_1d(5) == ([-1.0, -2.0, 0.0, 2.0, 1.0], [1.0, 4.0, 6.0, 4.0, 1.0])
    assert R.scharr_kernel_1d(3) == ([-1.0, 0.0, 1.0], [3.0, 10.0, 3.0])  # test_scharr_kernel_1d
    for bad in (lambda: R.sobel_kernel_1d(7), lambda: R.scharr_kernel_1d(5), lambda: R.scharr_kernel_1d(7)):
        with pytest.raises(ImageError):
            bad()
    want = np.array([0.00026386508, 0.10645077, 0.78657067, 0.10645077, 0.00026386508], np.float32)  # test_gaussian_kernel_1d (assert_eq!)
    assert np.array_equal(np.array(R.gaussian_kernel_1d(5, 0.5), np.float32), want)
    assert R.box_blur_kernel_1d(4) == [0.25] * 4
    assert R.box_blur_fast_kernels_1d(1.0, 5) == [1, 1, 1, 1, 3]  # test_box_blur_fast_kernels_1d
    sx, sy = R.normalized_sobel_kernel3()
    assert sx[1] == [-0.25, 0.0, 0.25] and sy[2] == [0.125, 0.25, 0.125] and sum(map(sum, sx)) == 0
    cx, cy = R.normalized_scharr_kernel3()
    assert cx[0] == [-0.09375, 0.0, 0.09375] and cy[0] == [-0.09375, -0.3125, -0.09375]
    assert R.cross_kernel(3).data.tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]] and R.box_kernel(2).data.tolist() == [[1, 1], [1, 1]]
    assert R.ellipse_kernel(5, 5).data.shape == (5, 5)
    f = Image.from_numpy(np.zeros((4, 4, 3), np.float32))
    u = Image.from_numpy(np.zeros((4, 4, 3), np.uint8))
    for call in (lambda: R.gray_from_rgb_u8(f, u), lambda: R.gray_from_rgb_f32(u, f), lambda: R.hsv_from_rgb_f32(u, f),
                 lambda: R.ycc_from_rgb_u8(f, u), lambda: R.resize_fast_rgb(f, u, "bilinear"), lambda: R.resize_opencv_f32(u, u, "nearest"),
                 lambda: R.resize_fast_mono(u, u, "bilinear")):
        with pytest.raises(ImageError) as e:
            call()
        assert e.value.kind == "NoDeviceKernel"
    with pytest.raises(ImageError):
        R.ycc_from_rgb_u8(u, u, "xyz")
    assert R.spatial_gradient_float_parallel is R.spatial_gradient_float


# ---- allocator abstraction (a3: T/allocator.rs:73-144) ---------------------------------------------------------------------
def test_cpu_allocator_zeroed_aligned_and_shared_handle():
    """`cpu_allocate_zeroed_and_aligned` (T/allocator.rs:150-160) + the process-global handle."""
    import kornia_rs as K
    from kornia_rs.allocator import CpuAllocator, Layout, TensorAllocatorError, host_alloc
    r = CpuAllocator().allocate(Layout(64, 1))
    assert r.len_bytes() == 64 and r.domain == "host" and not r.is_readonly()
    assert not r.as_any().any()
    for align in (8, 64, 4096):
        r = host_alloc().allocate(Layout(1024, align))
        assert r.len_bytes() == 1024 and r.as_ptr() % align == 0
    assert host_alloc() is host_alloc()  # one shared, stateless handle
    assert host_alloc().allocate(Layout(0, 8)).len_bytes() == 0  # zero-size layouts are legal
    for bad in ((-1, 1), (8, 0), (8, 3)):
        with pytest.raises(TensorAllocatorError) as e:
            Layout(*bad)
        assert e.value.kind == "LayoutError"
    assert Layout.array("float32", 10).size == 40 and Layout.array("float64", 1).align == 8
    assert K.host_alloc() is host_alloc()


def test_tensors_remember_their_allocator():
    import kornia_rs as K
    from kornia_rs.allocator import CpuAllocator, ForeignAllocator, TensorAllocator, TensorAllocatorError
    t = K.Tensor.zeros((2, 3), "float32")
    assert isinstance(t.alloc, CpuAllocator) and t.alloc is K.host_alloc() and not t.numpy().any()
    assert isinstance(K.Tensor.from_numpy(np.ones((2, 2), np.uint8)).alloc, CpuAllocator)

    class Counting(TensorAllocator):  # a user allocator: the trait is open (object-safe, one method)
        calls = 0

        def allocate(self, layout):
            Counting.calls += 1
            return CpuAllocator().allocate(layout)
    a = Counting()
    t = K.Tensor.zeros_in((4, 5), "int32", a)
    assert t.alloc is a and Counting.calls == 1 and t.shape == (4, 5) and t.dtype == "int32" and not t.numpy().any()
    t.numpy()[1, 2] = 7  # host tensors are writable views of the resource
    assert t.numpy()[1, 2] == 7
    with pytest.raises(TensorAllocatorError) as e:
        ForeignAllocator().allocate(K.Layout(8, 1))
    assert e.value.kind == "CannotAllocateForeign"
