"""Seeded random differential sweep — ragged shapes, tiny images, extreme scales — of the operators against the CPU
restatement.  Every case is small, so the file costs seconds on the device and runs in the host simulator on a GPU-less box."""
import numpy as np
import pytest

import oracle_ffi as O

import os

pytestmark = pytest.mark.gpu
EXTRA = int(os.environ.get("KH_FUZZ_SEEDS", "0"))  # e.g. KH_FUZZ_SEEDS=50 python scripts/hostsim_run.py tests/test_fuzz_gpu.py -n 16
MODES = ("nearest", "bilinear", "bicubic", "lanczos")


def _f32(rng, h, w, c):
    return rng.uniform(-2.0, 2.0, (h, w, c)).astype(np.float32)


def _u8(rng, h, w, c):
    return rng.integers(0, 256, (h, w, c), dtype=np.uint8)


def _up(img, stream):
    from kornia_rs import Image
    return Image.from_numpy(img).to_hip(stream)


def _shape(rng, lo=1, hi=70):
    return int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_fuzz_f32_resize_and_warps(gpu_stream, seed):
    from kornia_rs import imgproc
    rng = np.random.default_rng(1000 + seed)
    for _ in range(10):
        (sh, sw), (dh, dw), c = _shape(rng), _shape(rng), int(rng.choice([1, 3, 4]))
        src = _f32(rng, sh, sw, c)
        dev, mode = _up(src, gpu_stream), str(rng.choice(MODES))
        mapping = str(rng.choice(["half_pixel", "align_corners"]))
        assert np.array_equal(imgproc.resize_mapped(dev, (dh, dw), mode, mapping).numpy(), O.resize_mapped(src, dw, dh, mode, mapping)), \
            ("resize", sw, sh, dw, dh, c, mode, mapping)
        m = [float(v) for v in (rng.uniform(0.5, 1.5), rng.uniform(-0.5, 0.5), rng.uniform(-8, 8), rng.uniform(-0.5, 0.5), rng.uniform(0.5, 1.5),
                                rng.uniform(-8, 8))]
        assert np.array_equal(imgproc.warp_affine(dev, m, (dh, dw), mode).numpy(), O.warp_affine(src, m, dw, dh, mode)), ("affine", sw, sh, dw, dh, c, mode, m)
        hm = m + [float(rng.uniform(-2e-3, 2e-3)), float(rng.uniform(-2e-3, 2e-3)), 1.0]
        assert np.array_equal(imgproc.warp_perspective(dev, hm, (dh, dw), mode).numpy(), O.warp_perspective(src, hm, dw, dh, mode)), \
            ("perspective", sw, sh, dw, dh, c, mode, hm)
        mx = rng.uniform(-2, sw + 1, (dh, dw)).astype(np.float32)
        my = rng.uniform(-2, sh + 1, (dh, dw)).astype(np.float32)
        got = imgproc.remap(dev, _up(mx, gpu_stream), _up(my, gpu_stream), mode).numpy()
        assert np.array_equal(got, O.remap(src, mx, my, mode)), ("remap", sw, sh, dw, dh, c, mode)


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_fuzz_round6_resize_shapes_and_list_batches(gpu_stream, seed):
    """Shapes drawn INTO the gates of the round-6 kernels: bilinear downscales with whole-float4 rows and a vertical step >= 1.5
    (row-streamed kernel, random part splits), bicubic at a horizontal step of exactly 2 (wave-shift kernel; vertical step 2 or not),
    near-horizontal and tilted bilinear warps (two / one pixel per lane) — each also through the pointer-list batch form with a
    random number of separately placed images."""
    from kornia_rs import Image, Tensor, imgproc
    from kornia_rs.hip import DeviceBuffer
    rng = np.random.default_rng(7000 + seed)

    def batch(arrs):   # images at unequally spaced offsets of one arena
        nb = arrs[0].nbytes
        arena = DeviceBuffer((nb + 256 * 9) * len(arrs) + 256, gpu_stream, zeroed=False)
        imgs, off = [], 0
        for k, a in enumerate(arrs):
            off += 256 * int(rng.integers(0, 8))
            arena.copy_from_host(a.reshape(-1), offset=off)
            imgs.append(Image(Tensor(a.shape, "float32", device_ptr=arena.ptr + off, device=gpu_stream.device, stream=gpu_stream, keepalive=arena)))
            off = (off + nb + 255) // 256 * 256
        return imgs

    for _ in range(4):
        c = int(rng.choice([1, 3, 4]))
        n = int(rng.integers(1, 6))
        # (a) bilinear downscale into the row-streamed kernel's gate
        dw, dh = int(rng.integers(1, 40)), int(rng.integers(1, 20))
        sw = int(rng.integers(max(4, dw), max(5, min(8 * dw, 32 * dw // c + 4)))) // 4 * 4 + (0 if c != 3 else 0)
        sw = max(4, sw)
        sh = int(rng.integers(int(1.5 * dh) + 1, 6 * dh + 3))
        arrs = [_f32(rng, sh, sw, c) for _ in range(n)]
        outs = imgproc.resize_batch(batch(arrs), (dh, dw), "bilinear")
        for k in range(n):
            assert np.array_equal(outs[k].numpy(), O.resize(arrs[k], dw, dh, "bilinear")), ("rows", sw, sh, dw, dh, c, k)
        # (b) bicubic, horizontal step exactly 2
        dw2, dh2 = int(rng.integers(1, 90)), int(rng.integers(1, 24))
        sh2 = 2 * dh2 if rng.integers(0, 2) else int(rng.integers(1, 60))
        arrs = [_f32(rng, sh2, 2 * dw2, c) for _ in range(n)]
        outs = imgproc.resize_batch(batch(arrs), (dh2, dw2), "bicubic")
        for k in range(n):
            assert np.array_equal(outs[k].numpy(), O.resize(arrs[k], dw2, dh2, "bicubic")), ("bicubic half", 2 * dw2, sh2, dw2, dh2, c, k)
        # (c) bilinear warps, nearly horizontal source runs or tilted ones
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 24))
        tilt = float(rng.choice([0.0, 0.01, 0.3]))
        m = [float(rng.uniform(0.8, 1.2)), float(rng.uniform(-0.1, 0.1)), float(rng.uniform(-5, 5)), tilt, float(rng.uniform(0.8, 1.2)), float(rng.uniform(-3, 3))]
        arrs = [_f32(rng, int(rng.integers(1, 40)) if False else h + 3, w + 2, c) for _ in range(n)]
        outs = imgproc.warp_affine_batch(batch(arrs), m, (h, w), "bilinear")
        hm = m + [float(rng.uniform(-1e-4, 1e-4)), float(rng.uniform(-1e-4, 1e-4)), 1.0]
        outs_p = imgproc.warp_perspective_batch(batch(arrs), hm, (h, w), "bilinear")
        for k in range(n):
            assert np.array_equal(outs[k].numpy(), O.warp_affine(arrs[k], m, w, h, "bilinear")), ("affine px", w, h, c, m, k)
            assert np.array_equal(outs_p[k].numpy(), O.warp_perspective(arrs[k], hm, w, h, "bilinear")), ("perspective px", w, h, c, hm, k)


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_fuzz_filters(gpu_stream, seed):
    from kornia_rs import imgproc
    rng = np.random.default_rng(2000 + seed)
    for _ in range(8):
        (h, w), c = _shape(rng), int(rng.choice([1, 3, 4]))
        src = _f32(rng, h, w, c)
        dev = _up(src, gpu_stream)
        k = (int(rng.choice([1, 3, 5, 7, 9, 11, 15, 17, 21])), int(rng.choice([1, 3, 5, 7, 9, 13, 15, 19])))
        s = (float(rng.uniform(0.3, 4.0)), float(rng.uniform(0.3, 4.0)))
        assert np.array_equal(imgproc.gaussian_blur(dev, k, s).numpy(), O.gaussian_blur(src, k, s)), ("gaussian", w, h, c, k, s)
        u = _u8(rng, h, w, c)
        udev = _up(u, gpu_stream)
        assert np.array_equal(imgproc.gaussian_blur(udev, k, s).numpy(), O.gaussian_blur_u8(u, k, s)[0]), ("gaussian_u8", w, h, c, k, s)
        assert np.array_equal(imgproc.box_blur(udev, k).numpy(), O.box_blur_u8(u, k)[0] if isinstance(O.box_blur_u8(u, k), tuple) else O.box_blur_u8(u, k)), \
            ("box_u8", w, h, c, k)


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_fuzz_u8_gathers_and_resize(gpu_stream, seed):
    from kornia_rs import ImageError, imgproc
    rng = np.random.default_rng(3000 + seed)
    for _ in range(10):
        (sh, sw), (dh, dw), c = _shape(rng), _shape(rng), int(rng.choice([1, 2, 3, 4]))
        src = _u8(rng, sh, sw, c)
        dev = _up(src, gpu_stream)
        m = [float(v) for v in (rng.uniform(0.5, 1.5), rng.uniform(-0.5, 0.5), rng.uniform(-8, 8), rng.uniform(-0.5, 0.5), rng.uniform(0.5, 1.5),
                                rng.uniform(-8, 8))]
        assert np.array_equal(imgproc.warp_affine(dev, m, (dh, dw)).numpy(), O.warp_affine_u8(src, np.array(m, np.float32), dw, dh)), ("affine_u8", sw, sh, dw, dh, c, m)
        hm = m + [float(rng.uniform(-2e-3, 2e-3)), float(rng.uniform(-2e-3, 2e-3)), 1.0]
        assert np.array_equal(imgproc.warp_perspective(dev, hm, (dh, dw)).numpy(), O.warp_perspective_u8(src, np.array(hm, np.float32), dw, dh)), \
            ("perspective_u8", sw, sh, dw, dh, c, hm)
        mx = rng.uniform(-2, sw + 1, (dh, dw)).astype(np.float32)
        my = rng.uniform(-2, sh + 1, (dh, dw)).astype(np.float32)
        for mode in ("nearest", "bilinear"):
            got = imgproc.remap(dev, _up(mx, gpu_stream), _up(my, gpu_stream), mode).numpy()
            assert np.array_equal(got, O.remap_u8(src, mx, my, mode)), ("remap_u8", sw, sh, dw, dh, c, mode)
        mode, aa = str(rng.choice(MODES)), bool(rng.integers(0, 2))
        try:
            want = O.resize_fast_u8(src, dw, dh, mode, aa)[0]
        except ValueError:  # the reference's typed errors (2 channels beyond nearest, 1-px-wide bilinear source): same on the device
            with pytest.raises(ImageError):
                imgproc.resize_fast(dev, (dh, dw), mode, aa)
            continue
        assert np.array_equal(imgproc.resize_fast(dev, (dh, dw), mode, aa).numpy(), want), ("resize_fast", sw, sh, dw, dh, c, mode, aa)


@pytest.mark.parametrize("seed", range(4 + EXTRA))
def test_fuzz_exact_double_and_half_resizes(gpu_stream, seed):
    """The exact-2x cases of the two u8 resizes run on shared special kernels since round 6 (rolling upscale walk with each resize's own
    arithmetic, packed-byte box): random shapes either side of their lane / wave seams, every channel count, both modes and both APIs."""
    from kornia_rs import imgproc
    rng = np.random.default_rng(13000 + seed)
    for _ in range(6):
        base = int(rng.choice([2, 8, 64, 256, 512, 1024]))
        w, h, c = max(2, base + int(rng.integers(-9, 10))), int(rng.integers(2, 9)), int(rng.choice([1, 3, 4]))
        small, big = _u8(rng, h, w, c), _u8(rng, 2 * h, 2 * w, c)
        ds, db = _up(small, gpu_stream), _up(big, gpu_stream)
        for mode in ("nearest", "bilinear"):
            assert np.array_equal(imgproc.resize_fast(ds, (2 * h, 2 * w), mode, True).numpy(), O.resize_fast_u8(small, 2 * w, 2 * h, mode, True)[0]), ("fast up", w, h, c, mode)
            assert np.array_equal(imgproc.resize_fast(db, (h, w), mode, True).numpy(), O.resize_fast_u8(big, w, h, mode, True)[0]), ("fast down", w, h, c, mode)
            assert np.array_equal(imgproc.resize_opencv(ds, (2 * h, 2 * w), mode).numpy(), O.resize_opencv(small, 2 * w, 2 * h, mode)), ("cv up", w, h, c, mode)
            assert np.array_equal(imgproc.resize_opencv(db, (h, w), mode).numpy(), O.resize_opencv(big, w, h, mode)), ("cv down", w, h, c, mode)


@pytest.mark.parametrize("seed", range(4 + EXTRA))
def test_fuzz_pyramid_morphology_pointwise(gpu_stream, seed):
    from kornia_rs import imgproc
    rng = np.random.default_rng(4000 + seed)
    for _ in range(8):
        (h, w), c = _shape(rng), int(rng.choice([1, 3, 4]))
        u, f = _u8(rng, h, w, c), _f32(rng, h, w, c)
        ud, fd = _up(u, gpu_stream), _up(f, gpu_stream)
        assert np.array_equal(imgproc.pyrdown(ud).numpy(), O.pyrdown(u)) and np.array_equal(imgproc.pyrdown(fd).numpy(), O.pyrdown(f)), ("pyrdown", w, h, c)
        assert np.array_equal(imgproc.pyrup(ud).numpy(), O.pyrup(u)) and np.array_equal(imgproc.pyrup(fd).numpy(), O.pyrup(f)), ("pyrup", w, h, c)
        shape = str(rng.choice(["box", "cross", "ellipse"]))
        kh_, kw_ = (int(rng.integers(1, 8)),) * 2 if shape != "ellipse" else (int(rng.integers(1, 8)), int(rng.integers(1, 8)))
        border = str(rng.choice(["constant", "replicate", "reflect101", "reflect", "wrap"]))
        if border in ("reflect101", "reflect", "wrap") and (kh_ // 2 >= h or kw_ // 2 >= w):
            border = "replicate"  # the mirrored / wrapped index needs the pad smaller than the image, as in the reference
        for op, fn in (("dilate", imgproc.dilate), ("erode", imgproc.erode)):
            got = fn(ud, shape, size=(kh_, kw_), border=border, constant_value=9).numpy()
            assert np.array_equal(got, O.morphology_u8(u, op, O.morph_kernel(shape, kw_, kh_), border, [9] * c)), (op, w, h, c, shape, kh_, kw_, border)
        assert np.array_equal(imgproc.horizontal_flip(ud).numpy(), u[:, ::-1]) and np.array_equal(imgproc.vertical_flip(fd).numpy(), f[::-1])
        lo, hi = imgproc.find_min_max(fd)
        assert (lo, hi) == (float(f.min()), float(f.max()))


@pytest.mark.parametrize("seed", range(4 + EXTRA))
def test_fuzz_tiled_kernels_large_shapes(gpu_stream, seed):
    """The LDS-tile kernels (morphology, pyrdown_u8, separable u8 resize) and the pyrup kernels only reach their interior / staged
    paths on images a few hundred pixels wide: random shapes in that range, every channel count, against the restatement."""
    from kornia_rs import imgproc
    rng = np.random.default_rng(9000 + seed)
    for _ in range(3):
        h, w, c = int(rng.integers(34, 150)), int(rng.integers(130, 700)), int(rng.choice([1, 3, 4]))
        u, f = _u8(rng, h, w, c), _f32(rng, h, w, c)
        ud, fd = _up(u, gpu_stream), _up(f, gpu_stream)
        assert np.array_equal(imgproc.pyrdown(ud).numpy(), O.pyrdown(u)) and np.array_equal(imgproc.pyrdown(fd).numpy(), O.pyrdown(f)), ("pyrdown", w, h, c)
        assert np.array_equal(imgproc.pyrup(ud).numpy(), O.pyrup(u)) and np.array_equal(imgproc.pyrup(fd).numpy(), O.pyrup(f)), ("pyrup", w, h, c)
        shape = str(rng.choice(["box", "box", "cross", "ellipse"]))
        k = int(rng.choice([3, 5, 7, 2, 4, 9]))
        kh_, kw_ = (k, k) if shape != "ellipse" else (int(rng.integers(1, 10)), int(rng.integers(1, 10)))
        border = str(rng.choice(["constant", "replicate", "reflect101", "reflect", "wrap"]))
        for op, fn in (("dilate", imgproc.dilate), ("erode", imgproc.erode)):
            got = fn(ud, shape, size=(kh_, kw_), border=border, constant_value=9).numpy()
            assert np.array_equal(got, O.morphology_u8(u, op, O.morph_kernel(shape, kw_, kh_), border, [9] * c)), (op, w, h, c, shape, kh_, kw_, border)
        dw, dh = int(rng.integers(20, 400)), int(rng.integers(10, 120))
        mode, aa = str(rng.choice(["bicubic", "lanczos"])), bool(rng.integers(0, 2))
        want = O.resize_fast_u8(u, dw, dh, mode, aa)[0]
        assert np.array_equal(imgproc.resize_fast(ud, (dh, dw), mode, aa).numpy(), want), ("resize_fast", w, h, dw, dh, c, mode, aa)


@pytest.mark.parametrize("seed", range(4 + EXTRA))
def test_fuzz_round6_rolling_kernels_wide_rows(gpu_stream, seed):
    """The round-6 rolling kernels (gray / RGBA morphology, shapes, gray pyramids and blurs with ragged widths, the row-store rule) meet
    their lane / wave / block seams on rows of about a thousand to four thousand pixels: random widths there, a few rows, every channel
    count, random shapes / borders / kernel sizes, against the restatement."""
    from kornia_rs import imgproc
    rng = np.random.default_rng(12000 + seed)
    for _ in range(3):
        base = int(rng.choice([256, 512, 1024, 2048, 4096]))
        w, h, c = base + int(rng.integers(-17, 18)), int(rng.integers(1, 12)), int(rng.choice([1, 3, 4]))
        u = _u8(rng, h, w, c)
        ud = _up(u, gpu_stream)
        assert np.array_equal(imgproc.pyrdown(ud).numpy(), O.pyrdown(u)), ("pyrdown", w, h, c)
        assert np.array_equal(imgproc.pyrup(ud).numpy(), O.pyrup(u)), ("pyrup", w, h, c)
        shape = str(rng.choice(["box", "box", "cross", "ellipse"]))
        k = int(rng.choice([3, 5, 7, 9, 13])) if shape == "box" else int(rng.choice([3, 5, 7]))
        border = str(rng.choice(["constant", "replicate", "reflect101", "reflect"]))
        if border in ("reflect101", "reflect") and k // 2 >= h:
            border = "replicate"
        cv = int(rng.choice([0, 9, 250]))
        for op, fn in (("dilate", imgproc.dilate), ("erode", imgproc.erode)):
            got = fn(ud, shape, size=(k, k), border=border, constant_value=cv).numpy()
            assert np.array_equal(got, O.morphology_u8(u, op, O.morph_kernel(shape, k, k), border, [cv] * c)), (op, w, h, c, shape, k, border, cv)
        kb = int(rng.choice([3, 5, 7, 9, 11]))
        sig = (float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.5, 3.0)))
        assert np.array_equal(imgproc.gaussian_blur(ud, (kb, kb), sig).numpy(), O.gaussian_blur_u8(u, (kb, kb), sig)[0]), ("gaussian u8", w, h, c, kb, sig)
        assert np.array_equal(imgproc.box_blur(ud, (kb, 3)).numpy(), O.box_blur_u8(u, (kb, 3))), ("box u8", w, h, c, kb)
        f = _f32(rng, h, w, c)
        fd = _up(f, gpu_stream)
        kf = int(rng.choice([3, 5, 9, 13, 19]))
        got = imgproc.gaussian_blur(fd, (kf, kf), sig).numpy()
        assert np.array_equal(got.view(np.uint32), O.gaussian_blur(f, (kf, kf), sig).view(np.uint32)), ("gaussian f32", w, h, c, kf, sig)


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_fuzz_fused_preprocess(gpu_stream, seed):
    """The north-star family: every source format x resize mode x sampler x f32 / f16 at random (even where 4:2:x needs it)
    geometries, including 1:1 (the specialised kernel), up- and down-scales.  Everything bit-exact, Lanczos included (host-built weights)."""
    from test_preprocess_gpu import _run, _raw_for
    rng = np.random.default_rng(5000 + seed)
    for _ in range(10):
        fmt = str(rng.choice(["rgb", "bgr", "rgba", "bgra", "gray", "nv12", "yuyv"]))
        w, h = 2 * int(rng.integers(1, 40)), 2 * int(rng.integers(1, 30))
        if rng.integers(0, 4) == 0:
            dw, dh = w, h  # scale 1: the identity-geometry variant
        else:
            dw, dh = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        mode, sampling = str(rng.choice(["letterbox", "stretch"])), str(rng.choice(["nearest", "bilinear", "lanczos"]))
        f16 = bool(rng.integers(0, 2))
        norm = dict(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) if rng.integers(0, 2) else {}
        raw = _raw_for(fmt, w, h, seed=int(rng.integers(0, 1000)))
        got = _run(gpu_stream, raw, w, h, dw, dh, fmt=fmt, mode=mode, sampling=sampling, f16=f16, **norm)
        want = O.preprocess(raw, w, h, dw, dh, fmt=fmt, mode=mode, sampling=sampling, f16=f16, **norm)
        what = (fmt, w, h, dw, dh, mode, sampling, f16, bool(norm))
        assert np.array_equal(got.view(np.uint16 if f16 else np.uint32), want.view(np.uint16 if f16 else np.uint32)), what


@pytest.mark.parametrize("seed", range(4 + EXTRA))
def test_fuzz_colour_and_camera_formats(gpu_stream, seed):
    """Pointwise colour maps at odd pixel counts (quad / tail paths of the u8 map kernel), camera-format decode / encode at
    random even sizes, Bayer at any size."""
    from kornia_rs import imgproc
    from kornia_rs.hip import DeviceBuffer
    rng = np.random.default_rng(6000 + seed)
    for _ in range(6):
        h, w = _shape(rng, 1, 50)
        u = _u8(rng, h, w, 3)
        ud = _up(u, gpu_stream)
        for name, cout, extra in (("gray_from_rgb_u8", 1, ()), ("bgr_from_rgb_u8", 3, ()), ("sepia_from_rgb_u8", 3, ()),
                                  ("ycc_from_rgb_u8", 3, (int(rng.integers(0, 2)),))):
            fn = {"gray_from_rgb_u8": imgproc.gray_from_rgb, "bgr_from_rgb_u8": imgproc.bgr_from_rgb, "sepia_from_rgb_u8": imgproc.sepia_from_rgb,
                  "ycc_from_rgb_u8": (imgproc.ycbcr_from_rgb if extra == (0,) else imgproc.yuv_from_rgb)}[name]
            assert np.array_equal(fn(ud).numpy().reshape(-1), O.color_map(name, u, cout, *extra)), (name, w, h)
        f = (u.astype(np.float32))
        fd = _up(f, gpu_stream)
        for name, fn in (("hsv_from_rgb_f32", imgproc.hsv_from_rgb), ("hls_from_rgb_f32", imgproc.hls_from_rgb)):
            assert np.array_equal(fn(fd).numpy().reshape(-1), O.color_map(name, f, 3)), (name, w, h)
        w2, h2 = 2 * int(rng.integers(1, 30)), 2 * int(rng.integers(1, 20))
        rgb = _u8(rng, h2, w2, 3)
        rd = _up(rgb, gpu_stream)
        nv12 = imgproc.nv12_from_rgb(rd)
        assert np.array_equal(nv12.numpy_raw().reshape(-1), O.nv12_from_rgb(rgb)), ("nv12_from_rgb", w2, h2)
        yuyv = imgproc.yuyv_from_rgb(rd)
        assert np.array_equal(yuyv.numpy_raw().reshape(-1), O.yuyv_from_rgb(rgb)), ("yuyv_from_rgb", w2, h2)
        for layout, decode in enumerate((imgproc.rgb_from_nv12, imgproc.rgb_from_nv21, imgproc.rgb_from_i420, imgproc.rgb_from_yv12)):
            raw = rng.integers(0, 256, w2 * h2 * 3 // 2, dtype=np.uint8)
            got = decode(DeviceBuffer.from_numpy(raw, gpu_stream), w2, h2).numpy()
            assert np.array_equal(got, O.rgb_from_nv12(raw, w2, h2, layout)), ("planar420", layout, w2, h2)
        for layout, decode in enumerate((imgproc.rgb_from_yuyv, imgproc.rgb_from_uyvy, imgproc.rgb_from_yvyu)):
            raw = rng.integers(0, 256, w2 * h2 * 2, dtype=np.uint8)
            got = decode(DeviceBuffer.from_numpy(raw, gpu_stream), w2, h2).numpy()
            assert np.array_equal(got, O.rgb_from_yuyv(raw, w2, h2, layout)), ("packed422", layout, w2, h2)
        mosaic = _u8(rng, h, w, 1)
        pattern = str(rng.choice(sorted(O.BAYER)))
        assert np.array_equal(imgproc.rgb_from_bayer(_up(mosaic, gpu_stream), pattern).numpy(), O.rgb_from_bayer(mosaic, pattern)), ("bayer", pattern, w, h)


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_fuzz_filter_extra(gpu_stream, seed):
    """Spatial gradients, box_blur_fast, median and bilateral at random ragged sizes / parameters."""
    from kornia_rs import ImageError, imgproc
    rng = np.random.default_rng(7000 + seed)
    for _ in range(8):
        (h, w), c = _shape(rng), int(rng.choice([1, 2, 3, 4]))
        f = _f32(rng, h, w, c)
        dev = _up(f, gpu_stream)
        for kind, fn in (("sobel", imgproc.spatial_gradient_float), ("scharr", imgproc.scharr_spatial_gradient_float)):
            gx, gy = fn(dev)
            wx, wy = O.spatial_gradient(f, kind)
            assert np.array_equal(gx.numpy(), wx) and np.array_equal(gy.numpy(), wy), (kind, w, h, c)
        sigma = (float(rng.uniform(0.3, 4.0)), float(rng.uniform(0.3, 4.0)))
        want = O.box_blur_fast(f, sigma)
        if want is None:  # a box wider than the image: the reference indexes out of bounds, the entry refuses
            with pytest.raises(ImageError):
                imgproc.box_blur_fast(dev, sigma)
        else:
            assert np.array_equal(imgproc.box_blur_fast(dev, sigma).numpy(), want), ("box_blur_fast", w, h, c, sigma)
        u = _u8(rng, h, w, c)
        if rng.random() < 0.3:
            u[:] = rng.integers(0, 256)  # flat images: every window element equal
        udev, k = _up(u, gpu_stream), int(rng.choice([3, 5]))
        assert np.array_equal(imgproc.median_blur(udev, k).numpy(), O.median_blur(u, k)), ("median", w, h, c, k)
        g = _u8(rng, h, w, 1)
        d = int(rng.choice([-1, 0, 3, 5, 7, 9]))
        sc, ss = float(rng.uniform(5.0, 120.0)), float(rng.uniform(0.8, 6.0))
        assert np.array_equal(imgproc.bilateral_filter(_up(g, gpu_stream), d, sc, ss).numpy(), O.bilateral_filter(g, d, sc, ss)), \
            ("bilateral", w, h, d, sc, ss)
