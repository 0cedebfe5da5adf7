"""Parity at the BASELINE *batch* sizes: every workload of the default bench line — the BASELINE configs and,
further down, every other `summary` row — runs ONE step at its
real batch — configs[2] N = 1024 (28.7 GB), configs[1,3,4] N = 256 (25.5 GB buffers), the u8 4K twins N = 256 — and a handful
of frames spread over the batch is copied back and compared bit-for-bit with the CPU restatement.  The sampled frames sit on
both sides of every offset where 32-bit arithmetic would wrap inside the batch buffer: 2^31 bytes, 2^32 bytes, 2^31 elements,
2^32 elements, plus the first and the last frame and (u8 staged gathers: 16 images per block, kh_u8.hip) a block seam.

The reference's contract for the batch loop: crates/kornia-imgproc/src/preprocess.rs:1258-1282 (frame k at src + k*stride,
dst + k*3*oh*ow) and its test :1852-1893.  Device-only: the host simulator would need the same 25 GB in host memory."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(autouse=True)
def _device_only():
    import os
    if os.environ.get("KH_HOSTSIM") == "1":
        pytest.skip("full BASELINE batches (25-29 GB) are a device-only check")


class _Args:
    batch = 0  # the BASELINE batch of each workload


def boundary_frames(n: int, frame_bytes: int, elem_bytes: int, extra=()):
    """Frame indices on both sides of the 2^31-B, 2^32-B, 2^31-element and 2^32-element offsets of an n-frame buffer, plus the
    first, the last and `extra`."""
    ks = {0, n - 1, *extra}
    for limit in (1 << 31, 1 << 32, (1 << 31) * elem_bytes, (1 << 32) * elem_bytes):
        k = limit // frame_bytes          # the frame that CONTAINS the limit
        ks.update((k - 1, k, k + 1))
    return sorted(k for k in ks if 0 <= k < n)


def _run(bench, name, stream):
    wl = bench.WORKLOADS[name](_Args)
    wl.setup(stream)
    wl.step()
    stream.synchronize()
    return wl


def _fetch(buf, k, dtype, shape):
    """Frame k of a batch buffer (DeviceBuffer or Tensor) as a host array — one frame's D2H, not the batch's."""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if hasattr(buf, "to_numpy"):
        return buf.to_numpy(dtype, shape, offset=k * nbytes)
    from kornia_rs.hip import d2h
    out = np.empty(shape, dtype)
    d2h(out, buf.data_ptr + k * nbytes, buf.stream)
    return out


def _same(got, want, what):
    g = np.ascontiguousarray(got).reshape(-1)
    w = np.ascontiguousarray(want).reshape(-1)
    g, w = (g.view(np.uint32), w.view(np.uint32)) if g.dtype == np.float32 else (g, w)
    assert g.shape == w.shape, (what, g.shape, w.shape)
    bad = np.flatnonzero(g != w)
    assert bad.size == 0, f"{what}: {bad.size} of {g.size} elements differ, flat span [{bad[0]}, {bad[-1]}]"


def test_boundary_frames_cover_the_wrap_points():
    # north star: 24 883 200-B output frames -> 2^31 B inside frame 86, 4 GiB inside 172, 2^31 floats inside 345, 2^32 floats inside 690
    ks = boundary_frames(1024, 1920 * 1080 * 12, 4)
    for k in (0, 85, 86, 87, 171, 172, 173, 344, 345, 346, 689, 690, 691, 1023):
        assert k in ks, k
    # 4K f32 x3 images (99 532 800 B): 2^31 B inside image 21, 4 GiB inside 43, 2^31 floats inside 86, 2^32 floats inside 172
    ks = boundary_frames(256, 3840 * 2160 * 12, 4)
    for k in (0, 21, 22, 43, 44, 86, 87, 172, 173, 255):
        assert k in ks, k


def test_north_star_full_batch(gpu_stream, bench):
    """configs[2]: 1024 NV12 1080p frames -> [1024,3,1080,1920] f32 in ONE launch; 3 110 400-B input frames, 24 883 200-B outputs."""
    wl = _run(bench, "nv12_chw", gpu_stream)
    assert wl.N == 1024
    ks = sorted(set(boundary_frames(wl.N, wl.W * wl.H * 12, 4)) | set(boundary_frames(wl.N, wl.frame_bytes, 1)))
    for k in ks:
        raw = wl.base[31 * k: 31 * k + wl.frame_bytes]
        want = O.preprocess(raw, wl.W, wl.H, wl.W, wl.H, fmt="nv12", mode="stretch", mean=MEAN, std=STD)[0]
        _same(_fetch(wl.dst, k, np.float32, (3, wl.H, wl.W)), want, f"nv12_chw frame {k}")


def test_north_star_frame_list_full_batch(gpu_stream, bench):
    """configs[2] through the reference's own batch signature: 1024 SEPARATELY ALLOCATED 1080p NV12 buffers handed to
    `Preprocessor.run_raw_batch` as a list -> kh_preprocess_to_chw_list, four launches of 256 frame bases each.  The frames on both
    sides of every launch slice (255 | 256, 511 | 512, 767 | 768) and of the destination's 2^31 / 2^32 offsets, bit for bit."""
    wl = _run(bench, "nv12_chw_list", gpu_stream)
    assert wl.N == 1024 and len(wl.frames) == 1024
    assert len({b.ptr - a.ptr for a, b in zip(wl.frames, wl.frames[1:])}) > 1, "equally spaced frames would take the strided launch"
    ks = sorted(set(boundary_frames(wl.N, wl.W * wl.H * 12, 4)) | {254, 255, 256, 257, 511, 512, 767, 768})
    for k in ks:
        raw = wl.base[31 * k: 31 * k + wl.frame_bytes]
        want = O.preprocess(raw, wl.W, wl.H, wl.W, wl.H, fmt="nv12", mode="stretch", mean=MEAN, std=STD)[0]
        _same(_fetch(wl.dst, k, np.float32, (3, wl.H, wl.W)), want, f"nv12_chw_list frame {k}")


def test_api_list_rows_full_batch(gpu_stream, bench):
    """configs[1] / [3] / [4] through imgproc.*_batch on 256 separately allocated Images each side (two launches of 128 (src, dst)
    pairs per operator): the images at the launch seam (127 | 128), the group-of-four seams of remap, the first and the last."""
    ks = (0, 1, 3, 4, 126, 127, 128, 129, 255)
    wl = _run(bench, "resize_224_api_list", gpu_stream)
    assert wl.N == 256
    n = wl.SW * wl.SH * wl.C
    for k in ks:
        want = O.resize(wl.base[31 * k: 31 * k + n].reshape(wl.SH, wl.SW, wl.C), wl.DW, wl.DH)
        _same(wl.dst[k].numpy(), want, f"resize_224_api_list image {k}")
    del wl
    wl = _run(bench, "gaussian_4k_api_list", gpu_stream)
    n = wl.W * wl.H * wl.C
    for k in (0, 127, 128, 255):
        want = O.gaussian_blur(wl.base[31 * k: 31 * k + n].reshape(wl.H, wl.W, wl.C), (7, 7), (1.5, 1.5))
        _same(wl.dst[k].numpy(), want, f"gaussian_4k_api_list image {k}")
    del wl
    wl = _run(bench, "undistort_warp_4k_api_list", gpu_stream)
    mx, my = O.correction_map(wl.INTR, wl.DIST, wl.W, wl.H)
    for k in (0, 3, 4, 127, 128, 255):
        mid = O.remap(wl.base[31 * k: 31 * k + n].reshape(wl.H, wl.W, wl.C), mx, my)
        _same(wl.tmp[k].numpy(), mid, f"undistort (list) remap image {k}")
        _same(wl.dst[k].numpy(), O.warp_perspective(mid, wl.hm, wl.W, wl.H), f"undistort (list) warp image {k}")


def test_north_star_letterbox_full_batch(gpu_stream, bench):
    """The 640x640 letterbox secondary at N = 1024 (4 915 200-B outputs: the wrap points sit at frames 436, 873)."""
    wl = _run(bench, "nv12_chw_640", gpu_stream)
    ks = sorted(set(boundary_frames(wl.N, 640 * 640 * 12, 4)) | set(boundary_frames(wl.N, wl.frame_bytes, 1)))
    for k in ks:
        raw = wl.base[31 * k: 31 * k + wl.frame_bytes]
        want = O.preprocess(raw, wl.W, wl.H, 640, 640, fmt="nv12", mode="letterbox", mean=MEAN, std=STD)[0]
        _same(_fetch(wl.dst, k, np.float32, (3, 640, 640)), want, f"nv12_chw_640 frame {k}")


def test_resize_full_batch(gpu_stream, bench):
    """configs[1]: 256 x (1920x1080x3 f32 -> 224x224x3): the SOURCE buffer is the 25.5 GB one."""
    wl = _run(bench, "resize_224", gpu_stream)
    assert wl.N == 256
    n = wl.SW * wl.SH * wl.C
    for k in boundary_frames(wl.N, n * 4, 4):
        want = O.resize(wl.base[31 * k: 31 * k + n].reshape(wl.SH, wl.SW, wl.C), wl.DW, wl.DH)
        _same(_fetch(wl.dst, k, np.float32, (wl.DH, wl.DW, wl.C)), want, f"resize_224 image {k}")


def test_gaussian_full_batch(gpu_stream, bench):
    """configs[3]: 256 x 3840x2160x3 f32, 7x7 gaussian, one launch over 25.5 GB in and out."""
    wl = _run(bench, "gaussian_4k", gpu_stream)
    assert wl.N == 256
    n = wl.W * wl.H * wl.C
    for k in boundary_frames(wl.N, n * 4, 4):
        want = O.gaussian_blur(wl.base[31 * k: 31 * k + n].reshape(wl.H, wl.W, wl.C), (7, 7), (1.5, 1.5))
        _same(_fetch(wl.dst, k, np.float32, (wl.H, wl.W, wl.C)), want, f"gaussian_4k image {k}")


def test_undistort_warp_full_batch(gpu_stream, bench):
    """configs[4] per-GPU share: remap then warp_perspective over 256 4K f32 images (three 25.5 GB buffers)."""
    wl = _run(bench, "undistort_warp_4k", gpu_stream)
    assert wl.N == 256
    n = wl.W * wl.H * wl.C
    mx, my = O.correction_map(wl.INTR, wl.DIST, wl.W, wl.H)
    for k in boundary_frames(wl.N, n * 4, 4, extra=(3, 4)):  # the maps are shared by groups of 4 images in the remap kernel
        src = wl.base[31 * k: 31 * k + n].reshape(wl.H, wl.W, wl.C)
        mid = O.remap(src, mx, my)
        _same(_fetch(wl.tmp, k, np.float32, (wl.H, wl.W, wl.C)), mid, f"undistort remap image {k}")
        _same(_fetch(wl.dst, k, np.float32, (wl.H, wl.W, wl.C)), O.warp_perspective(mid, wl.hm, wl.W, wl.H), f"undistort warp image {k}")


@pytest.mark.parametrize("name", ["warp_affine_u8_4k", "warp_perspective_u8_4k", "remap_u8_4k", "gaussian_u8_4k"])
def test_u8_4k_full_batch(gpu_stream, bench, name):
    """The u8 4K twins at N = 256 (24 883 200-B images: 2^31 B inside image 86, 4 GiB inside 172); the staged gather handles
    16 images per block (kh_u8.hip), so the 15|16 seam is sampled too."""
    wl = _run(bench, name, gpu_stream)
    assert wl.N == 256
    n = wl.W * wl.H * wl.C
    if name == "remap_u8_4k":
        mx, my = O.correction_map(bench.UndistortWarp4K.INTR, bench.UndistortWarp4K.DIST, wl.W, wl.H)
    for k in boundary_frames(wl.N, n, 1, extra=(15, 16)):
        src = wl.base[31 * k: 31 * k + n].reshape(wl.H, wl.W, wl.C)
        if name == "warp_affine_u8_4k":
            want = O.warp_affine_u8(src, np.array(list(wl.m), np.float32), wl.W, wl.H)
        elif name == "warp_perspective_u8_4k":
            want = O.warp_perspective_u8(src, wl.hm, wl.W, wl.H)
        elif name == "remap_u8_4k":
            want = O.remap_u8(src, mx, my, "bilinear")
        else:
            want = O.gaussian_blur_u8(src, (7, 7), (1.5, 1.5))[0]
        _same(_fetch(wl.dst, k, np.uint8, (wl.H, wl.W, wl.C)), want, f"{name} image {k}")


# ---- the other rows of the default bench line, at their bench batch (round 4: every `summary` row is backed by a parity check on the
# ---- bytes the timed launch writes) ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["warp_affine_f32_1080p", "sobel_4k", "box_blur_4k", "normalize_1080p"])
def test_same_size_f32_rows_full_batch(gpu_stream, bench, name):
    """1R + 1W f32x3 operators: warp_affine 1080p b256 (6.4 GB buffers), sobel / box blur 4K b128 (12.7 GB), normalize 1080p b512
    (12.7 GB).  The f32 geometry kernels store through a per-ROW window (kh_geom.hip::out_row): frames past 2^31 B / 2^32 B matter."""
    wl = _run(bench, name, gpu_stream)
    n = wl.W * wl.H * wl.C
    for k in boundary_frames(wl.N, n * 4, 4):
        img = wl.base[31 * k: 31 * k + n].reshape(wl.H, wl.W, wl.C)
        _same(_fetch(wl.dst, k, np.float32, (wl.H, wl.W, wl.C)), wl.oracle_call(O, img), f"{name} image {k}")


def test_resize_bicubic_full_batch(gpu_stream, bench):
    wl = _run(bench, "resize_bicubic_540", gpu_stream)
    assert wl.N == 256
    n = wl.SW * wl.SH * wl.C
    ks = sorted(set(boundary_frames(wl.N, n * 4, 4)) | set(boundary_frames(wl.N, wl.DW * wl.DH * wl.C * 4, 4)))
    for k in ks:
        want = O.resize(wl.base[31 * k: 31 * k + n].reshape(wl.SH, wl.SW, wl.C), wl.DW, wl.DH, "bicubic")
        _same(_fetch(wl.dst, k, np.float32, (wl.DH, wl.DW, wl.C)), want, f"resize_bicubic_540 image {k}")


@pytest.mark.parametrize("name", ["gray_u8_1080p", "gray_f32_1080p", "ycbcr_u8_1080p", "ycbcr_f32_1080p", "hsv_f32_1080p"])
def test_colour_map_rows_full_batch(gpu_stream, bench, name):
    """Pointwise maps over N x 1080p pixels in one launch (gray f32: 25.5 GB in, 8.5 GB out)."""
    wl = _run(bench, name, gpu_stream)
    dt = np.uint8 if wl.dtype == "u8" else np.float32
    n_in, n_out = wl.W * wl.H * wl.cin, wl.W * wl.H * wl.cout
    ks = sorted(set(boundary_frames(wl.N, n_in * wl.item, wl.item)) | set(boundary_frames(wl.N, n_out * wl.item, wl.item)))
    for k in ks[:: max(1, len(ks) // 12)] + [wl.N - 1]:
        img = wl.base[31 * k: 31 * k + n_in].reshape(wl.H, wl.W, wl.cin)
        want = O.color_map(wl.entry[3:], img, wl.cout, *wl.EXTRA.get(wl.entry, ()))
        _same(_fetch(wl.dst, k, dt, (wl.H, wl.W, wl.cout)), want, f"{name} frame {k}")


@pytest.mark.parametrize("name,fmt,out", [("nv12_chw_608", "nv12", 608), ("yuyv_chw_640", "yuyv", 640)])
def test_letterbox_secondaries_full_batch(gpu_stream, bench, name, fmt, out):
    wl = _run(bench, name, gpu_stream)
    ks = sorted(set(boundary_frames(wl.N, out * out * 12, 4)) | set(boundary_frames(wl.N, wl.frame_bytes, 1)))
    for k in ks:
        raw = wl.base[31 * k: 31 * k + wl.frame_bytes]
        want = O.preprocess(raw, wl.W, wl.H, out, out, fmt=fmt, mode="letterbox", mean=MEAN, std=STD)[0]
        _same(_fetch(wl.dst, k, np.float32, (3, out, out)), want, f"{name} frame {k}")


# ---- opt-in rows whose kernels write through an image-wide streaming-store window (round 4) --------------------------------------
@pytest.mark.parametrize("name", ["pyrdown_u8_4k", "pyrup_u8_4k", "pyrdown_f32_4k", "pyrup_f32_4k"])
def test_pyramid_rows_full_batch(gpu_stream, bench, name):
    wl = _run(bench, name, gpu_stream)
    if name == "pyrdown_u8_4k":
        sw, sh, dw, dh, es, dt, fn = wl.W, wl.H, wl.W // 2, wl.H // 2, 1, np.uint8, O.pyrdown
    else:
        sw, sh, dw, dh, es, dt, fn = wl.sw, wl.sh, wl.dw, wl.dh, wl.es, (np.float32 if wl.f32 else np.uint8), (O.pyrup if wl.up else O.pyrdown)
    n_in, n_out = sw * sh * 3, dw * dh * 3
    ks = sorted(set(boundary_frames(wl.N, n_in * es, es)) | set(boundary_frames(wl.N, n_out * es, es)))
    for k in ks:
        img = wl.base[31 * k: 31 * k + n_in].reshape(sh, sw, 3)
        _same(_fetch(wl.dst, k, dt, (dh, dw, 3)), fn(img), f"{name} image {k}")


def test_dilate_and_gradient_full_batch(gpu_stream, bench):
    wl = _run(bench, "dilate_u8_4k", gpu_stream)
    n = wl.W * wl.H * wl.C
    for k in boundary_frames(wl.N, n, 1):
        img = wl.base[31 * k: 31 * k + n].reshape(wl.H, wl.W, wl.C)
        _same(_fetch(wl.dst, k, np.uint8, (wl.H, wl.W, wl.C)), O.morphology_u8(img, "dilate", O.morph_kernel("box", 5)), f"dilate image {k}")
    del wl
    wl = _run(bench, "spatial_gradient_1080p", gpu_stream)
    n = wl.W * wl.H * wl.C
    for k in boundary_frames(wl.N, n * 4, 4):
        img = wl.base[31 * k: 31 * k + n].reshape(wl.H, wl.W, wl.C)
        gx, gy = O.spatial_gradient(img, "sobel")
        _same(_fetch(wl.dst, k, np.float32, (wl.H, wl.W, wl.C)), gx, f"spatial_gradient dx image {k}")
        _same(_fetch(wl.dst_y, k, np.float32, (wl.H, wl.W, wl.C)), gy, f"spatial_gradient dy image {k}")
