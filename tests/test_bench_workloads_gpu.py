"""Every bench.py workload computes what it claims: at batch 2 and the REAL image sizes (1080p / 4K), one step of each
workload is read back and compared with the CPU restatement for both frames — the timed region skips no work and writes the
right bytes.  (Runs in the host simulator too; the 4K gathers take tens of seconds there.)"""
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


class _Args:
    batch = 2


def _run(bench, name, stream):
    wl = bench.WORKLOADS[name](_Args)
    wl.setup(stream)
    wl.step()
    stream.synchronize()
    return wl


def _out(wl, dtype, shape):
    dst = wl.dst
    if hasattr(dst, "numpy_raw"):  # Tensor
        return dst.numpy_raw().reshape((wl.N,) + shape)
    return dst.to_numpy(dtype, (wl.N,) + shape)


def _frame(wl, k, n, shape):
    return wl.base[31 * k: 31 * k + n].reshape(shape)


@pytest.mark.parametrize("name,size,fmt", [("nv12_chw", 0, "nv12"), ("nv12_chw_640", 640, "nv12"), ("nv12_chw_608", 608, "nv12"),
                                           ("yuyv_chw_640", 640, "yuyv")])
def test_north_star_workloads(gpu_stream, bench, name, size, fmt):
    wl = _run(bench, name, gpu_stream)
    oh, ow = (wl.H, wl.W) if size == 0 else (size, size)
    got = _out(wl, np.float32, (3, oh, ow))
    if size:  # the roofline numerator follows the variant that really runs: one tap on the 1/3 grid, four off it; padding reads nothing
        assert wl.variant == ("generic_bilinear_on_grid" if size == 640 else "generic"), wl.variant
        assert wl.taps == (1 if size == 640 else 4)
        assert wl.active_px == {640: 640 * 360, 608: 608 * 342}[size]
        assert wl.alg_bytes_per_launch == int(wl.N * (size * size * 12 + wl.active_px * wl.taps * wl.src_bytes_per_px))
    for k in range(wl.N):
        raw = wl.base[31 * k: 31 * k + wl.frame_bytes]
        want = O.preprocess(raw, wl.W, wl.H, ow, oh, fmt=fmt, mode="stretch" if size == 0 else "letterbox", mean=MEAN, std=STD)[0]
        if not np.array_equal(got[k], want):   # say where: a hole of zeros / stale bytes (a copy problem) looks different from wrong arithmetic
            bad = np.argwhere(got[k].view(np.uint32) != want.view(np.uint32))
            flat = np.flatnonzero(got[k].view(np.uint32).reshape(-1) != want.view(np.uint32).reshape(-1))
            raise AssertionError(f"{name} frame {k}: {len(bad)} of {want.size} words differ; first {bad[0].tolist()} last {bad[-1].tolist()}; flat span "
                                 f"[{flat[0]}, {flat[-1]}]; got {got[k].reshape(-1)[flat[:4]].tolist()} want {want.reshape(-1)[flat[:4]].tolist()}")


@pytest.mark.parametrize("name", ["nv12_h2d_preprocess", "nv12_h2d_preprocess_zero_copy"])
def test_h2d_preprocess_workload(gpu_stream, bench, name):
    """The capture-side workload: four steps through the upload ring (so both slots and a reused capture buffer are exercised); the
    last step's output equals the oracle on the frames of the capture buffer it consumed."""
    wl = bench.WORKLOADS[name](_Args)
    wl.setup(gpu_stream)
    for _ in range(wl.RING + 1):
        wl.step()
    gpu_stream.synchronize()
    assert wl.pageable == (name == "nv12_h2d_preprocess")   # the default row is the copy-into-the-ring contract; zero-copy is the opt-in
    got = _out(wl, np.float32, (3, wl.H, wl.W))
    b = (wl.turn - 1) % wl.RING
    for k in range(wl.N):
        i = b * wl.N + k
        want = O.preprocess(wl.base[31 * i: 31 * i + wl.frame_bytes], wl.W, wl.H, wl.W, wl.H, fmt="nv12", mode="stretch", mean=MEAN, std=STD)[0]
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), (name, k)
    extra = wl.roofline_extra(1e-3)   # (the host simulator's events have no resolution: only the keys are checked there)
    assert {"h2d_only_ms", "kernel_only_ms", "end_to_end_ms", "hidden_by_overlap_ms", "end_to_end_frac_of_pinned_h2d"} <= set(extra)


def test_frame_list_workload(gpu_stream, bench):
    """The north star through the reference's batch signature (a list of separately allocated frame buffers): same bits."""
    wl = _run(bench, "nv12_chw_list", gpu_stream)
    assert wl.kernel == "preprocess_nv12_identity_list" and wl.alg_bytes_per_launch == wl.N * (wl.frame_bytes + 12 * wl.W * wl.H)
    got = _out(wl, np.float32, (3, wl.H, wl.W))
    for k in range(wl.N):
        want = O.preprocess(wl.base[31 * k: 31 * k + wl.frame_bytes], wl.W, wl.H, wl.W, wl.H, fmt="nv12", mode="stretch", mean=MEAN, std=STD)[0]
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), k


def test_f16_twin_workload(gpu_stream, bench):
    """The north star's binary16 twin (run_raw_batch_f16): its own kernel since round 6; the oracle's f16 bits on every frame."""
    import ctypes as C
    from kornia_rs import _ffi
    wl = _run(bench, "nv12_chw_f16", gpu_stream)
    assert wl.kernel == "preprocess_nv12_identity_f16" and wl.alg_bytes_per_launch == wl.N * (wl.frame_bytes + 6 * wl.W * wl.H)
    p = wl.pre._params(wl.W, wl.H, wl.W, 1, _ffi.KH_FMT_NV12, wl.W, wl.H, wl.N, wl.frame_bytes, True, False)
    assert _ffi.lib.kh_preprocess_variant(C.byref(p)) == b"nv12_identity_f16"
    got = wl.dst.numpy_raw().view(np.uint16).reshape(wl.N, 3, wl.H, wl.W)
    for k in range(wl.N):
        want = O.preprocess(wl.base[31 * k: 31 * k + wl.frame_bytes], wl.W, wl.H, wl.W, wl.H, fmt="nv12", mode="stretch", f16=True, mean=MEAN, std=STD)[0]
        assert np.array_equal(got[k], want.view(np.uint16)), k


@pytest.mark.parametrize("name,size,fmt", [("nv12_chw_640_f16", 640, "nv12"), ("nv12_chw_608_f16", 608, "nv12"), ("yuyv_chw_640_f16", 640, "yuyv")])
def test_f16_letterbox_workloads(gpu_stream, bench, name, size, fmt):
    """The opt-in f16 letterbox rows: the oracle's binary16 bits on every frame; priced with two bytes per destination value."""
    wl = _run(bench, name, gpu_stream)
    assert wl.alg_bytes_per_launch == int(wl.N * (size * size * 6 + wl.active_px * wl.taps * wl.src_bytes_per_px))
    got = wl.dst.numpy_raw().view(np.uint16).reshape(wl.N, 3, size, size)
    for k in range(wl.N):
        want = O.preprocess(wl.base[31 * k: 31 * k + wl.frame_bytes], wl.W, wl.H, size, size, fmt=fmt, mode="letterbox", f16=True, mean=MEAN, std=STD)[0]
        assert np.array_equal(got[k], want.view(np.uint16)), (name, k)


@pytest.mark.parametrize("how", ["eager", "graph", "list"])
def test_resize_api_workloads(gpu_stream, bench, how):
    """configs[1] through imgproc.resize / hip.Graph / imgproc.resize_batch on separately allocated Images; ROTATE + 1 steps so that
    every rotating destination set (and every captured graph) has run and the first one has been rewritten."""
    wl = bench.WORKLOADS[f"resize_224_api_{how}"](_Args)
    wl.setup(gpu_stream)
    for _ in range(wl.ROTATE + 1):
        wl.step()
    gpu_stream.synchronize()
    n = wl.SW * wl.SH * wl.C
    for dst in wl.dsts:
        got = dst.to_numpy(np.float32, (wl.N, wl.DH, wl.DW, wl.C))
        for k in range(wl.N):
            assert np.array_equal(got[k], O.resize(_frame(wl, k, n, (wl.SH, wl.SW, wl.C)), wl.DW, wl.DH)), (how, k)


def test_list_workloads_4k(gpu_stream, bench):
    wl = _run(bench, "gaussian_4k_api_list", gpu_stream)
    n = wl.W * wl.H * wl.C
    got = _out(wl, np.float32, (wl.H, wl.W, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.gaussian_blur(_frame(wl, k, n, (wl.H, wl.W, wl.C)), (7, 7), (1.5, 1.5))), k
    wl = _run(bench, "undistort_warp_4k_api_list", gpu_stream)
    mx, my = O.correction_map(wl.INTR, wl.DIST, wl.W, wl.H)
    got = _out(wl, np.float32, (wl.H, wl.W, wl.C))
    for k in range(wl.N):
        want = O.warp_perspective(O.remap(_frame(wl, k, n, (wl.H, wl.W, wl.C)), mx, my), wl.hm, wl.W, wl.H)
        assert np.array_equal(got[k], want), k


def test_lanczos_secondary_workload(gpu_stream, bench):
    wl = _run(bench, "nv12_chw_640_lanczos", gpu_stream)
    got = _out(wl, np.float32, (3, 640, 640))
    for k in range(wl.N):
        raw = wl.base[31 * k: 31 * k + wl.frame_bytes]
        want = O.preprocess(raw, wl.W, wl.H, 640, 640, fmt="nv12", mode="letterbox", sampling="lanczos", mean=MEAN, std=STD)[0]
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), k  # host-built Lanczos weights: bit-exact since round 3


def test_resize_workloads(gpu_stream, bench):
    wl = _run(bench, "resize_224", gpu_stream)
    n = wl.SW * wl.SH * wl.C
    got = _out(wl, np.float32, (wl.DH, wl.DW, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.resize(_frame(wl, k, n, (wl.SH, wl.SW, wl.C)), wl.DW, wl.DH)), k
    # the line-granular floor (roofline.floor_bytes): between the tap bytes (alg) and the whole source; 128-byte lines
    fb = wl.floor_bytes()
    assert wl.alg_bytes_per_launch < fb < wl.N * (wl.SW * wl.SH + wl.DW * wl.DH) * wl.C * 4 and (fb - wl.N * wl.DW * wl.DH * wl.C * 4) % 128 == 0
    assert 2.5e9 < fb * 256 / wl.N < 3.2e9     # ~2.8 GB per 256-image step
    wl = _run(bench, "resize_normalize_f32_224", gpu_stream)
    got = _out(wl, np.float32, (wl.DH, wl.DW, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.resize_bilinear_normalize(_frame(wl, k, n, (wl.SH, wl.SW, wl.C)), wl.DW, wl.DH, MEAN, STD)), k
    wl = _run(bench, "resize_u8_224", gpu_stream)
    n8 = wl.W * wl.H * wl.C
    got = _out(wl, np.uint8, (wl.D, wl.D, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.resize_fast_u8(_frame(wl, k, n8, (wl.H, wl.W, wl.C)), wl.D, wl.D, "lanczos", True)[0]), k
    wl = _run(bench, "resize_norm_chw_224", gpu_stream)
    got = _out(wl, np.float32, (3, wl.D, wl.D))
    for k in range(wl.N):
        want = O.resize_normalize_to_chw(_frame(wl, k, n8, (wl.H, wl.W, wl.C)), wl.D, wl.D, wl.scale_np, wl.bias_np, "bilinear", True)[0]
        assert np.array_equal(got[k], want), k
    wl = _run(bench, "fused_rgb_640", gpu_stream)
    got = _out(wl, np.float32, (3, wl.D, wl.D))
    for k in range(wl.N):
        want = O.fused_pipeline(_frame(wl, k, n8, (wl.H, wl.W, wl.C)), wl.D, wl.D, [("normalize", [1 / 255.0] * 3, [0.0] * 3)], "chw")
        assert np.array_equal(got[k], want.reshape(got[k].shape)), k


def test_filter_workloads_4k(gpu_stream, bench):
    wl = _run(bench, "gaussian_4k", gpu_stream)
    n = wl.W * wl.H * wl.C
    got = _out(wl, np.float32, (wl.H, wl.W, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.gaussian_blur(_frame(wl, k, n, (wl.H, wl.W, wl.C)), (7, 7), (1.5, 1.5))), k
    wl = _run(bench, "gaussian_u8_4k", gpu_stream)
    got = _out(wl, np.uint8, (wl.H, wl.W, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.gaussian_blur_u8(_frame(wl, k, n, (wl.H, wl.W, wl.C)), (7, 7), (1.5, 1.5))[0]), k
    wl = _run(bench, "pyrdown_u8_4k", gpu_stream)
    got = _out(wl, np.uint8, (wl.H // 2, wl.W // 2, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.pyrdown(_frame(wl, k, n, (wl.H, wl.W, wl.C)))), k
    wl = _run(bench, "dilate_u8_4k", gpu_stream)
    got = _out(wl, np.uint8, (wl.H, wl.W, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.morphology_u8(_frame(wl, k, n, (wl.H, wl.W, wl.C)), "dilate", O.morph_kernel("box", 5), "constant", [0, 0, 0])), k


@pytest.mark.parametrize("name", ["pyrup_u8_4k", "pyrdown_f32_4k", "pyrup_f32_4k"])
def test_pyramid_workloads_4k(gpu_stream, bench, name):
    wl = _run(bench, name, gpu_stream)
    n = wl.sw * wl.sh * wl.C
    got = _out(wl, np.float32 if wl.f32 else np.uint8, (wl.dh, wl.dw, wl.C))
    for k in range(wl.N):
        src = _frame(wl, k, n, (wl.sh, wl.sw, wl.C))
        assert np.array_equal(got[k], O.pyrup(src) if wl.up else O.pyrdown(src)), k


def test_gather_workloads_4k(gpu_stream, bench):
    wl = _run(bench, "undistort_warp_4k", gpu_stream)
    n = wl.W * wl.H * wl.C
    # priced on the bytes the two kernels need (in-bounds taps, maps once per four images): below SURVEY 8(d)'s contract figure
    assert 0.75 * wl.survey_bytes_per_launch < wl.alg_bytes_per_launch < wl.survey_bytes_per_launch, (wl.alg_bytes_per_launch, wl.in_bounds)
    assert 0.9 < wl.in_bounds[0] <= 1.0 and 0.5 < wl.in_bounds[1] <= 1.0, wl.in_bounds
    assert all(0.5 < s_ <= 1.0 for s_ in wl.src_share), wl.src_share
    mx, my = O.correction_map(wl.INTR, wl.DIST, wl.W, wl.H)
    got = _out(wl, np.float32, (wl.H, wl.W, wl.C))
    for k in range(wl.N):
        want = O.warp_perspective(O.remap(_frame(wl, k, n, (wl.H, wl.W, wl.C)), mx, my), wl.hm, wl.W, wl.H)
        assert np.array_equal(got[k], want), k
    wl = _run(bench, "warp_affine_u8_4k", gpu_stream)
    got = _out(wl, np.uint8, (wl.H, wl.W, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.warp_affine_u8(_frame(wl, k, n, (wl.H, wl.W, wl.C)), np.array(list(wl.m), np.float32), wl.W, wl.H)), k
    wl = _run(bench, "warp_perspective_u8_4k", gpu_stream)
    got = _out(wl, np.uint8, (wl.H, wl.W, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.warp_perspective_u8(_frame(wl, k, n, (wl.H, wl.W, wl.C)), wl.hm, wl.W, wl.H)), k
    wl = _run(bench, "remap_u8_4k", gpu_stream)
    got = _out(wl, np.uint8, (wl.H, wl.W, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.remap_u8(_frame(wl, k, n, (wl.H, wl.W, wl.C)), mx, my, "bilinear")), k


def test_filter_extra_workloads_1080p(gpu_stream, bench):
    wl = _run(bench, "spatial_gradient_1080p", gpu_stream)
    n = wl.W * wl.H * wl.C
    gx, gy = _out(wl, np.float32, (wl.H, wl.W, wl.C)), wl.dst_y.to_numpy(np.float32, (wl.N, wl.H, wl.W, wl.C))
    for k in range(wl.N):
        wx, wy = O.spatial_gradient(_frame(wl, k, n, (wl.H, wl.W, wl.C)), "sobel")
        assert np.array_equal(gx[k], wx) and np.array_equal(gy[k], wy), k
    wl = _run(bench, "box_blur_fast_1080p", gpu_stream)
    got = _out(wl, np.float32, (wl.H, wl.W, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.box_blur_fast(_frame(wl, k, n, (wl.H, wl.W, wl.C)), (2.0, 2.0))), k


def test_north_star_operator_workloads(gpu_stream, bench):
    """The operators `north_star` names that joined the default run in round 4: bicubic resize, warp_affine f32, sobel, box_blur,
    normalize_mean_std — two images each at the real image size, bit for bit against the restatement."""
    wl = _run(bench, "resize_bicubic_540", gpu_stream)
    n = wl.SW * wl.SH * wl.C
    got = _out(wl, np.float32, (wl.DH, wl.DW, wl.C))
    for k in range(wl.N):
        assert np.array_equal(got[k], O.resize(_frame(wl, k, n, (wl.SH, wl.SW, wl.C)), wl.DW, wl.DH, "bicubic")), k
    for name in ("warp_affine_f32_1080p", "sobel_4k", "box_blur_4k", "normalize_1080p"):
        wl = _run(bench, name, gpu_stream)
        n = wl.W * wl.H * wl.C
        got = _out(wl, np.float32, (wl.H, wl.W, wl.C))
        for k in range(wl.N):
            want = wl.oracle_call(O, _frame(wl, k, n, (wl.H, wl.W, wl.C)))
            assert np.array_equal(got[k].view(np.uint32), np.ascontiguousarray(want).view(np.uint32).reshape(got[k].shape)), (name, k)


def test_colour_map_workloads_1080p(gpu_stream, bench):
    for name in ("gray_u8_1080p", "gray_f32_1080p", "hsv_f32_1080p", "bgr_u8_1080p", "ycbcr_u8_1080p", "ycbcr_f32_1080p"):
        wl = _run(bench, name, gpu_stream)
        np_dt = np.uint8 if wl.dtype == "u8" else np.float32
        n = wl.W * wl.H * wl.cin
        got = _out(wl, np_dt, (wl.H, wl.W, wl.cout))
        for k in range(wl.N):
            want = O.color_map(wl.entry[3:], _frame(wl, k, n, (wl.H, wl.W, wl.cin)), wl.cout, *wl.EXTRA.get(wl.entry, ()))
            assert np.array_equal(got[k].view(np.uint32 if np_dt == np.float32 else np.uint8),
                                  want.reshape(got[k].shape).view(np.uint32 if np_dt == np.float32 else np.uint8)), (name, k)
    wl = _run(bench, "gray_258x195", gpu_stream)  # BASELINE configs[0]
    got = wl.dst.to_numpy(np.uint8, (wl.H, wl.W))
    assert np.array_equal(got, O.gray_from_rgb_u8(wl.base.reshape(wl.H, wl.W, 3)).reshape(wl.H, wl.W))


def test_every_workload_is_covered(bench):
    covered = {"nv12_h2d_preprocess", "nv12_h2d_preprocess_zero_copy", "nv12_chw_list", "nv12_chw_f16", "nv12_chw_640_f16", "nv12_chw_608_f16", "yuyv_chw_640_f16", "resize_224_api_eager", "resize_224_api_graph", "resize_224_api_list",
               "gaussian_4k_api_list", "undistort_warp_4k_api_list", "nv12_chw", "nv12_chw_640", "nv12_chw_608", "yuyv_chw_640", "nv12_chw_640_lanczos", "resize_224", "resize_bicubic_540", "resize_normalize_f32_224",
               "resize_u8_224", "resize_norm_chw_224", "fused_rgb_640", "gaussian_4k", "sobel_4k", "box_blur_4k", "gaussian_u8_4k", "pyrdown_u8_4k",
               "pyrup_u8_4k", "pyrdown_f32_4k", "pyrup_f32_4k", "dilate_u8_4k", "undistort_warp_4k", "warp_affine_f32_1080p", "normalize_1080p",
               "warp_affine_u8_4k", "warp_perspective_u8_4k", "remap_u8_4k", "spatial_gradient_1080p", "box_blur_fast_1080p",
               "gray_u8_1080p", "gray_f32_1080p", "hsv_f32_1080p", "ycbcr_u8_1080p", "ycbcr_f32_1080p", "bgr_u8_1080p", "gray_258x195"}
    assert set(bench.ALSO_DEFAULT) <= set(bench.WORKLOADS)
    assert covered == set(bench.WORKLOADS)
