"""f64 colour family (P/color/cuda_dispatch.rs:48-61, 111-135).  CPU: the restatement is pinned on the reference's own
f64-vs-f32 tests and known answers, and the PRODUCT's per-pixel source (kh_color_f64.h, the file the gfx950 kernel
compiles) is built for the host and must equal the restatement bit for bit.  GPU legs live in
tests/test_zz_host_extras_gpu.py."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_ffi as O

ROOT = Path(__file__).resolve().parent.parent
NAMES = sorted(O.F64_CONV, key=O.F64_CONV.get)


def samples(conv_name, n=4096, seed=0):
    rng = np.random.default_rng(seed + O.F64_CONV[conv_name])
    cin = 1 if conv_name == "rgb_from_gray" else 3
    byte_domain = conv_name in ("hsv_from_rgb", "rgb_from_hsv", "hls_from_rgb", "rgb_from_hls")
    hi = 255.0 if byte_domain else 1.0
    x = rng.uniform(0.0, hi, (n, cin))
    x[: n // 8] = np.round(x[: n // 8] / hi * 255.0) / 255.0 * hi          # exact byte levels
    edge = np.array([0.0, hi, hi / 2, hi / 3, 1e-9, -0.25 * hi, 1.5 * hi, 0.04045, 0.0031308, 0.008856, 8.0 / 116.0])
    grid = np.stack(np.meshgrid(edge, edge, edge, indexing="ij"), -1).reshape(-1, 3)[:, :cin]
    if conv_name == "rgb_from_lab":
        x = np.stack([rng.uniform(0, 100, n), rng.uniform(-110, 110, n), rng.uniform(-110, 110, n)], -1)
    if conv_name == "rgb_from_luv":
        x = np.stack([rng.uniform(-1, 100, n), rng.uniform(-130, 220, n), rng.uniform(-140, 120, n)], -1)
    return np.ascontiguousarray(np.concatenate([x, grid]))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("f64") / "libcolor_f64_host.so"
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror",
           f"-I{ROOT / 'kornia-rs_amd' / 'csrc'}", str(ROOT / "tests" / "cpp" / "color_f64_host.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = C.CDLL(str(out))
    fp = np.ctypeslib.ndpointer(np.float64, flags="C")
    lib.host_color_convert_f64.argtypes = [fp, fp, C.c_size_t, C.c_int]
    return lib


@pytest.mark.parametrize("name", NAMES)
def test_product_pixel_source_equals_the_restatement_on_the_host(host_lib, name):
    x = samples(name)
    want = O.color_f64(name, x)
    got = np.empty_like(want)
    assert host_lib.host_color_convert_f64(x.reshape(-1), got.reshape(-1), x.shape[0], O.F64_CONV[name]) == 0
    assert got.tobytes() == want.tobytes(), f"{name}: max |diff| {np.nanmax(np.abs(got - want))}"
    special = np.array([[np.nan, 1.0, 2.0], [np.inf, 0.5, 0.5], [3.0, np.nan, -np.inf]])[:, : x.shape[1]].copy()
    want, got = O.color_f64(name, special), np.empty_like(O.color_f64(name, special))
    host_lib.host_color_convert_f64(special.reshape(-1), got.reshape(-1), 3, O.F64_CONV[name])
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    assert host_lib.host_color_convert_f64(x.reshape(-1), got.reshape(-1), 0, 99) == -1


def test_f64_arms_track_the_f32_arms_like_the_reference_tests():
    """P/color/hsv/mod.rs:216-235, P/color/hls/mod.rs f32_simd_matches_f64_scalar (< 1e-3 in the [0,255] domain); gray and
    YCbCr f64 are the same expression as f32 in double; the f64 YUV order only swaps the YCbCr chroma (yuv/mod.rs:102-117)."""
    rgb8 = O.pattern_u8(3 * 4000).astype(np.float64).reshape(-1, 3)
    for name in ("hsv_from_rgb", "hls_from_rgb"):
        f32 = O.color_map(name + "_f32", rgb8.astype(np.float32), 3).reshape(-1, 3)
        assert np.abs(O.color_f64(name, rgb8) - f32).max() < 1e-3, name
    for fwd, inv in (("hsv_from_rgb", "rgb_from_hsv"), ("hls_from_rgb", "rgb_from_hls")):
        assert np.abs(O.color_f64(inv, O.color_f64(fwd, rgb8)) - rgb8).max() < 1e-9
    unit = rgb8 / 255.0
    gray = O.color_f64("gray_from_rgb", unit)
    assert np.array_equal(gray[:, 0], 0.299 * unit[:, 0] + 0.587 * unit[:, 1] + 0.114 * unit[:, 2])
    assert np.abs(gray[:, 0] - O.color_map("gray_from_rgb_f32", unit.astype(np.float32), 1)).max() < 1e-6
    assert np.array_equal(O.color_f64("rgb_from_gray", gray), np.repeat(gray, 3, axis=1))
    ycc, yuv = O.color_f64("ycbcr_from_rgb", unit), O.color_f64("yuv_from_rgb", unit)
    assert np.array_equal(ycc[:, [0, 2, 1]], yuv)
    assert np.abs(ycc - O.color_map("ycc_from_rgb_f32", unit.astype(np.float32), 3, 0).reshape(-1, 3)).max() < 1e-6
    for fwd, inv in (("ycbcr_from_rgb", "rgb_from_ycbcr"), ("yuv_from_rgb", "rgb_from_yuv")):
        assert np.abs(O.color_f64(inv, O.color_f64(fwd, unit)) - unit).max() < 1e-12
    # known answers: pure red in the byte domain (hsv/mod.rs doc example), neutral grey in YCbCr
    assert np.allclose(O.color_f64("hsv_from_rgb", np.array([[255.0, 0.0, 0.0]])), [[0.0, 255.0, 255.0]])
    assert np.allclose(O.color_f64("hls_from_rgb", np.array([[0.0, 255.0, 0.0]])), [[85.0, 127.5, 255.0]])
    assert np.allclose(O.color_f64("ycbcr_from_rgb", np.array([[0.5, 0.5, 0.5]])), [[0.5, 0.5, 0.5]])
    # conv 0..7 route to the CIE scalar64 restatement that test_cie.py pins
    lab = O.color_f64("lab_from_rgb", unit[:64])
    assert np.array_equal(lab, O.cie("lab_from_rgb", unit[:64]))
