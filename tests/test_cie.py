"""CIE colour spaces: CPU pins of the oracle against the reference's tests (P/color/cie/mod.rs:150-341: f32 path
vs f64 formulas within per-channel tolerances, round trips, black / white) and GPU parity."""
import numpy as np
import pytest

import oracle_ffi as O

TOL = {"linear_rgb_from_rgb": [5e-4] * 3, "rgb_from_linear_rgb": [5e-4] * 3, "xyz_from_rgb": [5e-4] * 3, "rgb_from_xyz": [5e-4] * 3,
       "lab_from_rgb": [1e-2, 2e-2, 2e-2], "luv_from_rgb": [1e-2, 5e-2, 5e-2]}
PAIRS = [("linear_rgb_from_rgb", "rgb_from_linear_rgb"), ("xyz_from_rgb", "rgb_from_xyz"), ("lab_from_rgb", "rgb_from_lab"),
         ("luv_from_rgb", "rgb_from_luv")]


def pair(w, h):  # mod.rs:136-141
    v = ((np.arange(w * h * 3) * 7 % 251).astype(np.float32) / np.float32(250.0)).reshape(h, w, 3)
    return v, v.astype(np.float64)


@pytest.mark.parametrize("name", list(TOL))
def test_f32_path_tracks_f64_formulas(name):
    f32v, f64v = pair(7, 3)
    d = np.abs(O.cie(name, f32v).astype(np.float64) - O.cie(name, f64v))
    assert np.all(d.reshape(-1, 3).max(axis=0) <= np.array(TOL[name]))


@pytest.mark.parametrize("fwd,rev", PAIRS)
def test_round_trips(fwd, rev):  # mod.rs:216-241
    f32v, _ = pair(7, 3)
    back = O.cie(rev, O.cie(fwd, f32v))
    mask = np.ones(f32v.shape[:2], bool) if fwd != "luv_from_rgb" else f32v.sum(axis=2) > 0.05  # skip_near_zero
    assert np.abs(back - f32v)[mask].max() <= 1e-3


def test_black_and_white_known_values():  # mod.rs:243-275
    bw = np.array([[[0, 0, 0], [1, 1, 1]]], np.float32)
    lab = O.cie("lab_from_rgb", bw)
    assert abs(lab[0, 0, 0]) < 1e-3 and abs(lab[0, 1, 0] - 100.0) < 1e-2 and np.abs(lab[0, :, 1:]).max() < 0.1
    luv = O.cie("luv_from_rgb", bw)
    assert luv[0, 0].tolist() == [0.0, 0.0, 0.0] and abs(luv[0, 1, 0] - 100.0) < 1e-2
    xyz = O.cie("xyz_from_rgb", bw)
    assert np.allclose(xyz[0, 1], [0.950456, 1.0, 1.088754], atol=1e-5)
    assert O.cie("rgb_from_luv", np.array([[[0.0, 5.0, -3.0]]], np.float32)).reshape(-1).tolist() == [0.0, 0.0, 0.0]
    neg = O.cie("linear_rgb_from_rgb", np.array([[[-0.5, 0.04045, 1.0]]], np.float32)).reshape(-1)
    assert neg[0] == 0.0 and abs(neg[1] - 0.04045 / 12.92) < 1e-7 and abs(neg[2] - 1.0) < 1e-6


# ---- GPU --------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name", list(O.CIE))
def test_device_cie_matches_oracle(gpu_stream, name, libm_agrees):
    from kornia_rs import _ffi
    from gpu_util import assert_same_bits, dev, out_buf
    rng = np.random.default_rng(11)
    src = rng.random((97, 129, 3)).astype(np.float32)
    src[0, :4] = [[0, 0, 0], [1, 1, 1], [0.04045, 0.0031308, 0.008856], [-0.25, 1.5, 0.5]]
    if name in ("rgb_from_lab", "rgb_from_luv", "rgb_from_xyz", "rgb_from_linear_rgb"):
        fwd = {"rgb_from_lab": "lab_from_rgb", "rgb_from_luv": "luv_from_rgb", "rgb_from_xyz": "xyz_from_rgb",
               "rgb_from_linear_rgb": "linear_rgb_from_rgb"}[name]
        src = O.cie(fwd, src)  # inputs in the inverse conversion's own domain
    d_src, d_dst = dev(gpu_stream, src), out_buf(gpu_stream, src.nbytes)
    _ffi.check(_ffi.lib.kh_cie_convert_f32(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, src.size // 3, O.CIE[name]))
    got = d_dst.to_numpy(np.float32, src.shape)
    want32 = O.cie(name, src)
    # Bit-identical for every conversion WHERE THE BOX'S LIBM IS THE RESTATED ONE: the device evaluates glibc 2.35's powf / cbrtf
    # (csrc/kh_libm_glibc.h), the functions the restatement calls here, instead of the device library's.  Parity must not depend on
    # the test box's libm release (another glibc, an FMA-dispatched powf, musl, aarch64): when the host check below reports any
    # difference, the bound is round 2's 2e-4 + 2e-5 |v| — far inside the reference's own CIE tolerances (P/color/cie/mod.rs:150-341).
    if libm_agrees:
        assert_same_bits(got, want32, name)
    else:
        assert np.all(np.abs(got.astype(np.float64) - want32) <= 2e-4 + 2e-5 * np.abs(want32)), name
    want64 = O.cie(name, src.astype(np.float64))
    tol = np.array(TOL.get(name, [1e-3] * 3))
    assert np.all(np.abs(got.astype(np.float64) - want64).reshape(-1, 3).max(axis=0) <= tol)
    assert _ffi.lib.kh_cie_convert_f32(gpu_stream.cuda_stream_ptr, d_src.ptr, d_dst.ptr, 4, 99) == _ffi.KH_ERR_INVALID_ARG


@pytest.mark.gpu
def test_device_cie_host_api_round_trip(gpu_stream):
    from kornia_rs import Image, imgproc
    f32v, _ = pair(64, 48)
    img = Image.from_numpy(f32v).to_hip(gpu_stream)
    for fwd, rev in PAIRS:
        back = getattr(imgproc, rev)(getattr(imgproc, fwd)(img)).cpu().numpy()
        mask = np.ones(f32v.shape[:2], bool) if fwd != "luv_from_rgb" else f32v.sum(axis=2) > 0.05
        assert np.abs(back - f32v)[mask].max() <= 1e-3, fwd


# ---- the restated libm functions the device evaluates (round 3) -------------------------------------------------------------
@pytest.fixture(scope="module")
def libm_host(tmp_path_factory):
    import ctypes as C
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    out = tmp_path_factory.mktemp("libm") / "liblibm_glibc_host.so"
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fno-builtin", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror",
           f"-I{root / 'kornia-rs_amd' / 'csrc'}", str(root / "tests" / "cpp" / "libm_glibc_host.cpp"), "-o", str(out), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = C.CDLL(str(out))
    lib.host_check_powf.restype = C.c_long
    lib.host_check_powf.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.host_check_cbrtf.restype = C.c_long
    lib.host_check_cbrtf.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.host_powf_in_domain.argtypes = [C.c_float, C.c_float]
    return lib


def _libm_mismatches(lib) -> int:
    """How many of the sampled arguments the restated powf / cbrtf and THIS box's libm disagree on (0 on glibc 2.35 x86-64)."""
    import ctypes as C
    first = C.c_uint32(0)
    bad = 0
    for y in (2.4, 1.0 / 2.4, 3.0):
        bad += lib.host_check_powf(y, 0x30800000, 0x4E800001, 1009, C.byref(first))
    bad += lib.host_check_cbrtf(0x00000001, 0x7F800000, 4093, C.byref(first))
    return int(bad)


@pytest.fixture(scope="module")
def libm_agrees(libm_host):
    return _libm_mismatches(libm_host) == 0


def test_restated_powf_and_cbrtf_equal_this_boxes_libm(libm_host):
    """csrc/kh_libm_glibc.h (generated by scripts/gen_libm_tables.py from glibc 2.35's tables and algorithm) is what the device
    evaluates for the sRGB transfer and the Lab / Luv cube roots.  Here it is built for the host and compared with the libm the
    restatement (and the reference's f32::powf / f32::cbrt) use on this box, bit for bit: ~3.5 M arguments per exponent over the
    whole admitted domain, ~4.4 M cube roots over every binade incl. subnormals and negatives."""
    import ctypes as C
    import platform
    if _libm_mismatches(libm_host) and platform.libc_ver() != ("glibc", "2.35"):
        pytest.skip(f"this box's libm is {platform.libc_ver()}, not the glibc 2.35 the tables restate: the device tests use the tolerance bound here")
    first = C.c_uint32(0)
    for y in (2.4, 1.0 / 2.4, 3.0, 0.37, -1.7):
        bad = libm_host.host_check_powf(y, 0x30800000, 0x4E800001, 143, C.byref(first))
        assert bad == 0, f"powf(x, {y}): {bad} mismatches, first at x bits {first.value:#010x}"
    for lo, hi in ((0x00000001, 0x7F800000), (0x80000001, 0xFF800000)):
        bad = libm_host.host_check_cbrtf(lo, hi, 977, C.byref(first))
        assert bad == 0, f"cbrtf: {bad} mismatches, first at bits {first.value:#010x}"
    assert libm_host.host_check_cbrtf(0x7F800000, 0x7F800002, 1, C.byref(first)) == 0   # +inf, NaN
    assert libm_host.host_check_cbrtf(0, 1, 1, C.byref(first)) == 0                        # +0
    # outside the admitted domain the product takes the generic libm call
    for x, y, ok in ((1.0, 2.4, 1), (0.0, 2.4, 0), (1e-30, 2.4, 0), (float("inf"), 2.4, 0), (float("nan"), 2.4, 0), (2.0, 5.0, 0), (1e9, 0.5, 1), (2e9, 0.5, 0)):
        assert libm_host.host_powf_in_domain(x, y) == ok, (x, y)
