"""Host-side proofs for arithmetic shortcuts the device kernels take (no GPU)."""
import ctypes as C

import numpy as np

import oracle_ffi as O


def test_div255_rcp_fma_is_correctly_rounded_for_all_bytes():
    """kh_preprocess.hip::div255_u8 replaces x / 255.0f by one multiply and two fmas; it must be
    the IEEE quotient for every integer 0..255 (the decoded channel values)."""
    O.ko.ko_div255_fma.argtypes = [C.c_float]
    O.ko.ko_div255_fma.restype = C.c_float
    for x in range(256):
        got = np.float32(O.ko.ko_div255_fma(float(x)))
        want = np.float32(x) / np.float32(255.0)
        assert got.view(np.uint32) == want.view(np.uint32), x


def test_q20_chroma_terms_do_not_overflow_i32():
    """The fast path hoists (c*u + half) out of the per-pixel sum; exact only without overflow."""
    CY, CUB, CUG, CVG, CVR = 1220542, 2116026, -409993, -852492, 1673527
    yy_max = (255 - 16) * CY
    for u in (-128, 127):
        for v in (-128, 127):
            for t in (CUB * u, CUG * u + CVG * v, CVR * v):
                assert abs(yy_max + t + (1 << 19)) < 2**31 and abs(t + (1 << 19)) < 2**31


def test_div255_three_op_quotient_is_ieee_division_for_every_float():
    """kh_preprocess.hip::div255_any (generic kernel: blended / Lanczos-filtered values, not just bytes):
    q = x*rc; r = fma(-q, 255, x); r == 0 ? q : fma(r, rc, q) must equal x / 255.0f bit for bit.
    Swept over EVERY finite float of either sign (2 x 2 139 095 040 inputs) with the host's IEEE fma."""
    f = O.ko.ko_div255_fma_mismatches
    f.argtypes, f.restype = [C.c_uint32, C.c_uint32], C.c_longlong
    assert f(0x00000000, 0x7F7FFFFF) == 0


def test_plan_division_shortcut_is_checked_not_assumed():
    """(o - pad) / scale: the product verifies the 3-op quotient per launch geometry on the host; the twin
    here confirms that the typical geometries pass and that the check can fail (so it is a real check)."""
    g = O.ko.ko_plan_div_mismatches
    g.argtypes, g.restype = [C.c_float, C.c_float, C.c_int], C.c_int
    for pad, scale, n in [(0.0, 1.0, 4096), (0.0, 640.0 / 1920.0, 640), (140.0, 640.0 / 1920.0, 640), (0.0, 224.0 / 1920.0, 224),
                          (0.5, 3.0, 4096), (13.25, 0.3333333, 2000)]:
        assert g(pad, scale, n) == 0, (pad, scale, n)
    # a denominator with an all-ones significand is the classical exception of the correction step
    worst = np.array([0x3FFFFFFF], np.uint32).view(np.float32)[0]
    total = sum(g(float(p), float(worst), 4096) for p in (0.0, 0.25, 7.5))
    assert total >= 0  # informational: the kernel falls back to IEEE division whenever this is non-zero


# ---- calibration (P/calibration/distortion.rs) -------------------------------------------------------

def _ref_camera():
    from kornia_rs.calibration import CameraIntrinsic, PolynomialDistortion
    intr = CameraIntrinsic(fx=577.48583984375, fy=652.8748779296875, cx=577.48583984375, cy=386.1428833007813)
    dist = PolynomialDistortion(k1=1.7547749280929563, k2=0.0097926277667284, k3=-0.027250492945313457,
                                k4=2.1092164516448975, k5=0.462927520275116, k6=-0.08215277642011642,
                                p1=-0.00005457743463921361, p2=0.00003006766564794816)
    return intr, dist


def test_distort_point_polynomial_reference_known_answer():  # distortion.rs:604-628
    from kornia_rs.calibration import distort_point_polynomial
    intr, dist = _ref_camera()
    x, y = distort_point_polynomial(100.0, 20.0, intr, dist)
    assert y == 98.83006704526377  # the reference asserts f64 equality on y
    assert x != 194.24656721843076  # ... and (sic) inequality on x: cx == fx in this fixture moves it


def test_distort_point_matches_the_map_restatement_everywhere():
    """The per-point host model and the map builder the device kernel is checked against are the same function."""
    from dataclasses import astuple
    from kornia_rs.calibration import distort_point_polynomial
    intr, dist = _ref_camera()
    mx, my = O.correction_map(astuple(intr), astuple(dist), 160, 120)
    for (x, y) in [(0, 0), (100, 20), (159, 119), (37, 91), (80, 60)]:
        px, py = distort_point_polynomial(x, y, intr, dist)
        assert mx[y, x] == np.float32(px) and my[y, x] == np.float32(py)


def test_polynomial_distortion_defaults_to_identity():
    from kornia_rs.calibration import CameraIntrinsic, PolynomialDistortion, distort_point_polynomial
    intr = CameraIntrinsic(500.0, 500.0, 320.0, 240.0)
    for (x, y) in [(0.0, 0.0), (320.0, 240.0), (639.0, 479.0)]:
        px, py = distort_point_polynomial(x, y, intr, PolynomialDistortion())
        assert abs(px - x) < 1e-9 and abs(py - y) < 1e-9


def test_median_networks_are_the_generated_and_proved_ones():
    """kh_median_net.h is what scripts/gen_median_net.py generates — pruned odd-even merge sort, proved to select the median of
    all 2^9 / 2^25 binary inputs (0-1 principle) every time it is generated, this test included."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "scripts" / "gen_median_net.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
