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
