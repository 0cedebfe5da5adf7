// Host harness for tests/test_cie.py: the PRODUCT's restatement of glibc powf / cbrtf (csrc/kh_libm_glibc.h, the file the gfx950
// kernels compile) against the libm of this box, bit for bit.  Returns the number of mismatches; fills `first` with the first one.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "kh_libm_glibc.h"

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

extern "C" long host_check_powf(float y, uint32_t from_bits, uint32_t to_bits, uint32_t step, uint32_t* first) {
    long bad = 0;
    for (uint64_t b = from_bits; b < to_bits; b += step) {
        float x; const uint32_t bb = (uint32_t)b; memcpy(&x, &bb, 4);
        if (!kh_libm::powf_in_domain(x, y)) continue;
        if (bits(kh_libm::powf_glibc(x, y)) != bits(powf(x, y))) { if (!bad && first) *first = bb; ++bad; }
    }
    return bad;
}
extern "C" long host_check_cbrtf(uint32_t from_bits, uint32_t to_bits, uint32_t step, uint32_t* first) {
    long bad = 0;
    for (uint64_t b = from_bits; b < to_bits; b += step) {
        float x; const uint32_t bb = (uint32_t)b; memcpy(&x, &bb, 4);
        const float a = kh_libm::cbrtf_glibc(x), w = cbrtf(x);
        if (bits(a) != bits(w) && !(a != a && w != w)) { if (!bad && first) *first = bb; ++bad; }
    }
    return bad;
}
extern "C" int host_powf_in_domain(float x, float y) { return kh_libm::powf_in_domain(x, y) ? 1 : 0; }
