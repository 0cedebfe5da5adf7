// YUYV -> RGB8 with a selectable matrix (convert_yuyv_to_rgb_u8, P/color/yuv/mod.rs:319-480): Q10 integer arithmetic,
// one (U, V) per horizontal pixel pair.  Shared by the device kernel (kh_video_modes.hip) and a host harness
// (tests/cpp/video_modes_host.cpp) that sweeps all 2^24 (Y, U, V) triples per mode on a CPU.
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define KH_VM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define KH_VM_HD inline
#endif

namespace kh_vm {

enum { kBt601Full = 0, kBt709Full = 1, kBt601Limited = 2, kModes = 3 };

KH_VM_HD int clamp_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// `>>` on the signed intermediates is arithmetic, like Rust's i32 shift
KH_VM_HD uint32_t rgb_from_yuv(int mode, int y, int u, int v) {
    u -= 128;
    v -= 128;
    int r, g, b;
    if (mode == kBt601Full) {            // :417-438
        r = y + ((1436 * v + 512) >> 10);
        g = y - ((352 * u + 731 * v + 512) >> 10);
        b = y + ((1815 * u + 512) >> 10);
    } else if (mode == kBt709Full) {     // :441-461
        r = y + ((1612 * v + 512) >> 10);
        g = y - ((192 * u + 479 * v + 512) >> 10);
        b = y + ((1900 * u + 512) >> 10);
    } else {                             // Bt601Limited, :464-480
        const int ys = ((y - 16) * 1192 + 512) >> 10;
        r = ys + ((1634 * v + 512) >> 10);
        g = ys - ((401 * u + 832 * v + 512) >> 10);
        b = ys + ((2066 * u + 512) >> 10);
    }
    return (uint32_t)clamp_u8(r) | ((uint32_t)clamp_u8(g) << 8) | ((uint32_t)clamp_u8(b) << 16);  // packed R, G, B
}

}  // namespace kh_vm
