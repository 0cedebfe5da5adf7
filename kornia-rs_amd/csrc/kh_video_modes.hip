// YUYV -> RGB8 mode decode for gfx950 — replaces the three `yuyv_to_rgb_{bt601_full,bt709_full,bt601_limited}_u8`
// launchers (crates/kornia-imgproc/src/cuda/color/video.rs:128-190) == convert_yuyv_to_rgb_u8
// (crates/kornia-imgproc/src/color/yuv/mod.rs:342-410).  HBM-bound map: a thread owns one pixel pair = one dword in,
// six bytes out; a wave reads 256 contiguous bytes and writes 384.  As in the reference an odd width leaves the last
// pixel of every row untouched (the row is walked in whole 6-byte RGB chunks, :374-376).
#include "kh_common.h"

#include "kh_video_modes.h"

using namespace kh;

namespace {

__global__ __launch_bounds__(kBlock) void yuyv_mode_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int width,
                                                           int pairs_per_row, long long npairs, int mode) {
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= npairs) return;
    const long long row = p / pairs_per_row;
    const int col = (int)(p - row * pairs_per_row);
    const uint32_t q = *reinterpret_cast<const u32_unaligned*>(src + row * 2 * width + 4 * col);  // Y0 U Y1 V
    const int y0 = q & 0xff, u = (q >> 8) & 0xff, y1 = (q >> 16) & 0xff, v = q >> 24;
    const uint32_t a = kh_vm::rgb_from_yuv(mode, y0, u, v), b = kh_vm::rgb_from_yuv(mode, y1, u, v);
    uint8_t* o = dst + row * 3 * width + 6 * col;
    *reinterpret_cast<u32_unaligned*>(o) = a | (b << 24);        // R0 G0 B0 R1
    *reinterpret_cast<u16_unaligned*>(o + 4) = (uint16_t)(b >> 8);  // G1 B1
}

}  // namespace

extern "C" int32_t kh_yuyv_to_rgb_mode_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t width, int32_t height,
                                          int32_t mode) {
    const char* what = "kh_yuyv_to_rgb_mode_u8";
    KH_REQUIRE(mode >= 0 && mode < kh_vm::kModes, KH_ERR_INVALID_ARG, "%s: unknown mode %d", what, mode);
    KH_REQUIRE(width >= 0 && height >= 0, KH_ERR_INVALID_ARG, "%s: negative size %dx%d", what, width, height);
    KH_REQUIRE((int64_t)width * height * 3 <= kI32Max, KH_ERR_TOO_LARGE, "%s: image exceeds 32-bit indexing", what);
    const int pairs_per_row = width / 2;
    const long long npairs = (long long)pairs_per_row * height;
    if (npairs == 0) return KH_OK;
    KH_REQUIRE(src && dst, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    hipLaunchKernelGGL(yuyv_mode_kernel, dim3(cdiv(npairs, kBlock)), dim3(kBlock), 0, as_hip(stream), src, dst, (int)width,
                       pairs_per_row, npairs, (int)mode);
    return check_launch(what);
}
