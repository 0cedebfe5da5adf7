// Bayer mosaic -> RGB8 (bilinear, cv2-compatible) for gfx950 — replaces launch_rgb_from_bayer_u8
// (crates/kornia-imgproc/src/cuda/color/bayer.rs) == rgb_from_bayer (crates/kornia-imgproc/src/color/bayer/mod.rs:37-70,
// kernels.rs:30-200): rounded integer averages avg2 = (a+b+1)>>1, avg4 = (a+b+c+d+2)>>2 over replicate-clamped neighbours,
// then cv2's border rule — the 1-pixel frame takes its interior neighbour's result (rows first, then columns, so a corner
// equals pixel (1,1)) when the image has an interior in that direction (>= 3 rows / columns).
//
// HBM-bound map, 1 B in / 3 B out per pixel: one thread per pixel, 64 x 4 tiles so a wave reads three 66-byte row segments
// that its neighbours share through L1/L2 and writes 192 contiguous bytes.  The frame rule is a coordinate remap
// (output (r, c) = the demosaic of pixel (r', c') with r' = clamp(r, 1, rows - 2) ...), so there is one pass and no patch-up.
#include "kh_common.h"

using namespace kh;

namespace {

constexpr int kBx = 64, kBy = 4;

// Cell kinds: 0 = R, 1 = G on an R row, 2 = G on a B row, 3 = B; table[pattern][row & 1][col & 1] (kernels.rs:30-43)
__device__ __forceinline__ int cell_kind(int pattern, int r, int c) {
    // packed 2-bit entries, [row & 1][col & 1] -> bits 4 * (r & 1) + 2 * (c & 1)
    //            RGGB: [[R, GonR], [GonB, B]]   BGGR: [[B, GonB], [GonR, R]]   GRBG: [[GonR, R], [B, GonB]]   GBRG: [[GonB, B], [R, GonR]]
    const unsigned tables[4] = {0u | (1u << 2) | (2u << 4) | (3u << 6), 3u | (2u << 2) | (1u << 4) | (0u << 6),
                                1u | (0u << 2) | (3u << 4) | (2u << 6), 2u | (3u << 2) | (0u << 4) | (1u << 6)};
    return (int)((tables[pattern] >> (4 * (r & 1) + 2 * (c & 1))) & 3u);
}

__global__ __launch_bounds__(kBx* kBy) void bayer_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int cols,
                                                         int pattern) {
    const int c_out = blockIdx.x * kBx + threadIdx.x, r_out = blockIdx.y * kBy + threadIdx.y;
    if (c_out >= cols || r_out >= rows) return;
    const int r = rows >= 3 ? min(max(r_out, 1), rows - 2) : r_out;  // bayer_border_replicate, kernels.rs:175-200
    const int c = cols >= 3 ? min(max(c_out, 1), cols - 2) : c_out;
    const int rn = max(r - 1, 0), rs = min(r + 1, rows - 1), cw = max(c - 1, 0), ce = min(c + 1, cols - 1);  // replicate addressing
    const uint8_t* up = src + (long long)rn * cols;
    const uint8_t* mid = src + (long long)r * cols;
    const uint8_t* dn = src + (long long)rs * cols;
    const unsigned center = mid[c];
    const unsigned n = up[c], s = dn[c], w = mid[cw], e = mid[ce];
    const unsigned cross = (n + s + w + e + 2u) >> 2;
    unsigned red, green, blue;
    switch (cell_kind(pattern, r, c)) {
        case 0: {  // R: G from the cross, B from the diagonals
            const unsigned diag = ((unsigned)up[cw] + up[ce] + dn[cw] + dn[ce] + 2u) >> 2;
            red = center; green = cross; blue = diag;
            break;
        }
        case 3: {  // B
            const unsigned diag = ((unsigned)up[cw] + up[ce] + dn[cw] + dn[ce] + 2u) >> 2;
            red = diag; green = cross; blue = center;
            break;
        }
        case 1:  // G on an R row: R left / right, B above / below
            red = (w + e + 1u) >> 1; green = center; blue = (n + s + 1u) >> 1;
            break;
        default:  // G on a B row
            red = (n + s + 1u) >> 1; green = center; blue = (w + e + 1u) >> 1;
            break;
    }
    uint8_t* o = dst + ((long long)r_out * cols + c_out) * 3;
    o[0] = (uint8_t)red; o[1] = (uint8_t)green; o[2] = (uint8_t)blue;
}

}  // namespace

extern "C" int32_t kh_rgb_from_bayer_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t width, int32_t height, int32_t pattern) {
    const char* what = "kh_rgb_from_bayer_u8";
    KH_REQUIRE(pattern >= KH_BAYER_RGGB && pattern <= KH_BAYER_GBRG, KH_ERR_INVALID_ARG, "%s: unknown pattern %d", what, pattern);
    KH_REQUIRE(width >= 0 && height >= 0, KH_ERR_INVALID_ARG, "%s: negative size %dx%d", what, width, height);
    KH_REQUIRE((int64_t)width * height * 3 <= kI32Max, KH_ERR_TOO_LARGE, "%s: image exceeds 32-bit indexing", what);
    if ((int64_t)width * height == 0) return KH_OK;
    KH_REQUIRE(src && dst, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    KH_REQUIRE(cdiv(height, kBy) <= 65535u, KH_ERR_TOO_LARGE, "%s: %d rows exceed one launch", what, height);
    hipLaunchKernelGGL(bayer_kernel, dim3(cdiv(width, kBx), cdiv(height, kBy)), dim3(kBx, kBy), 0, as_hip(stream), src, dst, (int)height,
                       (int)width, (int)pattern);
    return check_launch(what);
}
