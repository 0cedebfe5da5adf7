// f32 geometric resampling for gfx950: resize, warp_affine, warp_perspective, remap, and the
// Brown-Conrady undistortion maps.
//
// Device twins of P/cuda/{resize,warp_affine,warp_perspective,remap}.rs; arithmetic follows the
// reference CPU paths (P/resize/mod.rs:134-238, P/warp/affine.rs:123-372,
// P/warp/perspective.rs:115-166, P/interpolation/{bilinear,nearest,bicubic,remap}.rs,
// P/calibration/distortion.rs:68-152) with identical expression trees, uncontracted f32
// (-ffp-contract=off), IEEE division, and fmaf exactly where the reference writes mul_add — so
// outputs are bit-identical to the CPU reference (asserted by tests/test_geom_gpu.py).
//
// All of these are gathers: one thread per destination pixel (all channels), 64x4 blocks so a
// wave owns 64 consecutive pixels of one output row — 768 B contiguous stores for C = 3 and
// source taps confined to two (bilinear) or four (bicubic) source rows.  grid.z = image in the
// batch (the reference launches once per image).
#include <math.h>

#include <algorithm>

#include "kh_common.h"

using namespace kh;

namespace {

constexpr int kBx = 64, kBy = 4;
constexpr float kPxMaxRows = 8.0f;   // two-pixels-per-lane warps: the most source rows a 128-pixel destination run may cross

struct Img {  // one batch of same-sized HWC f32 images
    const float* src;
    float* dst;
    int sw, sh, dw, dh;
    long long src_stride, dst_stride;  // elements between consecutive images
    XcdTiles tiles;                    // kBx x kBy output tiles, XCD-contiguous order
};

// Pointer-list launches (kh_*_f32_list, kh_common.h::PtrList) are separate instantiations (LIST = true) of the same kernels: the image's
// (src, dst) pair is one scalar load from the kernel arguments where an equally spaced batch computes base + k * stride.  Measured in
// round 6, same box, C5 shape (profiles/r06c, r06d, r06h): selecting between the two at RUN time inside one kernel cost the equally
// spaced launches 8 % (C5 20.4-20.8 vs 18.7-19.1 ms) — so LIST = false is the round-5 kernel, argument for argument; a 2-D grid with the
// image in blockIdx.y (so that the pair's load does not wait for the tile decode), with or without pinning the loaded pointers in
// the entry block, ran the list kernels SLOWER than the flattened XCD grid used here (remap 5.69 vs 5.35 ms, warp_perspective 4.88 vs
// 4.59 ms per 128 4K images; equally spaced 4.70 / 4.69 ms).
struct NoList { int unused; };
template <bool LIST> struct ListArg { typedef NoList type; };
template <> struct ListArg<true> { typedef PtrList type; };

// ---- samplers (expression trees of P/interpolation/*.rs; do not regroup) --------------------------

// bilinear_interpolation (P/interpolation/bilinear.rs:16-66): trunc, edge taps replicate val00
template <int C>
__device__ __forceinline__ void sample_bilinear(const float* __restrict__ img, int rows, int cols, float u,
                                                float v, float out[C]) {
    const int iu = (int)u, iv = (int)v;
    const float frac_u = u - truncf(u), frac_v = v - truncf(v);
    const float* p00 = img + ((long long)iv * cols + iu) * C;
    const bool hx = iu + 1 < cols, hy = iv + 1 < rows;
    const float* p01 = hx ? p00 + C : p00;
    const float* p10 = hy ? p00 + (long long)cols * C : p00;
    const float* p11 = (hx && hy) ? p00 + (long long)cols * C + C : p00;
    const float frac_uu = 1.0f - frac_u, frac_vv = 1.0f - frac_v;
    const float w00 = frac_vv * frac_uu, w10 = frac_vv * frac_u, w01 = frac_v * frac_uu, w11 = frac_v * frac_u;
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = w00 * p00[c] + w10 * p01[c] + w01 * p10[c] + w11 * p11[c];
}

// nearest_neighbor_interpolation (P/interpolation/nearest.rs:15-30)
template <int C>
__device__ __forceinline__ void sample_nearest(const float* __restrict__ img, int rows, int cols, float u,
                                               float v, float out[C]) {
    const float ru = roundf(u), rv = roundf(v);
    long long iu = ru > 0.0f ? (long long)ru : 0, iv = rv > 0.0f ? (long long)rv : 0;
    iu = min(iu, (long long)cols - 1);
    iv = min(iv, (long long)rows - 1);
    const float* p = img + (iv * cols + iu) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = p[c];
}

// keys_weights + bicubic_sample (P/interpolation/bicubic.rs:15-62)
__device__ __forceinline__ void keys_weights(float frac, float w[4]) {
    float t;
    t = 1.0f + frac; w[0] = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-0.5f, t, 2.5f), t, -4.0f), t, 2.0f);
    t = frac;        w[1] = __builtin_fmaf(__builtin_fmaf(1.5f, t, -2.5f) * t, t, 1.0f);
    t = 1.0f - frac; w[2] = __builtin_fmaf(__builtin_fmaf(1.5f, t, -2.5f) * t, t, 1.0f);
    t = 2.0f - frac; w[3] = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-0.5f, t, 2.5f), t, -4.0f), t, 2.0f);
}
template <int C>
__device__ __forceinline__ void sample_bicubic(const float* __restrict__ img, int rows, int cols, float sx,
                                               float sy, float out[C]) {
    const float x0f = floorf(sx), y0f = floorf(sy);
    float wx[4], wy[4];
    keys_weights(sx - x0f, wx);
    keys_weights(sy - y0f, wy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
        const int yi = min(max(y0 + dy - 1, 0), rows - 1);
        const float* row = img + (long long)yi * cols * C;
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int xi = min(max(x0 + dx - 1, 0), cols - 1);
            const float w = wx[dx] * wy[dy];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = __builtin_fmaf(w, row[(long long)xi * C + c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = acc[c];
}

// Lanczos-3 (P/interpolation/lanczos.rs).  sin_pi is the reference's libm-free polynomial (:19-37),
// plain mul/add; lanczos3 (:40-51) feeds the resize tables, lanczos3_weights (:107-141) the warps.
constexpr float kPi = 3.14159265358979323846f;
__host__ __device__ __forceinline__ float sin_pi(float x) {
    const float k = roundf(x);
    const float r = x - k;
    const float z = kPi * r;
    const float z2 = z * z;
    float p = -2.5052108e-8f;
    p = p * z2 + 2.7557319e-6f;
    p = p * z2 + -1.984127e-4f;
    p = p * z2 + 8.333334e-3f;
    p = p * z2 + -1.6666667e-1f;
    const float s = z + z * z2 * p;
    return ((int)k & 1) ? -s : s;
}
__host__ __device__ __forceinline__ float lanczos3(float x) {
    if (fabsf(x) < 1e-5f) return 1.0f;
    if (fabsf(x) >= 3.0f) return 0.0f;
    const float pix = kPi * x;
    const float pix3 = pix * 0.33333334f;
    return sin_pi(x) * sin_pi(x * (1.0f / 3.0f)) / (pix * pix3);
}
__device__ __forceinline__ float lanczos_den(float x) {
    const float pix = kPi * x;
    const float pix3 = pix * 0.33333334f;
    return pix * pix3;
}
__device__ __forceinline__ void lanczos3_weights(float frac, float w[6]) {
    const float s = sin_pi(frac);
    const float t0 = sin_pi(frac * (1.0f / 3.0f));
    const float t1 = sin_pi((frac - 1.0f) * (1.0f / 3.0f));
    const float t2 = sin_pi((frac - 2.0f) * (1.0f / 3.0f));
    const float st0 = s * t0, st1 = s * t1, st2 = s * t2;
    w[0] = -st1 / lanczos_den(frac + 2.0f);
    w[1] = st2 / lanczos_den(frac + 1.0f);
    w[2] = st0 / lanczos_den(frac);
    w[3] = -st1 / lanczos_den(frac - 1.0f);
    w[4] = st2 / lanczos_den(frac - 2.0f);
    w[5] = st0 / lanczos_den(frac - 3.0f);
    if (frac < 1e-5f) w[2] = 1.0f;
    if (fabsf(frac - 1.0f) < 1e-5f) w[3] = 1.0f;
}
// lanczos_sample (:143-187): per-axis normalisation, per-row fmaf chain, then fmaf by wy
template <int C>
__device__ __forceinline__ void sample_lanczos(const float* __restrict__ img, int rows, int cols, float sx,
                                               float sy, float out[C]) {
    const float x0f = floorf(sx), y0f = floorf(sy);
    float wx[6], wy[6];
    lanczos3_weights(sx - x0f, wx);
    lanczos3_weights(sy - y0f, wy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float sum_wx = wx[0] + wx[1] + wx[2] + wx[3] + wx[4] + wx[5];
    const float sum_wy = wy[0] + wy[1] + wy[2] + wy[3] + wy[4] + wy[5];
    const float inv_x = 1.0f / sum_wx, inv_y = 1.0f / sum_wy;
#pragma unroll
    for (int t = 0; t < 6; ++t) { wx[t] *= inv_x; wy[t] *= inv_y; }
    int xo[6];
#pragma unroll
    for (int dx = 0; dx < 6; ++dx) xo[dx] = min(max(x0 + dx - 2, 0), cols - 1) * C;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 6; ++dy) {
        const int yi = min(max(y0 + dy - 2, 0), rows - 1);
        const float* row = img + (long long)yi * cols * C;
        float rx[C];
#pragma unroll
        for (int c = 0; c < C; ++c) rx[c] = 0.0f;
#pragma unroll
        for (int dx = 0; dx < 6; ++dx) {
#pragma unroll
            for (int c = 0; c < C; ++c) rx[c] = __builtin_fmaf(wx[dx], row[xo[dx] + c], rx[c]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = __builtin_fmaf(wy[dy], rx[c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = acc[c];
}

template <int C, int MODE>
__device__ __forceinline__ void sample(const float* __restrict__ img, int rows, int cols, float u, float v,
                                       float out[C]) {
    if constexpr (MODE == KH_INTERP_NEAREST) sample_nearest<C>(img, rows, cols, u, v, out);
    else if constexpr (MODE == KH_INTERP_BILINEAR) sample_bilinear<C>(img, rows, cols, u, v, out);
    else if constexpr (MODE == KH_INTERP_BICUBIC) sample_bicubic<C>(img, rows, cols, u, v, out);
    else sample_lanczos<C>(img, rows, cols, u, v, out);
}

// Output pixels leave through the streaming store path (write-through, non-temporal: kh_common.h::stream_store) — the destination is
// written once and never read by the kernel.  The V# covers what is left of the pixel's ROW: a wave is one row of a 64 x 4 tile
// (kBx == 64), so the row base is wave-uniform and only x * C * 4 travels in the vector offset.
static_assert(kBx == 64, "out_row: a wave must be exactly one tile row");
struct OutRow { __amdgpu_buffer_rsrc_t rs; };
template <int C>
__device__ __forceinline__ OutRow out_row(float* row_base_of_this_wave, int dw) {
    const uint64_t p = reinterpret_cast<uint64_t>(row_base_of_this_wave);
    const uint64_t u = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)p) |   // (the builtin returns a signed int)
                       ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(p >> 32)) << 32);
    return OutRow{stream_window(reinterpret_cast<const void*>(u), (long long)dw * C * 4)};
}
template <int C>
__device__ __forceinline__ void put(const OutRow& r, int x, const float v[C]) {
    uint32_t w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = __float_as_uint(v[c]);
    stream_store<C>(r.rs, x * C * 4, w);
}
template <int C>
__device__ __forceinline__ void put_zero(const OutRow& r, int x) {
    uint32_t w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = 0u;
    stream_store<C>(r.rs, x * C * 4, w);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// The bases of image `k` of the launch.
template <bool LIST>
__device__ __forceinline__ const float* image_src(const Img& im, const typename ListArg<LIST>::type& lst, unsigned k) {
    if constexpr (LIST) return static_cast<const float*>(lst.src[k]);
    else return im.src + (long long)k * im.src_stride;
}
template <bool LIST>
__device__ __forceinline__ float* image_dst(const Img& im, const typename ListArg<LIST>::type& lst, unsigned k) {
    if constexpr (LIST) return static_cast<float*>(lst.dst[k]);
    else return im.dst + (long long)k * im.dst_stride;
}

#define KH_PIXEL_PROLOGUE                                             \
    unsigned bx_, by_, bz_;                                           \
    if (!xcd_tile(im.tiles, bx_, by_, bz_)) return;                   \
    const int x = bx_ * kBx + threadIdx.x;                            \
    const int y = by_ * kBy + threadIdx.y;                            \
    if (x >= im.dw || y >= im.dh) return;                             \
    const float* src = image_src<LIST>(im, lst, bz_);                 \
    const OutRow o = out_row<C>(image_dst<LIST>(im, lst, bz_) + (long long)y * im.dw * C, im.dw);

// resize (P/resize/mod.rs:161-176): half-pixel grid a*x + b, clamped to the source.
// Bound by the texture addresser on moderate scales (1080p -> 540p bicubic: sixteen 12-byte gathers per pixel, TA_BUSY 100 %,
// profiles/r04zk_resize_counters.csv).  An LDS-staged twin (64 x 8 tiles, the box copied with coalesced 16-byte loads, TA_BUSY 33 %)
// was built, bit-identical, and measured 9 % SLOWER (1.85 vs 1.69 ms, profiles/r04zl_*): one box per block leaves the wave waiting
// 69 % of its cycles (load -> LDS -> barrier -> sample, nothing to overlap with) and costs 1.6x the vector instructions.  Not kept.
// Taking a row's four interior taps as three 16-byte loads instead of four 12-byte ones: no change (r04zm): the addresser's cost
// follows the bytes, not the instruction count.
template <int C, int MODE, bool LIST>
__global__ __launch_bounds__(kBx* kBy) void resize_kernel(Img im, float ax, float bx, float ay, float by, typename ListArg<LIST>::type lst) {
    KH_PIXEL_PROLOGUE
    const float sx = clampf(ax * (float)x + bx, 0.0f, (float)(im.sw - 1));
    const float sy = clampf(ay * (float)y + by, 0.0f, (float)(im.sh - 1));
    float v[C];
    sample<C, MODE>(src, im.sh, im.sw, sx, sy, v);
    put<C>(o, x, v);
}

// One channel (round 6): resize_kernel gives a lane ONE float, so a wave stores 256 bytes per instruction and the per-row coordinate
// work is repeated for every pixel — gray f32 images (heat maps, flow fields) ran at 0.26-0.37 of peak on upscales against 0.65-0.9
// for RGB.  Four consecutive destination pixels per lane, the same sampler calls in the same order, one 16-byte streaming store
// (dw % 4 == 0, 16-byte-aligned destination images: host-checked).  A wave = one row of a 256 x 4 tile.
template <int MODE, bool LIST>
__global__ __launch_bounds__(kBx* kBy) void resize_quads1_kernel(Img im, float ax, float bx, float ay, float by, typename ListArg<LIST>::type lst) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(im.tiles, bx_, by_, bz_)) return;
    const int x0 = (bx_ * kBx + threadIdx.x) * 4;
    const int y = by_ * kBy + threadIdx.y;
    if (x0 >= im.dw || y >= im.dh) return;
    const float* src = image_src<LIST>(im, lst, bz_);
    const OutRow o = out_row<4>(image_dst<LIST>(im, lst, bz_) + (long long)y * im.dw, im.dw / 4);   // the row as dw / 4 four-float pixels
    const float sy = clampf(ay * (float)y + by, 0.0f, (float)(im.sh - 1));
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float sx = clampf(ax * (float)(x0 + j) + bx, 0.0f, (float)(im.sw - 1));
        sample<1, MODE>(src, im.sh, im.sw, sx, sy, &v[j]);
    }
    put<4>(o, x0 / 4, v);
}

// ---- bicubic, horizontal step exactly 2 (round 6; the reference's published 1080p -> 540p shape, benchmarks.md:374) ------------------------
// At sx = 2 x + 0.5 the four columns of output pixel x are source pixels 2x - 1 .. 2x + 2: a lane's window shares its outer columns
// with its neighbours' inner ones, and resize_kernel fetches every source pixel twice — 3 KiB of 12-byte gathers per wave and row for
// 1.5 KiB of source, with the texture addresser 100 % busy (profiles/r04zk_resize_counters.csv).  Here a lane loads ONLY its own two
// pixels of each of the four rows (24 contiguous bytes; a wave reads 1.5 KiB contiguous per row) and takes column 2x - 1 from the lane
// below and 2x + 2 from the lane above with wave shifts (v_mov_b32_dpp, kh_common.h::from_lane_below / above); the wave's first and
// last lanes load their outer column themselves.  Same expression as sample_bicubic (Keys weights at frac = sx - floor(sx) = 0.5,
// rows outer, columns inner, acc = fma(wx * wy, v, acc), taps clamped to the image), so the bits are resize_kernel's.
// VH: the vertical step is exactly 2 as well (sh == 2 dh): a lane owns TWO vertically adjacent outputs, whose windows share two of their
// four rows — six row loads for two outputs instead of eight (a 64 x 8 tile per block).  Each output's sixteen products are still added
// rows outer / columns inner in ascending order.
template <int C, bool VH, bool LIST>
__global__ __launch_bounds__(kBx* kBy) void resize_bicubic_half_kernel(Img im, float ax, float bx, float ay, float by, typename ListArg<LIST>::type lst) {
    constexpr int NY = VH ? 2 : 1, NR = VH ? 6 : 4;
    unsigned bx_, by_, bz_;
    if (!xcd_tile(im.tiles, bx_, by_, bz_)) return;
    const int lane = threadIdx.x;                        // a wave = one row (VH: two rows) of the tile
    const int x = bx_ * kBx + lane, ya = (by_ * kBy + threadIdx.y) * NY;
    if (ya >= im.dh) return;                             // wave-uniform
    const int xl = min(x, im.dw - 1);                    // lanes past the row keep computing (their neighbours read them) and store nothing
    const float* src = image_src<LIST>(im, lst, bz_);
    const float sx = clampf(ax * (float)xl + bx, 0.0f, (float)(im.sw - 1));   // == 2 xl + 0.5 (host-checked)
    const float x0f = floorf(sx);
    float wx[4], wy[NY][4];
    keys_weights(sx - x0f, wx);
    int y0[NY];
#pragma unroll
    for (int o = 0; o < NY; ++o) {
        const float sy = clampf(ay * (float)min(ya + o, im.dh - 1) + by, 0.0f, (float)(im.sh - 1));
        const float y0f = floorf(sy);
        keys_weights(sy - y0f, wy[o]);
        y0[o] = (int)y0f;                                // VH: y0[1] == y0[0] + 2 (host-checked: sy = 2 y + 0.5 unclamped)
    }
    const int x0 = (int)x0f;                             // == 2 xl
    const bool first = lane == 0, last = lane == kBx - 1;
    // the outer column this lane must fetch itself if it is an end of the wave: 2x - 1 (first lane) / 2x + 2 (last lane), clamped like a tap
    const int xh = first ? max(x0 - 1, 0) : min(x0 + 2, im.sw - 1);
    float acc[NY][C];
#pragma unroll
    for (int o = 0; o < NY; ++o)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[o][c] = 0.0f;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int yi = min(max(y0[0] + j - 1, 0), im.sh - 1);   // (output o's tap dy sits at j = dy + 2 o; the clamp is the tap's own)
        const float* row = src + (long long)yi * im.sw * C;
        float own[2 * C], halo[C];
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) own[k] = row[(long long)x0 * C + k];   // pixels x0, x0 + 1 (x0 + 1 <= sw - 1: sw == 2 dw)
#pragma unroll
        for (int c = 0; c < C; ++c) halo[c] = 0.0f;
        if (first || last) {
#pragma unroll
            for (int c = 0; c < C; ++c) halo[c] = row[(long long)xh * C + c];
        }
        float t[4][C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            // column x0 - 1 = the lane below's second pixel (clamped to 0 at the image's left edge = this lane's own first pixel)
            const float below = __uint_as_float(from_lane_below(__float_as_uint(own[C + c]), __float_as_uint(halo[c])));
            // column x0 + 2 = the lane above's first pixel (clamped to sw - 1 at the right edge = this lane's own second pixel)
            const float above = __uint_as_float(from_lane_above(__float_as_uint(own[c]), __float_as_uint(halo[c])));
            t[0][c] = x0 - 1 < 0 ? own[c] : below;
            t[1][c] = own[c];
            t[2][c] = own[C + c];
            t[3][c] = x0 + 2 > im.sw - 1 ? own[C + c] : above;
        }
#pragma unroll
        for (int o = 0; o < NY; ++o) {
            const int dy = j - 2 * o;
            if (dy < 0 || dy > 3) continue;              // compile-time after unrolling
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const float w = wx[dx] * wy[o][dy];
#pragma unroll
                for (int c = 0; c < C; ++c) acc[o][c] = __builtin_fmaf(w, t[dx][c], acc[o][c]);
            }
        }
    }
    if (x < im.dw) {
#pragma unroll
        for (int o = 0; o < NY; ++o) {
            if (ya + o >= im.dh) break;
            const OutRow orow = out_row<C>(image_dst<LIST>(im, lst, bz_) + (long long)(ya + o) * im.dw * C, im.dw);
            put<C>(orow, x, acc[o]);
        }
    }
}

// ---- bilinear downscale, row-streamed (round 6; BASELINE configs[1]) -------------------------------------------------------------------
// 1920 x 1080 -> 224 x 224: an output row needs exactly TWO source rows, and with a tap pair (24 bytes) every 103 bytes nearly every
// 128-byte line of both rows holds a tap byte — the gather kernel above already moved that line-granular floor (2.8 GB per 256 images
// against 0.77 GB of tap bytes), but as 896 scattered 12-byte requests per output row, at 5.3 TB/s (profiles/r05zzz_traffic.txt).  Here
// a 256-thread block owns ONE output row: it streams the two source rows it needs into LDS with 16-byte lane-contiguous loads — every
// load of the block requested before the first is waited for — and blends from LDS.  Same sampler expression as sample_bilinear
// (P/interpolation/bilinear.rs:16-66: trunc, edge taps replicate val00), so the bits are those of resize_kernel; the host routes a
// launch here when the rows are whole float4s on 16-byte-aligned images, a row pair fits LDS, the vertical step skips rows (>= 1.5:
// no row is shared by two output rows) and the horizontal tap stride is at most one line (every line is needed anyway).
typedef float f32x4g __attribute__((ext_vector_type(4)));
constexpr int kRowsBlockDefault = 256;
extern __shared__ __attribute__((aligned(16))) float lds_rows[];
// `split` blocks share an output row (each a run of output columns and the source-row segment under it): the LDS footprint of a block
// is what bounds the blocks per CU here, and a 1080p row pair is 46 KB.
__host__ __device__ __forceinline__ int rows_src_px(float ax, float bx, int x, int sw) {   // first tap column of output column x
    return (int)fminf(fmaxf(ax * (float)x + bx, 0.0f), (float)(sw - 1));
}
struct Norm3f { float mean[3], inv_std[3]; };
// NORM: the epilogue of resize_bilinear_normalize_3c (P/cuda/resize.rs:184-236), `(px - mean) * inv_std` per channel, on the blended value
template <int C, int ITER, int BLOCK, bool LIST, bool NORM>
__device__ __forceinline__ void resize_rows_body(const Img& im, float ax, float bx, float ay, float by, const FastDiv& by_rows, const FastDiv& by_parts,
                                                 int parts, int part_cols, int seg4_max, const typename ListArg<LIST>::type& lst, const Norm3f& nrm) {
    constexpr int kRowsBlock = BLOCK;
    const int t = threadIdx.x;
    // block -> (image z, output row y, column part): parts of `part_cols` output columns
    const unsigned rows_total = im.dh * (unsigned)parts;
    const unsigned z = fast_quot(blockIdx.x, by_rows), rem = blockIdx.x - z * rows_total;
    const unsigned y = fast_quot(rem, by_parts), part = rem - y * (unsigned)parts;
    const int x_lo = (int)part * part_cols, x_hi = min(x_lo + part_cols, im.dw);   // block-uniform
    const float* src = image_src<LIST>(im, lst, z);
    const float sy = clampf(ay * (float)y + by, 0.0f, (float)(im.sh - 1));
    const int iv = (int)sy;
    const float frac_v = sy - truncf(sy);
    const bool hy = iv + 1 < im.sh;                      // block-uniform
    // the float4 segment [s4, s4 + n4) of a source row that holds every tap of columns [x_lo, x_hi)
    const int n4row = (im.sw * C) >> 2;
    const int s4 = (rows_src_px(ax, bx, x_lo, im.sw) * C) >> 2;
    const int e4 = min(((min(rows_src_px(ax, bx, x_hi - 1, im.sw) + 1, im.sw - 1) + 1) * C + 3) >> 2, n4row);
    const int n4 = min(e4 - s4, seg4_max);              // (host-checked: e4 - s4 <= seg4_max <= ITER * 256)
    const f32x4g* g0 = reinterpret_cast<const f32x4g*>(src + (long long)iv * im.sw * C) + s4;
    const f32x4g* g1 = g0 + (hy ? n4row : 0);            // no row below: its taps replicate val00 and row 1 is never read; load row 0 twice
    f32x4g q0[ITER], q1[ITER];
#pragma unroll
    for (int j = 0; j < ITER; ++j) {                     // unconditional loads at clamped indices: 2 * ITER requests in flight per lane
        const int i = min(t + kRowsBlock * j, n4 - 1);
        q0[j] = g0[i];
        q1[j] = g1[i];
    }
    f32x4g* r0v = reinterpret_cast<f32x4g*>(lds_rows);
    f32x4g* r1v = r0v + seg4_max;
#pragma unroll
    for (int j = 0; j < ITER; ++j) {
        const int i = min(t + kRowsBlock * j, n4 - 1);   // (lanes past the segment rewrite its last float4 with the same value)
        r0v[i] = q0[j];
        r1v[i] = q1[j];
    }
    __syncthreads();
    const float* r0 = lds_rows;                          // segment-relative: float index of a row minus 4 * s4
    const float* r1 = r0 + 4 * seg4_max;
    const int base = 4 * s4;
    const __amdgpu_buffer_rsrc_t ow = stream_window(image_dst<LIST>(im, lst, z) + (long long)y * im.dw * C, (long long)im.dw * C * 4);
    const float frac_vv = 1.0f - frac_v;
    for (int x = x_lo + t; x < x_hi; x += kRowsBlock) {
        const float sx = clampf(ax * (float)x + bx, 0.0f, (float)(im.sw - 1));
        const int iu = (int)sx;
        const float frac_u = sx - truncf(sx);
        const bool hx = iu + 1 < im.sw;
        const float frac_uu = 1.0f - frac_u;
        const float w00 = frac_vv * frac_uu, w10 = frac_vv * frac_u, w01 = frac_v * frac_uu, w11 = frac_v * frac_u;
        const float* p00 = r0 + (iu * C - base);
        const float* p01 = hx ? p00 + C : p00;
        const float* p10 = hy ? r1 + (iu * C - base) : p00;
        const float* p11 = (hx && hy) ? r1 + (iu * C - base) + C : p00;
        uint32_t w[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float val = w00 * p00[c] + w10 * p01[c] + w01 * p10[c] + w11 * p11[c];
            if constexpr (NORM) val = (val - nrm.mean[c]) * nrm.inv_std[c];
            w[c] = __float_as_uint(val);
        }
        stream_store<C>(ow, x * C * 4, w);
    }
}
template <int C, int ITER, int BLOCK, bool LIST>
__global__ __launch_bounds__(BLOCK) void resize_rows_bilinear_kernel(Img im, float ax, float bx, float ay, float by, FastDiv by_rows, FastDiv by_parts,
                                                                     int parts, int part_cols, int seg4_max, typename ListArg<LIST>::type lst) {
    resize_rows_body<C, ITER, BLOCK, LIST, false>(im, ax, bx, ay, by, by_rows, by_parts, parts, part_cols, seg4_max, lst, Norm3f{});
}
// the same walk with the normalisation epilogue (three channels, 256-thread blocks)
template <int ITER, bool LIST>
__global__ __launch_bounds__(kRowsBlockDefault) void resize_rows_bilinear_normalize_kernel(Img im, float ax, float bx, float ay, float by, FastDiv by_rows, FastDiv by_parts,
                                                                                          int parts, int part_cols, int seg4_max, Norm3f nrm, typename ListArg<LIST>::type lst) {
    resize_rows_body<3, ITER, kRowsBlockDefault, LIST, true>(im, ax, bx, ay, by, by_rows, by_parts, parts, part_cols, seg4_max, lst, nrm);
}

// resize_bilinear_normalize_3c (P/cuda/resize.rs:184-236): bilinear resize fused with `(px - mean) * inv_std`, HWC in,
// HWC out — the sample is the same bilinear sampler as `resize`, the epilogue the reference kernel's expression.
template <bool LIST>
__global__ __launch_bounds__(kBx* kBy) void resize_normalize_kernel(Img im, float ax, float bx, float ay, float by, Norm3f n, typename ListArg<LIST>::type lst) {
    constexpr int C = 3;
    KH_PIXEL_PROLOGUE
    const float sx = clampf(ax * (float)x + bx, 0.0f, (float)(im.sw - 1));
    const float sy = clampf(ay * (float)y + by, 0.0f, (float)(im.sh - 1));
    float v[C];
    sample<C, KH_INTERP_BILINEAR>(src, im.sh, im.sw, sx, sy, v);
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = (v[c] - n.mean[c]) * n.inv_std[c];
    put<C>(o, x, v);
}

// Lanczos resize is separable in the reference (resize_lanczos_separable, lanczos.rs:189-245): an H
// pass into a dst_w x src_h f32 intermediate, then a V pass, both fmaf chains over six host-built
// table weights.  Here the two passes run fused per destination pixel — the row value `rx` IS the
// intermediate element (same chain, same f32 rounding), so the bits are identical and the
// dst_w x src_h scratch image never exists.  The tables are built on the device by the textual twin
// of lanczos_axis (:59-101): tab[i] = {x0, w0..w5} per destination index.
struct LzTap { int x0; float w[6]; };
__global__ __launch_bounds__(kBlock) void lanczos_axis_kernel(LzTap* __restrict__ tab, int dst_len, float a, float b,
                                                              float maxv) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= dst_len) return;
    const float s = clampf(a * (float)i + b, 0.0f, maxv);
    const float x0 = floorf(s), frac = s - x0;
    float w[6] = {lanczos3(frac + 2.0f), lanczos3(frac + 1.0f), lanczos3(frac),
                  lanczos3(frac - 1.0f), lanczos3(frac - 2.0f), lanczos3(frac - 3.0f)};
    const float sum = w[0] + w[1] + w[2] + w[3] + w[4] + w[5];
    const float inv = 1.0f / sum;
    LzTap t;
    t.x0 = (int)x0;
#pragma unroll
    for (int k = 0; k < 6; ++k) t.w[k] = w[k] * inv;
    tab[i] = t;
}

template <int C, bool LIST>
__global__ __launch_bounds__(kBx* kBy) void resize_lanczos_kernel(Img im, const LzTap* __restrict__ tx,
                                                                  const LzTap* __restrict__ ty, typename ListArg<LIST>::type lst) {
    KH_PIXEL_PROLOGUE
    const LzTap ax = tx[x], ay = ty[y];
    int xo[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) xo[t] = min(max(ax.x0 + t - 2, 0), im.sw - 1) * C;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 6; ++dy) {
        const int yi = min(max(ay.x0 + dy - 2, 0), im.sh - 1);
        const float* row = src + (long long)yi * im.sw * C;
        float rx[C];
#pragma unroll
        for (int c = 0; c < C; ++c) rx[c] = 0.0f;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
#pragma unroll
            for (int c = 0; c < C; ++c) rx[c] = __builtin_fmaf(ax.w[t], row[xo[t] + c], rx[c]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = __builtin_fmaf(ay.w[dy], rx[c], acc[c]);
    }
    put<C>(o, x, acc);
}

struct Mat6 { float m[6]; };
struct Mat9 { float m[9]; };

// warp_affine (P/warp/affine.rs:123-372); mi = inverse 2x3.  One destination pixel: false = out of bounds (the caller writes 0).
template <int C, int MODE>
__device__ __forceinline__ bool affine_pixel(const float* __restrict__ src, const Img& im, const Mat6& mi, int x, int y, float v[C]) {
    const float swf = (float)im.sw, shf = (float)im.sh;
    const float sx0 = mi.m[1] * (float)y + mi.m[2], sy0 = mi.m[4] * (float)y + mi.m[5];
    const float sx = mi.m[0] * (float)x + sx0, sy = mi.m[3] * (float)x + sy0;
    // in_bounds incl. the degenerate-axis rule (:201-215)
    const bool x_ok = fabsf(mi.m[0]) < 1e-6f ? (sx0 >= 0.0f && sx0 < swf) : (sx >= 0.0f && sx < swf);
    const bool y_ok = fabsf(mi.m[3]) < 1e-6f ? (sy0 >= 0.0f && sy0 < shf) : (sy >= 0.0f && sy < shf);
    if (!(x_ok && y_ok)) return false;
    if constexpr (MODE == KH_INTERP_NEAREST) {  // :270-276
        const long long xi = (long long)clampf(roundf(sx), 0.0f, swf - 1.0f);
        const long long yi = (long long)clampf(roundf(sy), 0.0f, shf - 1.0f);
        const float* p = src + (yi * im.sw + xi) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = p[c];
    } else if constexpr (MODE == KH_INTERP_BILINEAR) {  // inlined sampler, :281-318
        const float sxc = clampf(sx, 0.0f, swf - 1.0f), syc = clampf(sy, 0.0f, shf - 1.0f);
        const int x0 = (int)sxc, y0 = (int)syc;
        const int x1 = min(x0 + 1, im.sw - 1), y1 = min(y0 + 1, im.sh - 1);
        const float fx = sxc - (float)x0, fy = syc - (float)y0;
        const float w00 = (1.0f - fy) * (1.0f - fx), w10 = (1.0f - fy) * fx, w01 = fy * (1.0f - fx), w11 = fy * fx;
        const float* p00 = src + ((long long)y0 * im.sw + x0) * C;
        const float* p10 = src + ((long long)y0 * im.sw + x1) * C;
        const float* p01 = src + ((long long)y1 * im.sw + x0) * C;
        const float* p11 = src + ((long long)y1 * im.sw + x1) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = w00 * p00[c] + w10 * p10[c] + w01 * p01[c] + w11 * p11[c];
    } else {  // per-pixel samplers on the unclamped coordinate (:322-362)
        sample<C, MODE>(src, im.sh, im.sw, sx, sy, v);
    }
    return true;
}
template <int C, int MODE, bool LIST>
__global__ __launch_bounds__(kBx* kBy) void warp_affine_kernel(Img im, Mat6 mi, typename ListArg<LIST>::type lst) {
    KH_PIXEL_PROLOGUE
    float v[C];
    if (affine_pixel<C, MODE>(src, im, mi, x, y, v)) put<C>(o, x, v);
    else put_zero<C>(o, x);
}
// PX pixels of one row per lane, 64 apart (see warp_perspective_px_kernel)
template <int C, int MODE, int PX, bool LIST>
__global__ __launch_bounds__(kBx* kBy) void warp_affine_px_kernel(Img im, Mat6 mi, typename ListArg<LIST>::type lst) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(im.tiles, bx_, by_, bz_)) return;
    const int x0 = bx_ * (kBx * PX) + threadIdx.x;
    const int y = by_ * kBy + threadIdx.y;
    if (x0 >= im.dw || y >= im.dh) return;
    const float* src = image_src<LIST>(im, lst, bz_);
    const OutRow o = out_row<C>(image_dst<LIST>(im, lst, bz_) + (long long)y * im.dw * C, im.dw);
    float val[PX][C];
    bool in[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) in[j] = affine_pixel<C, MODE>(src, im, mi, min(x0 + kBx * j, im.dw - 1), y, val[j]);   // lanes past the row recompute its last pixel and store nothing
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int x = x0 + kBx * j;
        if (x >= im.dw) break;
        if (in[j]) put<C>(o, x, val[j]);
        else put_zero<C>(o, x);
    }
}

// warp_perspective (P/warp/perspective.rs:67-72,115-166); im9 = inverse 3x3
template <int C, int MODE, bool LIST>
__global__ __launch_bounds__(kBx* kBy) void warp_perspective_kernel(Img im, Mat9 h, typename ListArg<LIST>::type lst) {
    KH_PIXEL_PROLOGUE
    const float xf = (float)x, yf = (float)y;
    const float w = h.m[6] * xf + h.m[7] * yf + h.m[8];
    const float u = (h.m[0] * xf + h.m[1] * yf + h.m[2]) / w;
    const float v = (h.m[3] * xf + h.m[4] * yf + h.m[5]) / w;
    if (u >= 0.0f && u < (float)im.sw && v >= 0.0f && v < (float)im.sh) {
        float val[C];
        sample<C, MODE>(src, im.sh, im.sw, u, v, val);
        put<C>(o, x, val);
    } else {
        put_zero<C>(o, x);  // also catches NaN / Inf from w == 0
    }
}

// warp_perspective, PX pixels of one row per lane, 64 apart (round 6): every load / store instruction of a wave still covers 64
// consecutive pixels while PX independent tap quads are in flight per lane — what made the generic preprocess kernel (kGenPx = 4).
// Same expressions per pixel as warp_perspective_kernel.
template <int C, int MODE, int PX, bool LIST>
__global__ __launch_bounds__(kBx* kBy) void warp_perspective_px_kernel(Img im, Mat9 h, typename ListArg<LIST>::type lst) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(im.tiles, bx_, by_, bz_)) return;
    const int x0 = bx_ * (kBx * PX) + threadIdx.x;
    const int y = by_ * kBy + threadIdx.y;
    if (x0 >= im.dw || y >= im.dh) return;
    const float* src = image_src<LIST>(im, lst, bz_);
    const OutRow o = out_row<C>(image_dst<LIST>(im, lst, bz_) + (long long)y * im.dw * C, im.dw);
    const float yf = (float)y;
    float val[PX][C];
    bool in[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int x = min(x0 + kBx * j, im.dw - 1);   // lanes past the row recompute its last pixel and store nothing
        const float xf = (float)x;
        const float w = h.m[6] * xf + h.m[7] * yf + h.m[8];
        const float u = (h.m[0] * xf + h.m[1] * yf + h.m[2]) / w;
        const float v = (h.m[3] * xf + h.m[4] * yf + h.m[5]) / w;
        in[j] = u >= 0.0f && u < (float)im.sw && v >= 0.0f && v < (float)im.sh;
        if (in[j]) sample<C, MODE>(src, im.sh, im.sw, u, v, val[j]);
    }
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int x = x0 + kBx * j;
        if (x >= im.dw) break;
        if (in[j]) put<C>(o, x, val[j]);
        else put_zero<C>(o, x);
    }
}

// remap (P/interpolation/remap.rs:43-107).  The maps are shared by the whole batch (one camera), so
// a thread owns pixel (x, y) of kRemapNB consecutive images: the 8 B/px of map coordinates are
// read once per kRemapNB images instead of once per image (C5: 66 MB of maps per 99.5 MB image).
constexpr int kRemapNB = 4;
template <int C, int MODE, bool LIST>
__global__ __launch_bounds__(kBx* kBy) void remap_kernel(Img im, const float* __restrict__ map_x,
                                                         const float* __restrict__ map_y, int batch, typename ListArg<LIST>::type lst) {
    static_assert(kListMax % kRemapNB == 0, "remap_kernel: a block's group of images never straddles the end of a list slice");
    unsigned bx_, by_, bz_;
    if (!xcd_tile(im.tiles, bx_, by_, bz_)) return;
    const int x = bx_ * kBx + threadIdx.x;
    const int y = by_ * kBy + threadIdx.y;
    if (x >= im.dw || y >= im.dh) return;
    const long long i = (long long)y * im.dw + x;
    const float u = map_x[i], v = map_y[i];
    const bool inside = u >= 0.0f && u < (float)im.sw && v >= 0.0f && v < (float)im.sh;
    const int z0 = bz_ * kRemapNB;
#pragma unroll
    for (int k = 0; k < kRemapNB; ++k) {
        const int z = z0 + k;
        if (z >= batch) break;
        const float* src = image_src<LIST>(im, lst, (unsigned)z);
        const OutRow o = out_row<C>(image_dst<LIST>(im, lst, (unsigned)z) + (long long)y * im.dw * C, im.dw);
        if (inside) {
            float val[C];
            sample<C, MODE>(src, im.sh, im.sw, u, v, val);
            put<C>(o, x, val);
        } else {
            put_zero<C>(o, x);
        }
    }
}
// Measured against this kernel and warp_perspective_px_kernel in round 6 and not kept (scripts/ubench/bilinear3_wave_staged_r06.hip.txt,
// profiles/r06y_*): each WAVE staging the source box of its 64 x 4 / 64 x 8 tile of four images in its own LDS rows (no block barrier,
// geometry once per four images): remap 5.26 / 5.88 ms against 4.76, warp_perspective 4.62 / 4.58 against 4.63 per 128 4K images — 0.75x
// the vector-memory reads, but every wave re-reads the box rows its neighbours stage too (1.2x the L1 -> L2 requests); this kernel in
// 64 x 8 / 64 x 16 blocks: -1.6 %; neighbour-shared right-hand taps decided from the coordinates (wave shifts): -1.9 % with two pixels
// per lane, +3 % with one (scripts/ubench/warp_perspective_share2_r06.patch.txt, profiles/r06x_warp_share2.txt).

// generate_correction_map_polynomial (P/calibration/distortion.rs:68-152): all-f64 Brown-Conrady
struct Camera { double fx, fy, cx, cy, k1, k2, k3, k4, k5, k6, p1, p2; };
__global__ __launch_bounds__(kBx* kBy) void correction_map_kernel(float* __restrict__ map_x,
                                                                  float* __restrict__ map_y, int w, int h, Camera c) {
    const int xx = blockIdx.x * kBx + threadIdx.x, yy = blockIdx.y * kBy + threadIdx.y;
    if (xx >= w || yy >= h) return;
    const double x = ((double)xx - c.cx) / c.fx, y = ((double)yy - c.cy) / c.fy;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double kr = (1.0 + c.k1 * r2 + c.k2 * r4 + c.k3 * r6) / (1.0 + c.k4 * r2 + c.k5 * r4 + c.k6 * r6);
    const double x_2 = 2.0 * x, y_2 = 2.0 * y, xy_2 = x_2 * y;
    const double xd = x * kr + xy_2 * c.p1 + c.p2 * (r2 + x_2 * x);
    const double yd = y * kr + c.p1 * (r2 + y_2 * y) + xy_2 * c.p2;
    map_x[(long long)yy * w + xx] = (float)(c.fx * xd + c.cx);
    map_y[(long long)yy * w + xx] = (float)(c.fy * yd + c.cy);
}

// ---- host side ----------------------------------------------------------------------------------

int32_t check_img(const char* what, const BatchRef& b, int sw, int sh, int dw, int dh, int channels, int mode) {
    KH_REQUIRE(sw > 0 && sh > 0 && dw > 0 && dh > 0, KH_ERR_INVALID_ARG, "%s: zero-sized image (src %dx%d, dst %dx%d)",
               what, sw, sh, dw, dh);
    KH_REQUIRE(channels == 1 || channels == 3 || channels == 4, KH_ERR_UNSUPPORTED,
               "%s: no device kernel for %d channels (supported: 1, 3, 4)", what, channels);
    KH_REQUIRE(mode >= KH_INTERP_NEAREST && mode <= KH_INTERP_LANCZOS, KH_ERR_UNSUPPORTED,
               "%s: interpolation mode %d has no device kernel (nearest, bilinear, bicubic, lanczos)", what, mode);
    KH_REQUIRE(b.n >= 0 && b.n <= 65535, KH_ERR_TOO_LARGE, "%s: batch %d outside [0, 65535]", what, b.n);
    KH_REQUIRE((int64_t)sw * sh * channels <= kI32Max && (int64_t)dw * dh * channels <= kI32Max, KH_ERR_TOO_LARGE,
               "%s: image exceeds 32-bit indexing", what);
    // a destination row is one streaming-store window (out_row): 2 GiB at most
    KH_REQUIRE((int64_t)dw * channels * 4 <= kI32Max, KH_ERR_TOO_LARGE, "%s: destination rows of %d x %d floats exceed the 2 GiB store window", what, dw, channels);
    if (b.listed()) return check_list(what, b.srcs, b.dsts, b.n);
    KH_REQUIRE(b.ss >= 0 && b.ds >= 0, KH_ERR_INVALID_ARG, "%s: negative batch stride", what);
    if (b.n > 0) KH_REQUIRE(b.src && b.dst, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    return KH_OK;
}

// `b`: a strided batch or one <= kListMax slice of a list (for_each_launch); `groups` = images (or image groups) the tile grid covers
Img make_img(const BatchRef& b, int sw, int sh, int dw, int dh, int groups) {
    return Img{static_cast<const float*>(b.src), static_cast<float*>(b.dst), sw, sh, dw, dh, b.ss, b.ds,
               xcd_tiles(cdiv(dw, kBx), cdiv(dh, kBy), (unsigned)groups, cdiv(dw, kBx) * 8)};
}
#define KH_REQUIRE_TILES(what, im) \
    KH_REQUIRE((im).tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what)

#define KH_DISPATCH_C_MODE_L(KERNEL, LIST, channels, mode, grid, stream, ...)                                        \
    do {                                                                                                           \
        const dim3 blk(kBx, kBy);                                                                                  \
        switch ((channels) * 10 + (mode)) {                                                                        \
            case 10: hipLaunchKernelGGL((KERNEL<1, 0, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            case 11: hipLaunchKernelGGL((KERNEL<1, 1, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            case 12: hipLaunchKernelGGL((KERNEL<1, 2, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            case 30: hipLaunchKernelGGL((KERNEL<3, 0, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            case 31: hipLaunchKernelGGL((KERNEL<3, 1, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            case 32: hipLaunchKernelGGL((KERNEL<3, 2, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            case 13: hipLaunchKernelGGL((KERNEL<1, 3, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            case 33: hipLaunchKernelGGL((KERNEL<3, 3, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            case 43: hipLaunchKernelGGL((KERNEL<4, 3, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            case 40: hipLaunchKernelGGL((KERNEL<4, 0, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            case 41: hipLaunchKernelGGL((KERNEL<4, 1, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
            default: hipLaunchKernelGGL((KERNEL<4, 2, LIST>), grid, blk, 0, stream, __VA_ARGS__); break;           \
        }                                                                                                          \
    } while (0)
// the arguments end with `lst` (the launch's PtrList) for a list slice; an equally spaced batch passes the NoList placeholder instead
#define KH_DISPATCH_C_MODE(KERNEL, listed, channels, mode, grid, stream, lst, ...)                                  \
    do {                                                                                                           \
        if (listed) KH_DISPATCH_C_MODE_L(KERNEL, true, channels, mode, grid, stream, __VA_ARGS__, lst);            \
        else KH_DISPATCH_C_MODE_L(KERNEL, false, channels, mode, grid, stream, __VA_ARGS__, NoList{0});            \
    } while (0)

}  // namespace

extern "C" {

// P/warp/affine.rs:18-38
void kh_invert_affine_transform(const float m[6], float out[6]) {
    const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5];
    const float determinant = a * e - b * d;
    const float inv = determinant != 0.0f ? 1.0f / determinant : 0.0f;
    const float na = e * inv, nb = -b * inv, nd = -d * inv, ne = a * inv;
    out[0] = na; out[1] = nb; out[2] = -(na * c + nb * f);
    out[3] = nd; out[4] = ne; out[5] = -(nd * c + ne * f);
}

// P/warp/affine.rs:70-79
void kh_get_rotation_matrix2d(float cx, float cy, float angle_deg, float scale, float out[6]) {
    const float angle = angle_deg * 3.14159265358979323846f / 180.0f;
    const float alpha = scale * cosf(angle), beta = scale * sinf(angle);
    out[0] = alpha; out[1] = beta; out[2] = (1.0f - alpha) * cx - beta * cy;
    out[3] = -beta; out[4] = alpha; out[5] = beta * cx + (1.0f - alpha) * cy;
}

// P/warp/perspective.rs:12-60
int32_t kh_invert_homography(const float m[9], float inv[9]) {
    const float det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                      m[2] * (m[3] * m[7] - m[4] * m[6]);
    const float h8_sq = m[8] * m[8];
    const float det_norm = h8_sq > 1.1920929e-7f ? det / (h8_sq * fabsf(m[8])) : det;
    KH_REQUIRE(!(fabsf(det_norm) < 1e-10f), KH_ERR_SINGULAR, "cannot compute the determinant: singular homography");
    const float adj[9] = {
        m[4] * m[8] - m[5] * m[7], m[2] * m[7] - m[1] * m[8], m[1] * m[5] - m[2] * m[4],
        m[5] * m[6] - m[3] * m[8], m[0] * m[8] - m[2] * m[6], m[2] * m[3] - m[0] * m[5],
        m[3] * m[7] - m[4] * m[6], m[1] * m[6] - m[0] * m[7], m[0] * m[4] - m[1] * m[3],
    };
    const float inv_det = 1.0f / det;
    for (int i = 0; i < 9; ++i) inv[i] = adj[i] * inv_det;
    return KH_OK;
}

// PixelMapping::coeffs (P/cuda/resize.rs:456-473): src = a * dst + b per axis.  HalfPixel is exactly the CPU LUT's
// expression (P/resize/mod.rs:169-171); AlignCorners pins a 1-wide destination axis to source coordinate 0.
int32_t kh_pixel_mapping_coeffs(int32_t mapping, int32_t src_len, int32_t dst_len, float out[2]) {
    KH_REQUIRE(out, KH_ERR_INVALID_ARG, "kh_pixel_mapping_coeffs: null output");
    KH_REQUIRE(mapping == KH_MAP_HALF_PIXEL || mapping == KH_MAP_ALIGN_CORNERS, KH_ERR_INVALID_ARG,
               "kh_pixel_mapping_coeffs: unknown pixel mapping %d", mapping);
    KH_REQUIRE(src_len > 0 && dst_len > 0, KH_ERR_INVALID_ARG, "kh_pixel_mapping_coeffs: lengths must be positive (%d, %d)", src_len, dst_len);
    if (mapping == KH_MAP_HALF_PIXEL) {
        const float a = (float)src_len / (float)dst_len;
        out[0] = a; out[1] = 0.5f * a - 0.5f;
    } else if (dst_len > 1) {
        out[0] = (float)(src_len - 1) / (float)(dst_len - 1); out[1] = 0.0f;
    } else {
        out[0] = 0.0f; out[1] = 0.0f;
    }
    return KH_OK;
}

}  // extern "C"

namespace {

// The row-streamed bilinear downscale (resize_rows_bilinear_kernel): does this launch qualify, and how is an output row cut?
// Rows of whole float4s on 16-byte-aligned images, a vertical step >= 1.5 (no source row shared by two output rows), a tap stride of at
// most one 128-byte line (every line of the two rows is needed anyway); test option resize_rows = 0 keeps the gather kernel.
// An output row is cut into parts of `part_cols` columns, one block each (a block stages only the source-row segments under its
// columns).  Two things decide the width (profiles/r06k_resize_rows_split.txt, C2: 7 parts of 32 columns 0.450 ms; 1 / 2 / 4 / 8
// parts 0.484-0.492; 16 parts 0.90): a part's output bytes should be WHOLE 128-byte lines — two blocks writing halves of one line
// cost more than anything else here — and its segment pair small enough that eight blocks share a CU's LDS.  So: a multiple of
// the columns that make a whole line (32 for C = 1 / 3, 8 for C = 4) whose source segment is about 4 KB.
// Test option resize_rows = N > 0: N columns per part; + 1000: 64-thread blocks, + 2000: 128-thread blocks (plain resize only).
struct RowsPlan { int block, parts, part_cols, seg4_max, it; };
bool plan_rows(const BatchRef& b, int sw, int dw, int dh, int channels, float ax, float bx, float ay, RowsPlan& rp) {
    const int opt = dev_opt(kOptResizeRows);
    if (opt == 0 || (sw * channels) % 4 != 0 || !(ay >= 1.5f) || !(ax * (float)(channels * 4) <= 128.0f) || !batch_aligned(b, 16, sizeof(float))) return false;
    rp.block = opt >= 2000 ? 128 : (opt >= 1000 ? 64 : kRowsBlockDefault);
    const int unit = channels == 4 ? 8 : 32;
    rp.part_cols = opt > 0 && opt % 1000 ? opt % 1000 : std::max(1, (int)(4096.0f / (ax * (float)(channels * 4))) / unit) * unit;
    if (rp.part_cols > dw) rp.part_cols = dw;
    rp.parts = (dw + rp.part_cols - 1) / rp.part_cols;
    // the longest source segment any part needs (host evaluation of the kernel's own expressions)
    rp.seg4_max = 1;
    for (int part = 0; part < rp.parts; ++part) {
        const int x_lo = part * rp.part_cols, x_hi = std::min(x_lo + rp.part_cols, dw);
        const int s4 = (rows_src_px(ax, bx, x_lo, sw) * channels) >> 2;
        const int e4 = std::min(((std::min(rows_src_px(ax, bx, x_hi - 1, sw) + 1, sw - 1) + 1) * channels + 3) >> 2, (sw * channels) >> 2);
        rp.seg4_max = std::max(rp.seg4_max, e4 - s4);
    }
    const int iters = (rp.seg4_max + rp.block - 1) / rp.block;
    rp.it = iters <= 1 ? 1 : (iters <= 2 ? 2 : (iters <= 4 ? 4 : 8));
    return iters <= 8 && (int64_t)dh * rp.parts * b.n <= kI32Max && (size_t)rp.seg4_max * 32 <= 64 * 1024;
}

int32_t resize_impl(const char* what, kh_stream_t stream, const BatchRef& b, int sw, int sh, int dw, int dh, int channels, int mode,
                    int mapping) {
    if (int32_t rc = check_img(what, b, sw, sh, dw, dh, channels, mode)) return rc;
    float cx[2], cy[2];
    if (int32_t rc = kh_pixel_mapping_coeffs(mapping, sw, dw, cx)) return rc;
    if (int32_t rc = kh_pixel_mapping_coeffs(mapping, sh, dh, cy)) return rc;
    if (b.n == 0) return KH_OK;
    const float ax = cx[0], bx = cx[1], ay = cy[0], by = cy[1];
    hipStream_t st = as_hip(stream);
    if (mode == KH_INTERP_LANCZOS) {  // P/resize/mod.rs:139-146
        // (dw + dh) x 28 B of stream-ordered scratch for the axis tables, as the reference adapter
        // allocates its tables/intermediate per call (P/resize/cuda.rs:151-190)
        Scratch scratch;
        if (int32_t rc = get_scratch(stream, sizeof(LzTap) * ((size_t)dw + dh), "kh_resize_f32 (lanczos)", scratch)) return rc;
        LzTap* tab = scratch.as<LzTap>();
        hipLaunchKernelGGL(lanczos_axis_kernel, dim3(cdiv(dw, kBlock)), dim3(kBlock), 0, st, tab, dw, ax, bx,
                           (float)(sw - 1));
        hipLaunchKernelGGL(lanczos_axis_kernel, dim3(cdiv(dh, kBlock)), dim3(kBlock), 0, st, tab + dw, dh, ay, by,
                           (float)(sh - 1));
        return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
            const Img im = make_img(c, sw, sh, dw, dh, c.n);
            KH_REQUIRE_TILES(what, im);
            const dim3 blk(kBx, kBy), grid = xcd_grid(im.tiles);
            const NoList none{0};
            if (c.listed()) switch (channels) {
                case 1: hipLaunchKernelGGL((resize_lanczos_kernel<1, true>), grid, blk, 0, st, im, tab, tab + dw, lst); break;
                case 3: hipLaunchKernelGGL((resize_lanczos_kernel<3, true>), grid, blk, 0, st, im, tab, tab + dw, lst); break;
                default: hipLaunchKernelGGL((resize_lanczos_kernel<4, true>), grid, blk, 0, st, im, tab, tab + dw, lst); break;
            } else switch (channels) {
                case 1: hipLaunchKernelGGL((resize_lanczos_kernel<1, false>), grid, blk, 0, st, im, tab, tab + dw, none); break;
                case 3: hipLaunchKernelGGL((resize_lanczos_kernel<3, false>), grid, blk, 0, st, im, tab, tab + dw, none); break;
                default: hipLaunchKernelGGL((resize_lanczos_kernel<4, false>), grid, blk, 0, st, im, tab, tab + dw, none); break;
            }
            return check_launch("kh_resize_f32 (lanczos)");
        });
    }
    // bilinear downscales whose taps touch nearly every line of two source rows per output row: the row-streamed kernel (test option
    // resize_rows = 0 keeps the gather kernel)
    RowsPlan rp;
    if (mode == KH_INTERP_BILINEAR && plan_rows(b, sw, dw, dh, channels, ax, bx, ay, rp))
        return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
            Img im = make_img(c, sw, sh, dw, dh, c.n);
            const dim3 grid((unsigned)dh * (unsigned)rp.parts * (unsigned)c.n), blk(rp.block);
            const size_t lds = (size_t)2 * rp.seg4_max * 16;
            const FastDiv by_rows = fast_div((uint32_t)dh * (uint32_t)rp.parts), by_parts = fast_div((uint32_t)rp.parts);
            const NoList none{0};
            const int parts = rp.parts, part_cols = rp.part_cols, seg4_max = rp.seg4_max, block = rp.block;
#define KH_ROWS(CC, IT, BL)                                                                                                                                       \
    do {                                                                                                                                                          \
        if (c.listed()) hipLaunchKernelGGL((resize_rows_bilinear_kernel<CC, IT, BL, true>), grid, blk, lds, st, im, ax, bx, ay, by, by_rows, by_parts, parts, part_cols, seg4_max, lst); \
        else hipLaunchKernelGGL((resize_rows_bilinear_kernel<CC, IT, BL, false>), grid, blk, lds, st, im, ax, bx, ay, by, by_rows, by_parts, parts, part_cols, seg4_max, none);          \
    } while (0)
#define KH_ROWS_B(CC, IT) do { if (block == 64) KH_ROWS(CC, IT, 64); else if (block == 128) KH_ROWS(CC, IT, 128); else KH_ROWS(CC, IT, 256); } while (0)
#define KH_ROWS_C(CC) do { switch (rp.it) { case 1: KH_ROWS_B(CC, 1); break; case 2: KH_ROWS_B(CC, 2); break; case 4: KH_ROWS_B(CC, 4); break; default: KH_ROWS_B(CC, 8); break; } } while (0)
            switch (channels) { case 1: KH_ROWS_C(1); break; case 3: KH_ROWS_C(3); break; default: KH_ROWS_C(4); break; }
#undef KH_ROWS_C
#undef KH_ROWS_B
#undef KH_ROWS
            return check_launch(what);
        });
    // bicubic with a horizontal step of exactly 2 (sw == 2 dw, half-pixel grid): neighbours' columns through wave shifts
    // (resize_bicubic_half_kernel).  Test option resize_rows = 0 keeps the gather kernel here too.
    if (mode == KH_INTERP_BICUBIC && dev_opt(kOptResizeRows) != 0 && sw == 2 * dw && ax == 2.0f && bx == 0.5f && dw < (1 << 22)) {
        // both steps exactly 2: two output rows per lane (test option resize_rows = 1: one)
        const bool vh = sh == 2 * dh && ay == 2.0f && by == 0.5f && dh < (1 << 22) && dev_opt(kOptResizeRows) != 1;
        return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
            Img im = make_img(c, sw, sh, dw, dh, c.n);
            if (vh) im.tiles = xcd_tiles(cdiv(dw, kBx), cdiv(dh, 2 * kBy), (unsigned)c.n, cdiv(dw, kBx) * 4);
            KH_REQUIRE_TILES(what, im);
            const dim3 blk(kBx, kBy), grid = xcd_grid(im.tiles);
            const NoList none{0};
#define KH_HALF(CC, VH) do { if (c.listed()) hipLaunchKernelGGL((resize_bicubic_half_kernel<CC, VH, true>), grid, blk, 0, st, im, ax, bx, ay, by, lst); \
                             else hipLaunchKernelGGL((resize_bicubic_half_kernel<CC, VH, false>), grid, blk, 0, st, im, ax, bx, ay, by, none); } while (0)
#define KH_HALF_C(CC) do { if (vh) KH_HALF(CC, true); else KH_HALF(CC, false); } while (0)
            switch (channels) { case 1: KH_HALF_C(1); break; case 3: KH_HALF_C(3); break; default: KH_HALF_C(4); break; }
#undef KH_HALF_C
#undef KH_HALF
            return check_launch(what);
        });
    }
    // one channel, nearest / bilinear, destination rows of whole float4s on 16-byte-aligned images: four pixels per lane — nearest upscales
    // 0.160 -> 0.072 ms per 8 1080p -> 4K planes, bilinear 0.140 -> 0.130; bicubic LOSES 2x with its 64 gathers per lane (0.22 -> 0.48) and
    // keeps one pixel per lane (profiles/r06zz5_resize_f32_gray.txt; test option resize_rows = 0 keeps one pixel per lane everywhere)
    const bool quads1 = channels == 1 && (mode == KH_INTERP_NEAREST || mode == KH_INTERP_BILINEAR) && dw % 4 == 0 && batch_aligned(b, 16, 4) && dev_opt(kOptResizeRows) != 0;
    return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
        Img im = make_img(c, sw, sh, dw, dh, c.n);
        if (quads1) {
            im.tiles = xcd_tiles(cdiv(dw, 4 * kBx), cdiv(dh, kBy), (unsigned)c.n, cdiv(dw, 4 * kBx) * 8);
            KH_REQUIRE_TILES(what, im);
            const dim3 blk(kBx, kBy), grid = xcd_grid(im.tiles);
            const NoList none{0};
#define KH_Q1(MM) do { if (c.listed()) hipLaunchKernelGGL((resize_quads1_kernel<MM, true>), grid, blk, 0, st, im, ax, bx, ay, by, lst); \
                       else hipLaunchKernelGGL((resize_quads1_kernel<MM, false>), grid, blk, 0, st, im, ax, bx, ay, by, none); } while (0)
            if (mode == KH_INTERP_NEAREST) KH_Q1(0); else KH_Q1(1);
#undef KH_Q1
            return check_launch(what);
        }
        KH_REQUIRE_TILES(what, im);
        KH_DISPATCH_C_MODE(resize_kernel, c.listed(), channels, mode, xcd_grid(im.tiles), st, lst, im, ax, bx, ay, by);
        return check_launch(what);
    });
}

int32_t resize_normalize_impl(const char* what, kh_stream_t stream, const BatchRef& b, int sw, int sh, int dw, int dh, const float* mean,
                              const float* std_dev, int mapping) {
    if (int32_t rc = check_img(what, b, sw, sh, dw, dh, 3, KH_INTERP_BILINEAR)) return rc;
    KH_REQUIRE(mean && std_dev, KH_ERR_INVALID_ARG, "%s: null mean / std", what);
    // launch_resize_bilinear_normalize_cuda, P/cuda/resize.rs:606-610
    KH_REQUIRE(std_dev[0] != 0.0f && std_dev[1] != 0.0f && std_dev[2] != 0.0f, KH_ERR_INVALID_ARG,
               "%s: std must be non-zero for all channels", what);
    float cx[2], cy[2];
    if (int32_t rc = kh_pixel_mapping_coeffs(mapping, sw, dw, cx)) return rc;
    if (int32_t rc = kh_pixel_mapping_coeffs(mapping, sh, dh, cy)) return rc;
    if (b.n == 0) return KH_OK;
    Norm3f n;
    for (int c = 0; c < 3; ++c) { n.mean[c] = mean[c]; n.inv_std[c] = 1.0f / std_dev[c]; }
    // the row-streamed walk of `resize` with the normalisation epilogue (round 6), where a plain resize of this geometry takes it
    RowsPlan rp;
    if (plan_rows(b, sw, dw, dh, 3, cx[0], cx[1], cy[0], rp) && rp.block == kRowsBlockDefault)
        return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
            Img im = make_img(c, sw, sh, dw, dh, c.n);
            const dim3 grid((unsigned)dh * (unsigned)rp.parts * (unsigned)c.n), blk(rp.block);
            const size_t lds = (size_t)2 * rp.seg4_max * 16;
            const FastDiv by_rows = fast_div((uint32_t)dh * (uint32_t)rp.parts), by_parts = fast_div((uint32_t)rp.parts);
            const NoList none{0};
            hipStream_t st = as_hip(stream);
#define KH_ROWSN(IT) do { if (c.listed()) hipLaunchKernelGGL((resize_rows_bilinear_normalize_kernel<IT, true>), grid, blk, lds, st, im, cx[0], cx[1], cy[0], cy[1], by_rows, by_parts, rp.parts, rp.part_cols, rp.seg4_max, n, lst); \
                          else hipLaunchKernelGGL((resize_rows_bilinear_normalize_kernel<IT, false>), grid, blk, lds, st, im, cx[0], cx[1], cy[0], cy[1], by_rows, by_parts, rp.parts, rp.part_cols, rp.seg4_max, n, none); } while (0)
            switch (rp.it) { case 1: KH_ROWSN(1); break; case 2: KH_ROWSN(2); break; case 4: KH_ROWSN(4); break; default: KH_ROWSN(8); break; }
#undef KH_ROWSN
            return check_launch(what);
        });
    return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
        const Img im = make_img(c, sw, sh, dw, dh, c.n);
        KH_REQUIRE_TILES(what, im);
        if (c.listed()) hipLaunchKernelGGL(resize_normalize_kernel<true>, xcd_grid(im.tiles), dim3(kBx, kBy), 0, as_hip(stream), im, cx[0], cx[1], cy[0], cy[1], n, lst);
        else hipLaunchKernelGGL(resize_normalize_kernel<false>, xcd_grid(im.tiles), dim3(kBx, kBy), 0, as_hip(stream), im, cx[0], cx[1], cy[0], cy[1], n, NoList{0});
        return check_launch(what);
    });
}

int32_t warp_affine_impl(const char* what, kh_stream_t stream, const BatchRef& b, int sw, int sh, int dw, int dh, int channels,
                         const float* m, int mode) {
    if (int32_t rc = check_img(what, b, sw, sh, dw, dh, channels, mode)) return rc;
    KH_REQUIRE(m, KH_ERR_INVALID_ARG, "%s: null matrix", what);
    if (b.n == 0) return KH_OK;
    Mat6 mi;
    kh_invert_affine_transform(m, mi.m);
    // Two pixels per lane (see kh_warp_perspective_f32) only where a destination row maps to a nearly horizontal source run: the 12-degree
    // rotation of the bench (a 128-pixel run crosses 27 source rows) LOSES 10 % with it (2.62 vs 2.38 ms, profiles/r06p_px2_bench.txt).
    // Test option warp_f32_px: 1 = never, 2 = always.
    const int px_opt = dev_opt(kOptWarpF32Px);
    if (mode == KH_INTERP_BILINEAR && px_opt != 1 && (px_opt == 2 || fabsf(mi.m[3]) * (float)(2 * kBx) <= kPxMaxRows))
        return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
            constexpr int PX = 2;
            Img im = make_img(c, sw, sh, dw, dh, c.n);
            im.tiles = xcd_tiles(cdiv(dw, kBx * PX), cdiv(dh, kBy), (unsigned)c.n, cdiv(dw, kBx * PX) * 8);
            KH_REQUIRE_TILES(what, im);
            const dim3 blk(kBx, kBy), grid = xcd_grid(im.tiles);
            const NoList none{0};
#define KH_PX(CC) do { if (c.listed()) hipLaunchKernelGGL((warp_affine_px_kernel<CC, KH_INTERP_BILINEAR, PX, true>), grid, blk, 0, as_hip(stream), im, mi, lst); \
                       else hipLaunchKernelGGL((warp_affine_px_kernel<CC, KH_INTERP_BILINEAR, PX, false>), grid, blk, 0, as_hip(stream), im, mi, none); } while (0)
            switch (channels) { case 1: KH_PX(1); break; case 3: KH_PX(3); break; default: KH_PX(4); break; }
#undef KH_PX
            return check_launch(what);
        });
    return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
        const Img im = make_img(c, sw, sh, dw, dh, c.n);
        KH_REQUIRE_TILES(what, im);
        KH_DISPATCH_C_MODE(warp_affine_kernel, c.listed(), channels, mode, xcd_grid(im.tiles), as_hip(stream), lst, im, mi);
        return check_launch(what);
    });
}

int32_t warp_perspective_impl(const char* what, kh_stream_t stream, const BatchRef& b, int sw, int sh, int dw, int dh, int channels,
                              const float* m, int mode) {
    if (int32_t rc = check_img(what, b, sw, sh, dw, dh, channels, mode)) return rc;
    KH_REQUIRE(m, KH_ERR_INVALID_ARG, "%s: null matrix", what);
    Mat9 h;
    if (int32_t rc = kh_invert_homography(m, h.m)) return rc;  // rejected on the host, before any launch
    if (b.n == 0) return KH_OK;
    // Bilinear: two pixels of a row per lane, 64 apart (warp_perspective_px_kernel): 4.58 vs 4.88 ms per 128 4K images, four pixels
    // 4.76 (profiles/r06n_warp_px.txt); an LDS-staged twin of the u8 gather (source box of a 64 x 16 tile copied with 16-byte loads,
    // four images per block, next box requested while the current one is sampled) ran 7.77 ms (r06o) and is not in the library.
    // Only where a destination row maps to a nearly horizontal source run (source rows crossed by a 128-pixel destination run, sampled at
    // the corners and the centre of the destination <= kPxMaxRows): a rotated run spreads a wave's taps over many rows and the wider
    // tile then loses (warp_affine by 12 degrees: -10 %, r06p).  Test option warp_f32_px: 1 = the one-pixel kernel, 2 = always two.
    const int px_opt = dev_opt(kOptWarpF32Px);
    bool flat_rows = true;
    for (int k = 0; k < 5 && flat_rows; ++k) {
        const float xs = k == 4 ? 0.5f * (float)dw : ((k & 1) ? (float)(dw - 1) : 0.0f), ys = k == 4 ? 0.5f * (float)dh : ((k & 2) ? (float)(dh - 1) : 0.0f);
        auto vmap = [&](float xf, float yf) { return (h.m[3] * xf + h.m[4] * yf + h.m[5]) / (h.m[6] * xf + h.m[7] * yf + h.m[8]); };
        const float tilt = fabsf(vmap(xs + 1.0f, ys) - vmap(xs, ys)) * (float)(2 * kBx);
        flat_rows = tilt <= kPxMaxRows;   // (false for NaN)
    }
    if (mode == KH_INTERP_BILINEAR && px_opt != 1 && (px_opt == 2 || flat_rows))
        return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
            constexpr int PX = 2;
            Img im = make_img(c, sw, sh, dw, dh, c.n);
            im.tiles = xcd_tiles(cdiv(dw, kBx * PX), cdiv(dh, kBy), (unsigned)c.n, cdiv(dw, kBx * PX) * 8);
            KH_REQUIRE_TILES(what, im);
            const dim3 blk(kBx, kBy), grid = xcd_grid(im.tiles);
            const NoList none{0};
#define KH_PX(CC) do { if (c.listed()) hipLaunchKernelGGL((warp_perspective_px_kernel<CC, KH_INTERP_BILINEAR, PX, true>), grid, blk, 0, as_hip(stream), im, h, lst); \
                       else hipLaunchKernelGGL((warp_perspective_px_kernel<CC, KH_INTERP_BILINEAR, PX, false>), grid, blk, 0, as_hip(stream), im, h, none); } while (0)
            switch (channels) { case 1: KH_PX(1); break; case 3: KH_PX(3); break; default: KH_PX(4); break; }
#undef KH_PX
            return check_launch(what);
        });
    return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
        const Img im = make_img(c, sw, sh, dw, dh, c.n);
        KH_REQUIRE_TILES(what, im);
        KH_DISPATCH_C_MODE(warp_perspective_kernel, c.listed(), channels, mode, xcd_grid(im.tiles), as_hip(stream), lst, im, h);
        return check_launch(what);
    });
}

int32_t remap_impl(const char* what, kh_stream_t stream, const BatchRef& b, const float* map_x, const float* map_y, int sw, int sh,
                   int dw, int dh, int channels, int mode) {
    if (int32_t rc = check_img(what, b, sw, sh, dw, dh, channels, mode)) return rc;
    if (b.n == 0) return KH_OK;
    KH_REQUIRE(map_x && map_y, KH_ERR_INVALID_ARG, "%s: null map pointer", what);
    return for_each_launch(b, [&](const BatchRef& c, int, const PtrList& lst) -> int32_t {
        const int groups = (c.n + kRemapNB - 1) / kRemapNB;
        const Img im = make_img(c, sw, sh, dw, dh, groups);
        KH_REQUIRE_TILES(what, im);
        KH_DISPATCH_C_MODE(remap_kernel, c.listed(), channels, mode, xcd_grid(im.tiles), as_hip(stream), lst, im, map_x, map_y, c.n);
        return check_launch(what);
    });
}

}  // namespace

extern "C" {

int32_t kh_resize_f32(kh_stream_t stream, const float* src, float* dst, int32_t sw, int32_t sh, int32_t dw, int32_t dh,
                      int32_t channels, int32_t mode, int32_t batch, int64_t src_stride, int64_t dst_stride) {
    return resize_impl("kh_resize_f32", stream, strided_batch(src, dst, batch, src_stride, dst_stride), sw, sh, dw, dh, channels, mode, KH_MAP_HALF_PIXEL);
}

int32_t kh_resize_mapped_f32(kh_stream_t stream, const float* src, float* dst, int32_t sw, int32_t sh, int32_t dw, int32_t dh,
                             int32_t channels, int32_t mode, int32_t mapping, int32_t batch, int64_t src_stride,
                             int64_t dst_stride) {
    return resize_impl("kh_resize_f32", stream, strided_batch(src, dst, batch, src_stride, dst_stride), sw, sh, dw, dh, channels, mode, mapping);
}

int32_t kh_resize_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t sw, int32_t sh,
                           int32_t dw, int32_t dh, int32_t channels, int32_t mode, int32_t mapping) {
    return resize_impl("kh_resize_f32_list", stream, listed_batch(srcs, dsts, n), sw, sh, dw, dh, channels, mode, mapping);
}

int32_t kh_resize_bilinear_normalize_f32(kh_stream_t stream, const float* src, float* dst, int32_t sw, int32_t sh, int32_t dw,
                                         int32_t dh, const float* mean, const float* std_dev, int32_t mapping, int32_t batch,
                                         int64_t src_stride, int64_t dst_stride) {
    return resize_normalize_impl("kh_resize_bilinear_normalize_f32", stream, strided_batch(src, dst, batch, src_stride, dst_stride), sw, sh, dw, dh,
                                 mean, std_dev, mapping);
}

int32_t kh_resize_bilinear_normalize_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t sw,
                                              int32_t sh, int32_t dw, int32_t dh, const float* mean, const float* std_dev,
                                              int32_t mapping) {
    return resize_normalize_impl("kh_resize_bilinear_normalize_f32_list", stream, listed_batch(srcs, dsts, n), sw, sh, dw, dh, mean, std_dev, mapping);
}

int32_t kh_warp_affine_f32(kh_stream_t stream, const float* src, float* dst, int32_t sw, int32_t sh, int32_t dw,
                           int32_t dh, int32_t channels, const float* m, int32_t mode, int32_t batch,
                           int64_t src_stride, int64_t dst_stride) {
    return warp_affine_impl("kh_warp_affine_f32", stream, strided_batch(src, dst, batch, src_stride, dst_stride), sw, sh, dw, dh, channels, m, mode);
}

int32_t kh_warp_affine_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t sw, int32_t sh,
                                int32_t dw, int32_t dh, int32_t channels, const float* m, int32_t mode) {
    return warp_affine_impl("kh_warp_affine_f32_list", stream, listed_batch(srcs, dsts, n), sw, sh, dw, dh, channels, m, mode);
}

int32_t kh_warp_perspective_f32(kh_stream_t stream, const float* src, float* dst, int32_t sw, int32_t sh, int32_t dw,
                                int32_t dh, int32_t channels, const float* m, int32_t mode, int32_t batch,
                                int64_t src_stride, int64_t dst_stride) {
    return warp_perspective_impl("kh_warp_perspective_f32", stream, strided_batch(src, dst, batch, src_stride, dst_stride), sw, sh, dw, dh, channels,
                                 m, mode);
}

int32_t kh_warp_perspective_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t sw,
                                     int32_t sh, int32_t dw, int32_t dh, int32_t channels, const float* m, int32_t mode) {
    return warp_perspective_impl("kh_warp_perspective_f32_list", stream, listed_batch(srcs, dsts, n), sw, sh, dw, dh, channels, m, mode);
}

int32_t kh_remap_f32(kh_stream_t stream, const float* src, const float* map_x, const float* map_y, float* dst,
                     int32_t sw, int32_t sh, int32_t dw, int32_t dh, int32_t channels, int32_t mode, int32_t batch,
                     int64_t src_stride, int64_t dst_stride) {
    return remap_impl("kh_remap_f32", stream, strided_batch(src, dst, batch, src_stride, dst_stride), map_x, map_y, sw, sh, dw, dh, channels, mode);
}

int32_t kh_remap_f32_list(kh_stream_t stream, const float* const* srcs, const float* map_x, const float* map_y, float* const* dsts,
                          int32_t n, int32_t sw, int32_t sh, int32_t dw, int32_t dh, int32_t channels, int32_t mode) {
    return remap_impl("kh_remap_f32_list", stream, listed_batch(srcs, dsts, n), map_x, map_y, sw, sh, dw, dh, channels, mode);
}

int32_t kh_correction_map_polynomial_f32(kh_stream_t stream, float* map_x, float* map_y, int32_t w, int32_t h,
                                         const double* intrinsic, const double* distortion) {
    KH_REQUIRE(w > 0 && h > 0, KH_ERR_INVALID_ARG, "kh_correction_map_polynomial_f32: zero-sized map %dx%d", w, h);
    KH_REQUIRE(map_x && map_y && intrinsic && distortion, KH_ERR_INVALID_ARG,
               "kh_correction_map_polynomial_f32: null pointer");
    KH_REQUIRE((int64_t)w * h <= kI32Max, KH_ERR_TOO_LARGE, "kh_correction_map_polynomial_f32: map too large");
    const Camera c{intrinsic[0], intrinsic[1], intrinsic[2], intrinsic[3], distortion[0], distortion[1],
                   distortion[2], distortion[3], distortion[4], distortion[5], distortion[6], distortion[7]};
    hipLaunchKernelGGL(correction_map_kernel, dim3(cdiv(w, kBx), cdiv(h, kBy)), dim3(kBx, kBy), 0, as_hip(stream),
                       map_x, map_y, w, h, c);
    return check_launch("kh_correction_map_polynomial_f32");
}

}  // extern "C"
