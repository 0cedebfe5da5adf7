// u8 resize cascade (resize_fast_u8_aa) and the OpenCV-compatible resize for gfx950.
//
// Device twins of P/cuda/resize_u8.rs (adapters P/resize/cuda.rs:207-330); arithmetic and routing are
// those of the CPU ops: resize_u8_path + resize_fast_u8_aa (P/resize/mod.rs:283-400), the exact-2x RGB
// box / 75-25 paths (P/resize/pyramid.rs, P/resize/kernels.rs:62-74,166-183,272-281), nearest
// (P/resize/nearest.rs:18-21), Q14 bilinear with f64 coordinates (P/resize/bilinear.rs:25-39,
// P/resize/kernels.rs:1141-1165), Q14 separable bicubic / Lanczos-3 with optional antialias
// (P/resize/common.rs:62-125, P/resize/kernels.rs:403-425,699-708), and resize_opencv_{u8,f32}
// (P/resize/opencv_compat.rs:22-250; CPU-only in the reference).  Byte-identical results
// (tests/test_resize_u8_gpu.py, incl. the reference's cv2 golden vectors).
//
// All gathers, one thread per destination pixel in 64x4 tiles.  Coordinates of the nearest / bilinear
// / OpenCV paths are evaluated per thread in f64 exactly as the reference's host LUT builders do
// (IEEE f64, no contraction), so no tables exist; the separable path's Q14 contribution tables need
// libm `sin` in f64 and are therefore built on the HOST by the reference's algorithm, uploaded once
// and cached per (src, dst, filter, antialias, device) — the reference's sync-before-publish rule
// (P/resize/cuda.rs:151-190).
#include <math.h>

#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>

#include "kh_common.h"
#include "kh_table_cache.h"

using namespace kh;

namespace {

constexpr int kBx = 64, kBy = 4;

struct Rz {
    const uint8_t* src;
    uint8_t* dst;
    int sw, sh, dw, dh;
    long long ss, ds;  // bytes between consecutive images
    double scale_x, scale_y;
    XcdTiles tiles;
    int plain;         // quad kernels: write-back instead of streaming stores (kh_common.h::plain_row_stores)
};

#define KH_RZ_PROLOGUE                                          \
    unsigned bx_, by_, bz_;                                     \
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;              \
    const int x = bx_ * kBx + threadIdx.x;                      \
    const int y = by_ * kBy + threadIdx.y;                      \
    if (x >= a.dw || y >= a.dh) return;                         \
    const uint8_t* __restrict__ src = a.src + (long long)bz_ * a.ss; \
    uint8_t* __restrict__ dst = a.dst + (long long)bz_ * a.ds;

// ---- exact-2x RGB -------------------------------------------------------------------------------------
// Every simple path below computes ONE destination pixel as a packed word (channel c = bits [8c, 8c + 8)); the scalar kernel stores its
// C bytes, the quad kernel (round 6) four consecutive pixels of a row as C dwords.
__device__ __forceinline__ uint32_t px_down2(const Rz& a, const uint8_t* __restrict__ src, int x, int y) {
    const uint8_t* r0 = src + ((long long)(2 * y) * a.sw + 2 * x) * 3;
    const uint8_t* r1 = r0 + (long long)a.sw * 3;
    uint32_t v = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) v |= ((((uint32_t)r0[ch] + r0[3 + ch] + r1[ch] + r1[3 + ch] + 2u) >> 2) & 0xffu) << (8 * ch);
    return v;
}

__device__ __forceinline__ uint32_t rh(uint32_t p, uint32_t q) { return (p + q + 1u) >> 1; }
// hinterp_row_rgb_u8 at output column X of one source row (P/resize/kernels.rs:166-183)
__device__ __forceinline__ uint32_t hval(const uint8_t* row, int sw, int X, int ch) {
    if (X == 0) return row[ch];
    if (X == 2 * sw - 1) return row[(sw - 1) * 3 + ch];
    const int j = (X - 1) >> 1;
    const uint32_t p = row[j * 3 + ch], q = row[(j + 1) * 3 + ch], avg = rh(p, q);
    return (X & 1) ? rh(p, avg) : rh(q, avg);
}
__device__ __forceinline__ uint32_t px_up2(const Rz& a, const uint8_t* __restrict__ src, int x, int y) {
    const long long stride = (long long)a.sw * 3;
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        uint32_t v;
        if (y == 0) v = hval(src, a.sw, x, ch);
        else if (y == 2 * a.sh - 1) v = hval(src + (a.sh - 1) * stride, a.sw, x, ch);
        else {
            const int i = (y - 1) >> 1;  // rows 2i+1, 2i+2 blend source rows i and i+1
            const uint32_t ha = hval(src + i * stride, a.sw, x, ch), hb = hval(src + (i + 1) * stride, a.sw, x, ch);
            v = (y & 1) ? rh(ha, rh(ha, hb)) : rh(hb, rh(hb, ha));  // blend_75_25_row
        }
        out |= (v & 0xffu) << (8 * ch);
    }
    return out;
}

// ---- nearest / bilinear -------------------------------------------------------------------------------
__device__ __forceinline__ int nearest_index(int i, double scale, int src_len) {  // nearest.rs:18-21
    const double v = floor(((double)i + 0.5) * scale);
    return (int)fmin(fmax(v, 0.0), (double)(src_len - 1));
}
__device__ __forceinline__ int cv_nearest_index(int i, double iscale, int src_len) {  // opencv_compat.rs nearest_axis, :67-74
    return (int)fmin(floor((double)i * iscale), (double)(src_len - 1));
}
__device__ __forceinline__ void bilinear_tap(int i, double scale, int src_len, int& ofs, uint32_t& fq) {  // bilinear.rs:25-39
    const double s = ((double)i + 0.5) * scale - 0.5;
    const double fl = floor(s);
    double f = s - fl;
    long long i0 = (long long)fl;
    if (i0 < 0) { i0 = 0; f = 0.0; }
    else if (i0 >= (long long)src_len - 1) { i0 = (long long)src_len - 2; f = 1.0; }
    const uint32_t q = (uint32_t)round(f * 16384.0);
    ofs = (int)i0;
    fq = q > 16384u ? 16384u : q;
}
template <int C>
__device__ __forceinline__ uint32_t px_nearest(const Rz& a, const uint8_t* __restrict__ src, int x, int sy) {
    const int sx = nearest_index(x, a.scale_x, a.sw);
    return load_px_u8<C>(src + ((long long)sy * a.sw + sx) * C);
}
template <int C>
__device__ __forceinline__ uint32_t px_bilinear(const Rz& a, const uint8_t* __restrict__ src, int x, int yi, uint32_t fy) {
    int xi;
    uint32_t fx;
    bilinear_tap(x, a.scale_x, a.sw, xi, fx);
    const uint64_t fx1 = 16384u - fx, fy1 = 16384u - fy;
    // xi <= sw - 2, yi <= sh - 2: both neighbours exist
    const QuadU8 q = load_quad_u8<C>(src + (unsigned)(yi * a.sw) * C, src + (unsigned)((yi + 1) * a.sw) * C, xi, a.sw);
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {  // bilinear_row_u8_scalar, kernels.rs:1141-1165 (u64 accumulate)
        const uint64_t top = (uint64_t)chan_u8(q.p00, ch) * fx1 + (uint64_t)chan_u8(q.p01, ch) * fx;
        const uint64_t bot = (uint64_t)chan_u8(q.p10, ch) * fx1 + (uint64_t)chan_u8(q.p11, ch) * fx;
        out |= ((uint32_t)((top * fy1 + bot * fy + (1ull << 27)) >> 28) & 0xffu) << (8 * ch);
    }
    return out;
}
// OP: 0 nearest, 1 Q14 bilinear, 2 exact-2x RGB box, 3 exact-2x RGB upscale (75 / 25)
enum { kRzNearest = 0, kRzBilinear = 1, kRzDown2 = 2, kRzUp2 = 3 };
template <int OP>
__device__ __forceinline__ void px_row_setup(const Rz& a, int y, int& yi, uint32_t& fy) {
    yi = y; fy = 0;
    if constexpr (OP == kRzNearest) yi = nearest_index(y, a.scale_y, a.sh);
    else if constexpr (OP == kRzBilinear) bilinear_tap(y, a.scale_y, a.sh, yi, fy);
}
template <int C, int OP>
__device__ __forceinline__ uint32_t px_any(const Rz& a, const uint8_t* __restrict__ src, int x, int y, int yi, uint32_t fy) {
    if constexpr (OP == kRzNearest) return px_nearest<C>(a, src, x, yi);
    else if constexpr (OP == kRzBilinear) return px_bilinear<C>(a, src, x, yi, fy);
    else if constexpr (OP == kRzDown2) { static_assert(C == 3, "px_down2 is the RGB special case"); return px_down2(a, src, x, y); }
    else return px_up2(a, src, x, y);
}
// four packed pixels -> C dwords -> one streaming store at pixel x0 of the row window
template <int C>
__device__ __forceinline__ void store_quad_u8(__amdgpu_buffer_rsrc_t ow, int x0, const uint32_t (&p)[4], int plain) {
    uint32_t w[C];
    if constexpr (C == 1) w[0] = p[0] | (p[1] << 8) | (p[2] << 16) | (p[3] << 24);
    else if constexpr (C == 2) { w[0] = p[0] | (p[1] << 16); w[1] = p[2] | (p[3] << 16); }
    else if constexpr (C == 3) { w[0] = p[0] | (p[1] << 24); w[1] = (p[1] >> 8) | (p[2] << 16); w[2] = (p[2] >> 16) | (p[3] << 8); }
    else { w[0] = p[0]; w[1] = p[1]; w[2] = p[2]; w[3] = p[3]; }
    row_store<C>(ow, x0 * C, w, plain);
}
template <int C, int OP>
__global__ __launch_bounds__(kBx* kBy) void resize_u8_px_kernel(Rz a) {
    KH_RZ_PROLOGUE
    int yi;
    uint32_t fy;
    px_row_setup<OP>(a, y, yi, fy);
    const uint32_t v = px_any<C, OP>(a, src, x, y, yi, fy);
    uint8_t* o = dst + ((long long)y * a.dw + x) * C;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) o[ch] = (uint8_t)(v >> (8 * ch));
}
// Four consecutive destination pixels of a row per lane, written as C dwords through the streaming store path (round 6): the byte
// stores of the kernel above — three per pixel for RGB — cost far more than the arithmetic on the large outputs (1080p -> 720p, 2x
// downscales, upscales: 2.5-5x the time of a copy of the same bytes, profiles/r06zg_resize_u8_modes.txt).  dw % 4 == 0 and 4-byte
// aligned destination images (host-checked); a wave is one row of the 256 x 4 tile.
template <int C, int OP>
__global__ __launch_bounds__(kBx* kBy) void resize_u8_quads_kernel(Rz a) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int x0 = (bx_ * kBx + threadIdx.x) * 4;
    const int y = by_ * kBy + __builtin_amdgcn_readfirstlane(threadIdx.y);
    if (y >= a.dh) return;   // wave-uniform
    const uint8_t* __restrict__ src = a.src + (long long)bz_ * a.ss;
    const __amdgpu_buffer_rsrc_t ow = stream_window(a.dst + (long long)bz_ * a.ds + (long long)y * a.dw * C, (long long)a.dw * C);
    if (x0 >= a.dw) return;
    if constexpr (OP == kRzDown2 && C == 3) {
        // RGB exact 2x (round 6, later): the per-pixel form above costs 24 byte loads per output pixel; here a lane's four outputs come from six
        // dword loads per source row (eight pixels = 24 bytes), de-interleaved into channel dwords (kh_common.h::deinterleave_quad), the
        // 2 x 2 sums on packed bytes as for one channel, and three dwords out.  Same integers: (a + b + c + d + 2) >> 2.
        const uint8_t* r0 = src + ((long long)(2 * y) * a.sw + 2 * x0) * 3;
        const uint8_t* r1 = r0 + (long long)a.sw * 3;
        constexpr uint32_t kM = 0x00ff00ffu;
        uint32_t t[6], b[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { t[k] = *reinterpret_cast<const u32_unaligned*>(r0 + 4 * k); b[k] = *reinterpret_cast<const u32_unaligned*>(r1 + 4 * k); }
        uint32_t tA[3], tB[3], bA[3], bB[3], pl[3], w[3];
        deinterleave_quad<3>(t, tA); deinterleave_quad<3>(t + 3, tB);
        deinterleave_quad<3>(b, bA); deinterleave_quad<3>(b + 3, bB);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t s0 = (tA[c] & kM) + ((tA[c] >> 8) & kM) + (bA[c] & kM) + ((bA[c] >> 8) & kM) + 0x00020002u;   // outputs 0, 1 in 16-bit lanes
            const uint32_t s1 = (tB[c] & kM) + ((tB[c] >> 8) & kM) + (bB[c] & kM) + ((bB[c] >> 8) & kM) + 0x00020002u;   // outputs 2, 3
            pl[c] = __builtin_amdgcn_perm((s1 >> 2) & kM, (s0 >> 2) & kM, 0x06040200u);
        }
        interleave_quad<3>(pl, w);
        row_store<3>(ow, x0 * 3, w, a.plain);
        return;
    } else if constexpr (OP == kRzDown2) {
        // Exact 2x downscale of 1 / 4 channels (round 6).  The reference has the box special case for RGB only; its generic Q14 bilinear
        // at this scale has fx = fy = 8192 for every pixel (s = 2 i + 0.5, no clamp reached), and ((p00 + p01) 2^13 2^13 + (p10 + p11) 2^13 2^13
        // + 2^27) >> 28 == (p00 + p01 + p10 + p11 + 2) >> 2 exactly: the 2 x 2 box on packed bytes (two bytes per 16-bit lane) instead
        // of four u64 multiply-accumulates per channel.  The byte-pair sums stay below 2^10.
        const uint8_t* r0 = src + ((long long)(2 * y) * a.sw + 2 * x0) * C;
        const uint8_t* r1 = r0 + (long long)a.sw * C;
        constexpr uint32_t kM = 0x00ff00ffu;
        if constexpr (C == 1) {
            const uint64_t t = *reinterpret_cast<const u64_unaligned*>(r0), b = *reinterpret_cast<const u64_unaligned*>(r1);
            uint32_t o2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // four source pixels of both rows -> two outputs in 16-bit lanes
                const uint32_t tv = (uint32_t)(t >> (32 * h)), bv = (uint32_t)(b >> (32 * h));
                const uint32_t sum = (tv & kM) + ((tv >> 8) & kM) + (bv & kM) + ((bv >> 8) & kM) + 0x00020002u;
                o2[h] = (sum >> 2) & kM;
            }
            const uint32_t w[1] = {__builtin_amdgcn_perm(o2[1], o2[0], 0x06040200u)};
            row_store<1>(ow, x0, w, a.plain);
        } else {
            static_assert(C == 4, "exact-2x box: 1, 3 or 4 channels");
            uint32_t p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // two source pixels of both rows -> one output pixel, even / odd channel bytes in 16-bit lanes
                const uint64_t t = *reinterpret_cast<const u64_unaligned*>(r0 + 8 * j), b = *reinterpret_cast<const u64_unaligned*>(r1 + 8 * j);
                const uint32_t t0 = (uint32_t)t, t1 = (uint32_t)(t >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
                const uint32_t ev = (t0 & kM) + (t1 & kM) + (b0 & kM) + (b1 & kM) + 0x00020002u;
                const uint32_t od = ((t0 >> 8) & kM) + ((t1 >> 8) & kM) + ((b0 >> 8) & kM) + ((b1 >> 8) & kM) + 0x00020002u;
                p[j] = ((ev >> 2) & kM) | (((od >> 2) & kM) << 8);
            }
            store_quad_u8<C>(ow, x0, p, a.plain);
        }
        return;
    } else {
        int yi;
        uint32_t fy;
        px_row_setup<OP>(a, y, yi, fy);
        uint32_t p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = px_any<C, OP>(a, src, x0 + j, y, yi, fy);
        store_quad_u8<C>(ow, x0, p, a.plain);
    }
}

// ---- nearest UPSCALE of one-channel images (round 6) ---------------------------------------------------------------------------------
// Label maps and masks are upscaled with nearest; the quad kernel above spends four byte gathers and four f64 index computations per lane
// and ran at 0.25 of peak (1080p -> 4K gray 0.083 ms per 16 planes).  The column mapping does not depend on the row: a lane owns SIXTEEN
// destination columns for a strip of rows, evaluates the reference's column index for them ONCE (the same f64 expression), and turns it
// into byte selectors over the sixteen source bytes at its first column (kh_common.h::remap16_*: with a horizontal step <= 1 the four
// sources of a destination dword lie within four bytes); per destination row it then does one 16-byte load, twelve v_perm_b32 and one
// 16-byte store.  CV = the cv2-compatible index (floor(i * iscale)) instead of floor((i + 0.5) * scale).  Any width; sw >= 16, dw >= sw.
constexpr int kNuRows = 32;   // destination rows per block strip
// C = 3: the same on the ROW BYTES of an interleaved image — destination byte B is channel B % 3 of pixel B / 3 and comes from source byte
// 3 col(B / 3) + B % 3; the four sources of a destination dword still lie within four bytes, and the launcher checks that the sixteen
// bytes of every lane span at most sixteen source bytes (upscales of about 1.5x and more).
template <int C, bool CV>
__global__ __launch_bounds__(256) void nearest_up_gray_kernel(Rz a) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int rowb = a.dw * C, srowb = a.sw * C;
    const int B0 = ((int)bx_ * 256 + (int)threadIdx.x) * 16;   // this lane's first destination byte of a row
    if (B0 >= rowb) return;
    const uint8_t* __restrict__ src = a.src + (long long)bz_ * a.ss;
    uint8_t* __restrict__ dst = a.dst + (long long)bz_ * a.ds;
    auto sbyte = [&](int B) {   // source byte of destination byte B (clamped to the row: lanes past its end are never stored)
        const int Bc = min(B, rowb - 1), X = C == 1 ? Bc : Bc / C, ch = C == 1 ? 0 : Bc - X * C;
        return (CV ? cv_nearest_index(X, a.scale_x, a.sw) : nearest_index(X, a.scale_x, a.sw)) * C + ch;
    };
    // the window starts at the first byte of B0's source PIXEL: a later destination pixel that repeats it reaches back to its channel 0
    const int pc = min(sbyte(B0) - (C == 1 ? 0 : B0 % C), srowb - 16);
    const Remap16 rm = remap16_setup(B0, pc, sbyte);
    const int nvalid = min(rowb - B0, 16);
    const int y0 = (int)by_ * kNuRows, y1 = min(y0 + kNuRows, a.dh);
    const __amdgpu_buffer_rsrc_t ow = stream_window(dst, (long long)rowb * a.dh);   // (row bytes * dh < 2^31: host-checked)
    for (int y = y0; y < y1; ++y) {
        const int sy = CV ? cv_nearest_index(y, a.scale_y, a.sh) : nearest_index(y, a.scale_y, a.sh);   // block-uniform
        const u32x4_t v = *reinterpret_cast<const u32x4_unaligned*>(src + (long long)sy * srowb + pc);
        uint32_t L[4] = {v.x, v.y, v.z, v.w};
        remap16_apply(rm, L, 0u);
        const int off = y * rowb + B0;
        if (nvalid == 16 && a.plain != 2) row_store<4>(ow, off, L, a.plain);
        else if (nvalid == 16) *reinterpret_cast<u32x4_unaligned*>(dst + off) = u32x4_t{L[0], L[1], L[2], L[3]};
        else store_head_bytes(dst + off, L, nvalid);
    }
}

// ---- separable Q14 ------------------------------------------------------------------------------------
// k taps per destination sample; `w` rows of kp = roundup(k, 4) i16 weights (zero padded).  aofs / w8 / kp8 / span8: the aligned windows,
// byte-split weights, padded tap count and widest tile span of sep_h_u8_dot4_kernel; vty / vrows: segment height and staged rows of
// sep_v_u8_lds_kernel (see get_tab).
struct SepTab { const int32_t* ofs; const int16_t* w; int k, kp; const int32_t* aofs; const uint32_t* w8; int kp8, span8, vty, vrows; };

// horizontal_row_scalar (kernels.rs:403-425): (x, source row) -> i16
template <int C>
__global__ __launch_bounds__(kBx* kBy) void sep_h_u8_kernel(Rz a, int16_t* __restrict__ hbuf, SepTab tx) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int x = bx_ * kBx + threadIdx.x, sy = by_ * kBy + threadIdx.y;
    if (x >= a.dw || sy >= a.sh) return;
    const uint8_t* __restrict__ row = a.src + (long long)bz_ * a.ss + (long long)sy * a.sw * C;
    const int x0 = tx.ofs[x];
    const int16_t* w = tx.w + (long long)x * tx.kp;
    int32_t acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) acc[ch] = 0;
    for (int t = 0; t < tx.k; ++t) {
        const int sx = min(max(x0 + t, 0), a.sw - 1);  // build_xsrc_lut, common.rs:127-137
        const int32_t wt = w[t];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) acc[ch] += (int32_t)row[sx * C + ch] * wt;
    }
    int16_t* o = hbuf + (((long long)bz_ * a.sh + sy) * a.dw + x) * C;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) o[ch] = (int16_t)min(max((acc[ch] + 8192) >> 14, -32768), 32767);
}

// The LDS-staged horizontal pass works on tiles of kSepTX destination columns x kSepRows source rows (256 threads: a thread owns one
// column for four rows).  Round 2's version of it kept the source bytes INTERLEAVED in LDS (v_alignbyte + v_perm + v_dot2 per four
// taps: 1.53 ms per 256 1080p -> 224 Lanczos frames, profiles/r04f); the planar v_dot4 kernel below replaced it.
constexpr int kSepTX = 64, kSepRows = 16, kSepRpt = 2;   // kSepRpt: rows per thread of the v_dot4 kernel (r04t: 2 vs 4)
extern __shared__ __attribute__((aligned(16))) uint8_t kh_sep_lds[];

typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int32_t dot2_i16(uint32_t a, uint32_t b, int32_t c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2_t, a), __builtin_bit_cast(i16x2_t, b), c, false);
}

// horizontal pass, planar i8 + v_dot4 (round 4).  profiles/r04a_limiter_resize_u8.txt: round 2's tile kernel spent 1849 vector instructions per
// wave-tile — 3.4 per multiply-add — and 75 % of its LDS cycles in bank conflicts: per four taps of one row it reads C + 1 dwords of
// INTERLEAVED bytes, re-aligns them (v_alignbyte), picks pixel pairs out per channel (v_perm) and only then multiplies (v_dot2).  Here
// the staging pass de-interleaves once: LDS holds each source row as C PLANES of signed bytes p - 128, and a destination column's
// taps are read as aligned 8-byte runs of ONE plane that feed v_dot4_i32_i8 directly:
//   * the tap window of column x is widened down to a multiple of 8 source pixels (aofs[x] = ofs[x] & ~7) and its weight row is
//     shifted to match, the new leading / trailing taps being ZERO weights (kp8 = roundup8(k + 7) taps): every LDS read is an
//     aligned ds_read_b64, no v_alignbyte, no v_perm;
//   * a Q14 weight w is split on the host into w = 256 * wh + wl with wl, wh signed bytes, so that
//         sum p w = 256 * dot4(p - 128, wh) + dot4(p - 128, wl) + 128 * sum w,   sum w == 16384 exactly (precompute_contribs),
//     all in i32 — the same integer as the reference's scalar sum, so the i16 intermediate is byte-identical;
//   * one 16-byte weight load (8 wl + 8 wh) serves the thread's four rows x C channels: 4 v_dot4 per 8 taps per (row, channel) =
//     0.5 vector instructions per multiply-add.
// Tile, thread mapping and edge handling (per-byte staging with the reference's clamp) are those of the kernel above.
// bytes c, C + c, 2C + c, 3C + c of the 4 C-byte pixels in d[] — one plane's four samples — with v_perm_b32 (2 per plane for RGB, 3 for
// RGBA) instead of the shift / mask / or chains the generic expression compiles to (r04k: the staging arithmetic, not the dot
// products, kept the vector ALUs 86 % busy)
constexpr uint32_t rgb_sel1(int ch, int j) { return 3 * j + ch < 8 ? (uint32_t)(3 * j + ch) : 0u; }              // sample j of channel ch is byte 3 j + ch
constexpr uint32_t rgb_sel2(int ch, int j) { return 3 * j + ch < 8 ? (uint32_t)j : (uint32_t)(4 + 3 * j + ch - 8); }
template <int C, int CH>
__device__ __forceinline__ uint32_t plane4(const uint32_t (&d)[C]) {
    if constexpr (C == 1) {
        return d[0];
    } else if constexpr (C == 3) {
        constexpr uint32_t sel1 = rgb_sel1(CH, 0) | (rgb_sel1(CH, 1) << 8) | (rgb_sel1(CH, 2) << 16) | (rgb_sel1(CH, 3) << 24);   // from {d1 : d0}
        constexpr uint32_t sel2 = rgb_sel2(CH, 0) | (rgb_sel2(CH, 1) << 8) | (rgb_sel2(CH, 2) << 16) | (rgb_sel2(CH, 3) << 24);   // from {d2 : t}
        return __builtin_amdgcn_perm(d[2], __builtin_amdgcn_perm(d[1], d[0], sel1), sel2);
    } else {
        static_assert(C == 4, "1, 3 or 4 channels");
        constexpr uint32_t lo = (uint32_t)CH | ((uint32_t)(4 + CH) << 8);           // bytes CH of d0, d1 (and of d2, d3)
        const uint32_t t01 = __builtin_amdgcn_perm(d[1], d[0], lo), t23 = __builtin_amdgcn_perm(d[3], d[2], lo);
        return __builtin_amdgcn_perm(t23, t01, 0x05040100u);
    }
}
template <int C, int... CH>
__device__ __forceinline__ void store_planes(uint8_t* base, int pitchp, const uint32_t (&d)[C], std::integer_sequence<int, CH...>) {
    ((*reinterpret_cast<uint32_t*>(base + CH * pitchp) = plane4<C, CH>(d) ^ 0x80808080u), ...);
}

// PITCH (bytes per plane row in LDS) is a template constant so that the twelve (row, plane) reads of a step are ONE address register plus
// immediate offsets.
// RPT rows per thread: a block is 64 x (kSepRows / RPT) threads.  RPT = 2 (512 threads, 8 waves) keeps twice the waves resident
// per staged tile — the tile's LDS, not registers, limits the blocks per CU — at the price of reading each weight twice.
template <int C, int PITCH, int RPT>
__global__ __launch_bounds__(64 * (kSepRows / RPT)) void sep_h_u8_dot4_kernel(Rz a, int16_t* __restrict__ hbuf, SepTab tx) {
    constexpr int pitchp = PITCH, NW = kSepRows / RPT, NT = 64 * NW;   // waves, threads per block
    uint8_t* S = kh_sep_lds;  // [kSepRows][C][pitchp] signed bytes (p - 128), then the tile's weights [steps][kSepTX] x 16 B
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int X0 = bx_ * kSepTX, sy0 = by_ * kSepRows;
    const int p0a = tx.aofs[X0], span = tx.aofs[min(X0 + kSepTX, a.dw) - 1] + tx.kp8 - p0a;   // block-uniform, multiples of 8
    const int nrows = min(kSepRows, a.sh - sy0), steps = tx.kp8 >> 3;
    const uint8_t* __restrict__ src = a.src + (long long)bz_ * a.ss + (long long)sy0 * a.sw * C;
    u32x4_t* W = reinterpret_cast<u32x4_t*>(S + kSepRows * C * pitchp);
    // Staging is where this pass used to lose its time (r04f: 10 % of the vector ALUs busy): a loop of load -> LDS write round trips
    // per row, and a per-BYTE path for every tile that touches the left or right image border — half of the tiles of a 224-wide
    // destination.  Now a thread has all its wide loads in flight before the first LDS write, and only the few groups that actually
    // straddle a border take the clamped bytes.
    // thread -> (4-pixel group q = lane + 64 j, row NW i + wave): no per-item division; kQ x RPT independent wide loads in flight per thread
    constexpr int kQ = 3;
    const int nq = span >> 2;
    for (int q0 = lane; q0 < nq; q0 += 64 * kQ) {
        uint32_t d[RPT][kQ][C];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = min(NW * i + wave, nrows - 1);
#pragma unroll
            for (int j = 0; j < kQ; ++j) {
                const int q = min(q0 + 64 * j, nq - 1), px0 = p0a + 4 * q;
                const uint8_t* g = src + (uint32_t)((r * a.sw + min(max(px0, 0), a.sw - 4)) * C);   // always 4 whole pixels of the row (sw >= 4: host); 32-bit offsets: one image < 2^31 B (check_rz)
#pragma unroll
                for (int c = 0; c < C; ++c) d[i][j][c] = *reinterpret_cast<const u32_unaligned*>(g + 4 * c);
            }
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = NW * i + wave;
            if (r >= nrows) break;
#pragma unroll
            for (int j = 0; j < kQ; ++j) {
                const int q = q0 + 64 * j, px0 = p0a + 4 * q;
                if (q >= nq) break;
                uint8_t* o = S + r * (C * pitchp) + 4 * q;
                if (px0 >= 0 && px0 + 3 < a.sw) {
                    store_planes<C>(o, pitchp, d[i][j], std::make_integer_sequence<int, C>{});
                } else {   // a group across the image border: the reference's clamp, per pixel (build_xsrc_lut, common.rs:127-137)
                    const uint8_t* grow = src + (uint32_t)(r * a.sw * C);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        uint32_t v = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v |= (uint32_t)grow[min(max(px0 + k, 0), a.sw - 1) * C + c] << (8 * k);
                        *reinterpret_cast<uint32_t*>(o + c * pitchp) = v ^ 0x80808080u;
                    }
                }
            }
        }
    }
    for (int e = tid; e < kSepTX * steps; e += NT) {   // the tile's weight rows, transposed to [step][column]: contiguous in global memory
        const int xr = e / steps, st = e - xr * steps;
        W[st * kSepTX + xr] = reinterpret_cast<const u32x4_t*>(tx.w8)[(uint32_t)(min(X0 + xr, a.dw - 1) * steps + st)];
    }
    __syncthreads();
    const int x = X0 + lane;
    if (x >= a.dw) return;
    const int rel = tx.aofs[x] - p0a;   // multiple of 8
    // rows past the image (a ragged last tile) read LDS rows nobody staged — whatever they hold is finite integer data, and the
    // results are not stored — so the four rows of a thread are always base + i * C * PITCH
    const uint8_t* base = kh_sep_lds + wave * RPT * (C * pitchp) + rel;
    int32_t al[RPT][C], ah[RPT][C];
#pragma unroll
    for (int i = 0; i < RPT; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) { al[i][c] = (128 << 14) + 8192; ah[i][c] = 0; }   // 128 * sum(w) and the rounding constant ride in the low sum
    // NOT unrolled: with two steps in one body the compiler fuses the two 8-byte reads of a (row, plane) into one ds_read_b128 at an
    // address that is only 8-byte aligned — legal, and measured at 700 M unaligned-stall cycles per launch (SQ_LDS_UNALIGNED_STALL,
    // profiles/r04h): 43 % of the wave-cycles waiting on the LDS queue.
#pragma unroll 1
    for (int s8 = 0; s8 < steps; ++s8) {
        const u32x4_t w = W[s8 * kSepTX + lane];   // {wl[0..3], wl[4..7], wh[0..3], wh[4..7]}
        const uint8_t* pb = base + 8 * s8;
#pragma unroll
        for (int i = 0; i < RPT; ++i)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const u32x2_t p = *reinterpret_cast<const u32x2_t*>(pb + (i * C + c) * pitchp);
                al[i][c] = __builtin_amdgcn_sdot4((int)p.x, (int)w.x, al[i][c], false);
                al[i][c] = __builtin_amdgcn_sdot4((int)p.y, (int)w.y, al[i][c], false);
                ah[i][c] = __builtin_amdgcn_sdot4((int)p.x, (int)w.z, ah[i][c], false);
                ah[i][c] = __builtin_amdgcn_sdot4((int)p.y, (int)w.w, ah[i][c], false);
            }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = wave * RPT + i;
        if (r >= nrows) break;
        int16_t* o = hbuf + (long long)bz_ * a.sh * a.dw * C + (uint32_t)(((sy0 + r) * a.dw + x) * C);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            o[ch] = (int16_t)min(max((ah[i][ch] * 256 + al[i][ch]) >> 14, -32768), 32767);
        }
    }
}

// vertical_row_scalar (kernels.rs:699-708): thread = one flat i16 column of the intermediate
__global__ __launch_bounds__(kBx* kBy) void sep_v_u8_kernel(Rz a, const int16_t* __restrict__ hbuf, SepTab ty, int hrow) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int i = bx_ * kBx + threadIdx.x, y = by_ * kBy + threadIdx.y;
    if (i >= hrow || y >= a.dh) return;
    const int16_t* __restrict__ h = hbuf + (long long)bz_ * a.sh * hrow + i;
    const int y0 = ty.ofs[y];
    const int16_t* w = ty.w + (long long)y * ty.kp;
    int32_t acc = 0;
    for (int k = 0; k < ty.k; ++k) {
        const int sy = min(max(y0 + k, 0), a.sh - 1);
        acc += (int32_t)h[(long long)sy * hrow] * (int32_t)w[k];
    }
    a.dst[(long long)bz_ * a.ds + (long long)y * hrow + i] = (uint8_t)min(max((acc + 8192) >> 14, 0), 255);
}

// vertical pass, LDS-staged (round 4).  The kernel above reads every tap from global memory: 1.6 vector loads per multiply-add, each a
// 2-byte gather a whole intermediate row apart, 13 instructions per tap (profiles/r04a_limiter_resize_u8.txt: 0.55 ms for 1.1 G
// multiply-adds).  Here a block owns 128 flat columns x a SEGMENT of ty.vty destination rows: the i16 rows that segment taps
// (ofs[Y0] .. ofs[Y1 - 1] + k, each clamped to the image like the reference's row index) are staged in LDS once, two columns per
// dword; a wave then walks its destination rows with one conflict-free ds_read_b32 + two v_dot2 per tap (weights are wave-uniform).
// i32 sums of the same products: byte-identical.
constexpr int kSepVRows = 192;   // staged intermediate rows per block (96 row pairs x 512 B = 48 KiB)
// Second form (r04zw): a dword of the staged window holds one COLUMN's values of two consecutive ROWS, so that one v_dot2_i32_i16
// multiplies two taps — with a column pair per dword (the first form) every v_dot2 carried a zero weight and the tap loop spent 6
// vector + 4 LDS instructions per two taps of two columns; now 2 + 3.  A destination row whose first tap sits on an odd window row
// takes its weights shifted by one slot ({0, w0}, {w1, w2}, ...): the weight rows are laid out per destination row accordingly.
// A lane owns columns `lane` and `lane + 64` of the block's 128.  i32 sums of the same products: byte-identical.
__global__ __launch_bounds__(256) void sep_v_u8_lds_kernel(Rz a, const int16_t* __restrict__ hbuf, SepTab ty, int hrow) {
    uint32_t* S = reinterpret_cast<uint32_t*>(kh_sep_lds);   // [row pairs][128] dwords (dynamic: the launch asks for what this table needs)
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int colb = bx_ * 128;                                  // the block's first flat column
    const int Y0 = by_ * ty.vty, Y1 = min(Y0 + ty.vty, a.dh);
    const int r0 = ty.ofs[Y0], nr = ty.ofs[Y1 - 1] + ty.k - r0;  // block-uniform; nr <= kSepVRows (host-checked per table)
    const int npairs = (nr + 1) >> 1;
    const int16_t* __restrict__ h = hbuf + (long long)bz_ * a.sh * hrow;
    // staging: lane = column pair (colb + 2 lane, + 1) of one row pair; sixteen loads (eight row pairs) in flight per lane
    const int col = colb + 2 * lane, c0 = min(col, hrow - 1), c1 = min(col + 1, hrow - 1);
    const bool pairs = (hrow & 1) == 0;   // rows start dword-aligned: one load per (row, column pair) instead of two 2-byte loads
    const int cp = min(col, max(hrow - 2, 0));   // even (lanes past the row re-read its last pair: row + hrow - 1 alone is out of bounds)
    // Everything the block needs from memory is requested before the first wait: up to sixteen row pairs per lane (thirty-two loads;
    // 128 window rows, the usual segment has ~100), then the weight rows' table reads; a taller window takes a second trip.
    const int wpr = (ty.kp >> 1) + 1;
    uint32_t* Wp = S + (((ty.vrows + 1) >> 1) + 2) * 128;   // after two spare row pairs (finite data for the zero-weight slots past the window)
    // weight rows, one per destination row of the segment, as dwords {w[2 j - p], w[2 j + 1 - p]} with p = parity of the row's
    // first window row (slots outside 0 .. k - 1 are zero): kp / 2 + 1 dwords each.  EVERY thread of the block fills its share —
    // a wave with no row pair to stage (wave >= npairs: a one-row window) still runs it, after the loop below.
    auto fill_weights = [&]() {
        for (int e = threadIdx.x; e < (Y1 - Y0) * wpr; e += 256) {
            const int yy = e / wpr, j = e - yy * wpr, y = Y0 + yy, p = (ty.ofs[y] - r0) & 1;
            const int16_t* w = ty.w + (long long)y * ty.kp;
            const int t0 = 2 * j - p, t1 = t0 + 1;
            const uint32_t lo = (t0 >= 0 && t0 < ty.k) ? (uint16_t)w[t0] : 0u, hi = (t1 >= 0 && t1 < ty.k) ? (uint16_t)w[t1] : 0u;
            Wp[e] = lo | (hi << 16);
        }
    };
    S[npairs * 128 + threadIdx.x] = 0u;   // the two spare row pairs: the tap loop's rounded-up last trip reads one of them (zero weights)
    bool weights_done = false;
    for (int pb = wave; pb < npairs; pb += 4 * 16) {
        uint32_t v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int wr = min(2 * (pb + 4 * (j >> 1)) + (j & 1), nr - 1);                                 // window row (the last pair may repeat the last row)
            const int16_t* row = h + (long long)min(max(r0 + wr, 0), a.sh - 1) * hrow;                      // vertical_row_scalar's clamp, kernels.rs:699-708
            if (pairs) v[j] = *reinterpret_cast<const uint32_t*>(row + cp);
            else v[j] = (uint32_t)(uint16_t)row[c0] | ((uint32_t)(uint16_t)row[c1] << 16);
        }
        if (!weights_done) {   // behind the first trip's loads: the table reads join them in flight
            fill_weights();
            weights_done = true;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int rp = pb + 4 * j;
            if (rp < npairs) {   // (even row, odd row) of column 2 lane, then of column 2 lane + 1: 8 bytes, adjacent columns
                const uint32_t e = v[2 * j], o = v[2 * j + 1];
                *reinterpret_cast<u32x2_t*>(&S[rp * 128 + 2 * lane]) =
                    u32x2_t{__builtin_amdgcn_perm(o, e, 0x05040100u), __builtin_amdgcn_perm(o, e, 0x07060302u)};
            }
        }
    }
    if (!weights_done) fill_weights();
    __syncthreads();
    for (int y = Y0 + wave; y < Y1; y += 4) {
        const int s0 = ty.ofs[y] - r0, p = s0 & 1, nj = (ty.k + p + 1) >> 1;
        const uint32_t* tap = S + (s0 >> 1) * 128 + lane;
        const uint32_t* w = Wp + (y - Y0) * wpr;
        int32_t acc0 = 0, acc1 = 0;
        for (int j = 0; j < nj; j += 2) {   // two row pairs per trip: independent LDS reads in flight (slot nj, if read, has zero weights)
            const uint32_t a0 = tap[j * 128], b0 = tap[j * 128 + 64], a1 = tap[(j + 1) * 128], b1 = tap[(j + 1) * 128 + 64];
            const uint32_t w0 = w[j], w1 = j + 1 < wpr ? w[j + 1] : 0u;
            acc0 = dot2_i16(a0, w0, acc0); acc1 = dot2_i16(b0, w0, acc1);
            acc0 = dot2_i16(a1, w1, acc0); acc1 = dot2_i16(b1, w1, acc1);
        }
        uint8_t* o = a.dst + (long long)bz_ * a.ds + (long long)y * hrow + colb;
        if (colb + lane < hrow) o[lane] = (uint8_t)min(max((acc0 + 8192) >> 14, 0), 255);
        if (colb + lane + 64 < hrow) o[lane + 64] = (uint8_t)min(max((acc1 + 8192) >> 14, 0), 255);
    }
}

// ---- fused RGB8 -> normalised CHW f32 (P/resize/fused.rs) ---------------------------------------------
struct Norm3 { float scale[3], bias[3]; };

// MODE 0 nearest (:885-936), 1 bilinear (:147-232 + blerp :285-289), 2 exact-2x box (:528-558)
template <int MODE>
__global__ __launch_bounds__(kBx* kBy) void fused_rgb_chw_kernel(Rz a, Norm3 n, float scale_xf, float scale_yf) {
    KH_RZ_PROLOGUE
    float* __restrict__ out = reinterpret_cast<float*>(a.dst) + (long long)bz_ * a.ds + (long long)y * a.dw + x;
    (void)dst;
    const long long plane = (long long)a.dw * a.dh;
    float v[3];
    if constexpr (MODE == 0) {
        const uint8_t* p = src + ((long long)nearest_index(y, a.scale_y, a.sh) * a.sw + nearest_index(x, a.scale_x, a.sw)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (float)p[c] * n.scale[c] + n.bias[c];
    } else if constexpr (MODE == 2) {
        const uint8_t* r0 = src + ((long long)(2 * y) * a.sw + 2 * x) * 3;
        const uint8_t* r1 = r0 + (long long)a.sw * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t sum = (uint32_t)r0[c] + r0[3 + c] + r1[c] + r1[3 + c];
            v[c] = (float)sum * (n.scale[c] * 0.25f) + n.bias[c];
        }
    } else {
        const float fy = fmaxf(((float)y + 0.5f) * scale_yf - 0.5f, 0.0f);
        const float fx = fmaxf(((float)x + 0.5f) * scale_xf - 0.5f, 0.0f);
        const int y0 = min((int)fy, a.sh - 1), y1 = min(y0 + 1, a.sh - 1);
        const int x0 = min((int)fx, a.sw - 1), x1 = min(x0 + 1, a.sw - 1);
        const float wy = fy - (float)y0, w = fx - (float)x0;
        // x1 = min(x0 + 1, sw - 1): load_quad_u8's second pixel
        const QuadU8 t = load_quad_u8<3>(src + (unsigned)(y0 * a.sw) * 3, src + (unsigned)(y1 * a.sw) * 3, x0, a.sw);
        (void)x1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p = (float)chan_u8(t.p00, c), q = (float)chan_u8(t.p01, c), r = (float)chan_u8(t.p10, c), s = (float)chan_u8(t.p11, c);
            const float top = p + w * (q - p), bot = r + w * (s - r);
            v[c] = (top + wy * (bot - top)) * n.scale[c] + n.bias[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * plane] = v[c];
}

// vertical pass of the separable variant (:1003-1035): i32 accumulate, acc * (scale / 2^14) + bias
__global__ __launch_bounds__(kBx* kBy) void fused_sep_v_kernel(Rz a, const int16_t* __restrict__ hbuf, SepTab ty, Norm3 n) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int x = bx_ * kBx + threadIdx.x, y = by_ * kBy + threadIdx.y;
    if (x >= a.dw || y >= a.dh) return;
    const int hrow = a.dw * 3;
    const int16_t* __restrict__ h = hbuf + (long long)bz_ * a.sh * hrow + x * 3;
    const int y0 = ty.ofs[y];
    const int16_t* w = ty.w + (long long)y * ty.kp;
    int32_t acc[3] = {0, 0, 0};
    for (int k = 0; k < ty.k; ++k) {
        const int sy = min(max(y0 + k, 0), a.sh - 1);
        const int32_t wt = w[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += (int32_t)h[(long long)sy * hrow + c] * wt;
    }
    float* out = reinterpret_cast<float*>(a.dst) + (long long)bz_ * a.ds + (long long)y * a.dw + x;
    const long long plane = (long long)a.dw * a.dh;
    const float inv_q = 1.0f / 16384.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * plane] = (float)acc[c] * (n.scale[c] * inv_q) + n.bias[c];
}

// ---- OpenCV-compatible (opencv_compat.rs) -------------------------------------------------------------
struct LinTap { int ofs; bool border; float w0, w1; int i0, i1; };
__device__ __forceinline__ LinTap linear_tap(int d, double scale, int src_len) {  // linear_axis, :22-65
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    const float fl = floorf(fx);
    long long sx = (long long)fl;
    fx -= (float)sx;
    LinTap t;
    t.border = false;
    if (sx < 0) { sx = 0; fx = 0.0f; }
    if (sx >= (long long)src_len - 1) { sx = (long long)src_len - 1; fx = 0.0f; t.border = true; }
    t.ofs = (int)sx;
    t.w0 = 1.0f - fx;
    t.w1 = fx;
    t.i0 = (int)rintf((1.0f - fx) * 2048.0f);  // round_ties_even
    t.i1 = (int)rintf(fx * 2048.0f);
    return t;
}

template <typename T, int C>
__global__ __launch_bounds__(kBx* kBy) void cv_nearest_kernel(Rz a) {
    KH_RZ_PROLOGUE
    const int sx = cv_nearest_index(x, a.scale_x, a.sw), sy = cv_nearest_index(y, a.scale_y, a.sh);
    const T* p = reinterpret_cast<const T*>(src) + ((long long)sy * a.sw + sx) * C;
    T* o = reinterpret_cast<T*>(dst) + ((long long)y * a.dw + x) * C;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) o[ch] = p[ch];
}

template <int C>
__device__ __forceinline__ uint32_t px_cv_linear_u8(const Rz& a, const uint8_t* __restrict__ src, int x, const LinTap& ty) {  // resize_linear_u8, :139-195
    const LinTap tx = linear_tap(x, a.scale_x, a.sw);
    const int sy1 = min(ty.ofs + 1, a.sh - 1);
    // border columns (ofs == sw - 1) only use the first pixel
    const QuadU8 q = load_quad_u8<C>(src + (unsigned)(ty.ofs * a.sw) * C, src + (unsigned)(sy1 * a.sw) * C, tx.ofs, a.sw);
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int32_t p00 = (int32_t)chan_u8(q.p00, k), p01 = (int32_t)chan_u8(q.p01, k), p10 = (int32_t)chan_u8(q.p10, k), p11 = (int32_t)chan_u8(q.p11, k);
        int32_t s0, s1;
        if (tx.border) { s0 = p00 << 11; s1 = p10 << 11; }
        else { s0 = p00 * tx.i0 + p01 * tx.i1; s1 = p10 * tx.i0 + p11 * tx.i1; }
        out |= ((uint32_t)((((ty.i0 * (s0 >> 4)) >> 16) + ((ty.i1 * (s1 >> 4)) >> 16) + 2) >> 2) & 0xffu) << (8 * k);
    }
    return out;
}
template <int C>
__global__ __launch_bounds__(kBx* kBy) void cv_linear_u8_kernel(Rz a) {
    KH_RZ_PROLOGUE
    const uint32_t v = px_cv_linear_u8<C>(a, src, x, linear_tap(y, a.scale_y, a.sh));
    uint8_t* o = dst + ((long long)y * a.dw + x) * C;
#pragma unroll
    for (int k = 0; k < C; ++k) o[k] = (uint8_t)(v >> (8 * k));
}
// u8, four pixels per lane with dword stores (see resize_u8_quads_kernel); LINEAR = false: INTER_NEAREST
template <int C, bool LINEAR>
__global__ __launch_bounds__(kBx* kBy) void cv_u8_quads_kernel(Rz a) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int x0 = (bx_ * kBx + threadIdx.x) * 4;
    const int y = by_ * kBy + __builtin_amdgcn_readfirstlane(threadIdx.y);
    if (y >= a.dh) return;   // wave-uniform
    const uint8_t* __restrict__ src = a.src + (long long)bz_ * a.ss;
    const __amdgpu_buffer_rsrc_t ow = stream_window(a.dst + (long long)bz_ * a.ds + (long long)y * a.dw * C, (long long)a.dw * C);
    if (x0 >= a.dw) return;
    uint32_t p[4];
    if constexpr (LINEAR) {
        const LinTap ty = linear_tap(y, a.scale_y, a.sh);
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = px_cv_linear_u8<C>(a, src, x0 + j, ty);
    } else {
        const int sy = cv_nearest_index(y, a.scale_y, a.sh);
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = load_px_u8<C>(src + ((long long)sy * a.sw + cv_nearest_index(x0 + j, a.scale_x, a.sw)) * C);
    }
    store_quad_u8<C>(ow, x0, p, a.plain);
}

template <int C>
__global__ __launch_bounds__(kBx* kBy) void cv_linear_f32_kernel(Rz a) {  // resize_linear_f32, :197-250
    KH_RZ_PROLOGUE
    const LinTap tx = linear_tap(x, a.scale_x, a.sw), ty = linear_tap(y, a.scale_y, a.sh);
    const int sy1 = min(ty.ofs + 1, a.sh - 1);
    const float* r0 = reinterpret_cast<const float*>(src) + ((long long)ty.ofs * a.sw + tx.ofs) * C;
    const float* r1 = reinterpret_cast<const float*>(src) + ((long long)sy1 * a.sw + tx.ofs) * C;
    float* o = reinterpret_cast<float*>(dst) + ((long long)y * a.dw + x) * C;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const float s0 = tx.border ? r0[k] : r0[k] * tx.w0 + r0[C + k] * tx.w1;
        const float s1 = tx.border ? r1[k] : r1[k] * tx.w0 + r1[C + k] * tx.w1;
        o[k] = s0 * ty.w0 + s1 * ty.w1;
    }
}

// ---- host side ------------------------------------------------------------------------------------------

// FilterKind (common.rs:11-47): 0 = Cubic (a = -0.5), 1 = Lanczos3
double filt_weight(int filt, double x) {
    const double ax = fabs(x);
    if (filt == 0) {
        const double a = -0.5;
        if (ax < 1.0) return (a + 2.0) * ax * ax * ax - (a + 3.0) * ax * ax + 1.0;
        if (ax < 2.0) return a * ax * ax * ax - 5.0 * a * ax * ax + 8.0 * a * ax - 4.0 * a;
        return 0.0;
    }
    if (ax < 1e-12) return 1.0;
    if (ax < 3.0) {
        const double px = 3.14159265358979323846 * x;
        return 3.0 * sin(px) * sin(px / 3.0) / (px * px);
    }
    return 0.0;
}

// precompute_contribs (common.rs:62-125) + pack_xw_i16 (:141-143)
int build_contribs(int src_size, int dst_size, int filt, bool antialias, std::vector<int32_t>& offsets,
                   std::vector<int16_t>& weights) {
    const double scale = (double)src_size / (double)dst_size;
    const double filt_scale = antialias ? (scale > 1.0 ? scale : 1.0) : 1.0;
    const double support = (filt == 0 ? 2.0 : 3.0) * filt_scale;
    int ksize = (int)ceil(support) * 2;
    if (ksize < 2) ksize = 2;
    offsets.assign(dst_size, 0);
    weights.assign((size_t)dst_size * ksize, 0);
    std::vector<double> raw(ksize);
    std::vector<int32_t> qw(ksize);
    const double inv_filt_scale = 1.0 / filt_scale;
    for (int i = 0; i < dst_size; ++i) {
        const double center = ((double)i + 0.5) * scale - 0.5;
        const long long left = (long long)ceil(center - support);
        offsets[i] = (int32_t)left;
        double sum = 0.0;
        for (int k = 0; k < ksize; ++k) {
            const double x = (double)(left + k) - center;
            const double w = filt_weight(filt, x * inv_filt_scale) * inv_filt_scale;
            raw[k] = w;
            sum += w;
        }
        int qsum = 0;
        const double norm = fabs(sum) > 1e-12 ? 16384.0 / sum : 0.0;
        for (int k = 0; k < ksize; ++k) {
            qw[k] = (int32_t)round(raw[k] * norm);
            qsum += qw[k];
        }
        if (qsum != 16384) {
            int max_k = 0, max_abs = 0;
            for (int k = 0; k < ksize; ++k)
                if (abs(qw[k]) > max_abs) { max_abs = abs(qw[k]); max_k = k; }
            qw[max_k] += 16384 - qsum;
        }
        for (int k = 0; k < ksize; ++k) weights[(size_t)i * ksize + k] = (int16_t)qw[k];
    }
    return ksize;
}

// Contribution tables live on the device, keyed like the reference's per-(device, geometry) cache (P/resize/cuda.rs:151-190).
// A table is uploaded with a blocking copy into a fresh allocation BEFORE it is published, so no stream ever observes a
// half-written table; eviction (LRU beyond 256 geometries) never frees a table a launch may still read — kh_table_cache.h.
// never destroyed: at process exit the HIP runtime may already be gone when static destructors run
TableCache<std::tuple<int, int, int, int, int>>& g_tabs = *new TableCache<std::tuple<int, int, int, int, int>>(256);

int32_t get_tab(int src_size, int dst_size, int filt, bool aa, hipStream_t stream, const char* what, SepTab& out, TableLease& lease) {
    int dev = 0;
    KH_HIP(hipGetDevice(&dev));
    const auto key = std::make_tuple(dev, src_size, dst_size, filt, (int)aa);
    const int32_t rc = g_tabs.lookup(key, stream, what, [&](DevTable& t) -> int32_t {
        std::vector<int32_t> ofs;
        std::vector<int16_t> wk, w;
        const int k = build_contribs(src_size, dst_size, filt, aa, ofs, wk), kp = (k + 3) & ~3;
        w.assign((size_t)dst_size * kp, 0);  // rows padded to a multiple of four taps with zero weights
        for (int i = 0; i < dst_size; ++i) std::copy(wk.begin() + (size_t)i * k, wk.begin() + (size_t)(i + 1) * k, w.begin() + (size_t)i * kp);
        // sep_h_u8_dot4_kernel: windows widened down to multiples of 8 source pixels, weights shifted to match (zero taps around them) and
        // split into signed bytes w = 256 wh + wl; per column and 8 taps: wl[0..7] then wh[0..7] (16 bytes)
        const int kp8 = (k + 7 + 7) & ~7, steps = kp8 / 8;
        std::vector<int32_t> aofs(dst_size);
        std::vector<int8_t> w8((size_t)dst_size * steps * 16, 0);
        for (int i = 0; i < dst_size; ++i) {
            aofs[i] = ofs[i] & ~7;   // rounds towards -inf for negative offsets too (two's complement)
            const int shift = ofs[i] - aofs[i];
            for (int t_ = 0; t_ < k; ++t_) {
                const int wq = wk[(size_t)i * k + t_], j = shift + t_;
                const int wl = ((wq + 128) & 255) - 128, wh = (wq - wl) / 256;   // wl in [-128, 127]; |wh| <= 127 for any i16 weight below 32640
                w8[((size_t)i * steps + j / 8) * 16 + (j & 7)] = (int8_t)wl;
                w8[((size_t)i * steps + j / 8) * 16 + 8 + (j & 7)] = (int8_t)wh;
            }
        }
        int span8 = 0;
        for (int x0 = 0; x0 < dst_size; x0 += kSepTX) span8 = std::max(span8, aofs[std::min(x0 + kSepTX, dst_size) - 1] + kp8 - aofs[x0]);
        // sep_v_u8_lds_kernel (this table as the VERTICAL one): the tallest segment of destination rows whose source-row window fits kSepVRows
        int vty = 0, vrows = 0;
        for (int ty_ = 16; ty_ >= 1 && !vty; ty_ >>= 1) {
            int rows = 0;
            for (int y0 = 0; y0 < dst_size; y0 += ty_) rows = std::max(rows, ofs[std::min(y0 + ty_, dst_size) - 1] + k - ofs[y0]);
            if (rows <= kSepVRows) { vty = ty_; vrows = rows; }
        }
        t.meta[8] = vty; t.meta[9] = vrows;
        int wmax = 0;
        for (int16_t v : wk) wmax = std::max(wmax, abs((int)v));
        t.meta[0] = k; t.meta[1] = dst_size; t.meta[2] = kp; t.meta[3] = 0; t.meta[4] = kp8; t.meta[5] = wmax < 32640 ? span8 : 0;   // span8 0: the split does not fit signed bytes
        ofs.resize((ofs.size() + 3) & ~(size_t)3, 0);  // the weight block starts 16-byte aligned
        aofs.resize((aofs.size() + 3) & ~(size_t)3, 0);
        w.resize((w.size() + 7) & ~(size_t)7, 0);      // ... and so do the two blocks after it
        const size_t ofs_bytes = sizeof(int32_t) * ofs.size(), w_bytes = sizeof(int16_t) * w.size(), aofs_bytes = sizeof(int32_t) * aofs.size();
        t.meta[6] = (int)(ofs_bytes + w_bytes); t.meta[7] = (int)(ofs_bytes + w_bytes + aofs_bytes);
        t.bytes = ofs_bytes + w_bytes + aofs_bytes + w8.size();
        KH_HIP(hipMalloc(&t.dev, t.bytes));
        hipError_t err = hipMemcpy(t.dev, ofs.data(), ofs_bytes, hipMemcpyHostToDevice);
        if (err == hipSuccess) err = hipMemcpy((char*)t.dev + ofs_bytes, w.data(), w_bytes, hipMemcpyHostToDevice);
        if (err == hipSuccess) err = hipMemcpy((char*)t.dev + t.meta[6], aofs.data(), aofs_bytes, hipMemcpyHostToDevice);
        if (err == hipSuccess) err = hipMemcpy((char*)t.dev + t.meta[7], w8.data(), w8.size(), hipMemcpyHostToDevice);
        if (err != hipSuccess) return fail_hip(err, "hipMemcpy (resize contribution table)");  // ~DevTable frees the allocation
        return KH_OK;
    }, lease);
    if (rc != KH_OK) return rc;
    out.ofs = (const int32_t*)lease->dev;
    out.w = (const int16_t*)((const char*)lease->dev + sizeof(int32_t) * (((size_t)lease->meta[1] + 3) & ~(size_t)3));
    out.k = lease->meta[0]; out.kp = lease->meta[2];
    out.kp8 = lease->meta[4]; out.span8 = lease->meta[5]; out.vty = lease->meta[8]; out.vrows = lease->meta[9];
    out.aofs = (const int32_t*)((const char*)lease->dev + lease->meta[6]);
    out.w8 = (const uint32_t*)((const char*)lease->dev + lease->meta[7]);
    return KH_OK;
}

int32_t check_rz(const char* what, const void* src, const void* dst, int sw, int sh, int dw, int dh, int channels,
                 int batch, int64_t ss, int64_t ds, int elem) {
    KH_REQUIRE(sw > 0 && sh > 0 && dw > 0 && dh > 0, KH_ERR_INVALID_ARG, "%s: zero-sized image (src %dx%d, dst %dx%d)",
               what, sw, sh, dw, dh);
    KH_REQUIRE(channels >= 1 && channels <= 4, KH_ERR_UNSUPPORTED, "%s: no device kernel for %d channels (supported: 1..4)",
               what, channels);
    KH_REQUIRE(batch >= 0 && batch <= 65535, KH_ERR_TOO_LARGE, "%s: batch %d outside [0, 65535]", what, batch);
    KH_REQUIRE((int64_t)sw * sh * channels * elem <= kI32Max && (int64_t)dw * dh * channels * elem <= kI32Max &&
                   (int64_t)dw * sh * channels * 2 <= kI32Max,
               KH_ERR_TOO_LARGE, "%s: image exceeds 32-bit indexing", what);
    KH_REQUIRE(ss >= 0 && ds >= 0, KH_ERR_INVALID_ARG, "%s: negative batch stride", what);
    if (batch > 0) KH_REQUIRE(src && dst, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    return KH_OK;
}

Rz make_rz(const void* src, void* dst, int sw, int sh, int dw, int dh, int64_t ss, int64_t ds, int batch, int gw, int gh) {
    Rz a;
    a.src = (const uint8_t*)src; a.dst = (uint8_t*)dst;
    a.sw = sw; a.sh = sh; a.dw = dw; a.dh = dh; a.ss = ss; a.ds = ds;
    a.scale_x = (double)sw / (double)dw;
    a.scale_y = (double)sh / (double)dh;
    a.tiles = xcd_tiles(cdiv(gw, kBx), cdiv(gh, kBy), (unsigned)batch, cdiv(gw, kBx) * 8);
    a.plain = 0;
    return a;
}

// nearest_index / cv_nearest_index on the host (the same IEEE f64 operations)
static int nearest_col_host(bool cv, int i, double scale, int src_len) {
    const double v = cv ? std::floor((double)i * scale) : std::floor(((double)i + 0.5) * scale);
    return (int)std::fmin(cv ? v : std::fmax(v, 0.0), (double)(src_len - 1));
}
// RGB: do the sixteen destination bytes of every lane come from at most sixteen source bytes?  (dw * 3 / 16 evaluations, memoised per thread)
static bool nearest_up_rgb_spans_fit(bool cv, int sw, int dw, double scale) {
    static thread_local int last_sw = 0, last_dw = 0; static thread_local bool last_cv = false, last_ok = false;
    if (sw == last_sw && dw == last_dw && cv == last_cv) return last_ok;
    bool ok = true;
    const int rowb = dw * 3;
    for (int B0 = 0; ok && B0 < rowb; B0 += 16) {
        const int Bl = std::min(B0 + 15, rowb - 1);
        const int lo = nearest_col_host(cv, B0 / 3, scale, sw) * 3, hi = nearest_col_host(cv, Bl / 3, scale, sw) * 3 + 2;   // whole source pixels: repeated pixels reach back / ahead inside them
        const int pc = std::min(lo, sw * 3 - 16);
        ok = hi - pc <= 15 && lo - pc >= 0;
    }
    last_sw = sw; last_dw = dw; last_cv = cv; last_ok = ok;
    return ok;
}
// nearest upscale of one / three channels on nearest_up_gray_kernel (false = not taken)
template <bool CV>
bool launch_nearest_up_gray(hipStream_t st, Rz a, int sw, int sh, int dw, int dh, int channels, int batch, const void* dst, int64_t ds) {
    const int C = channels;
    if (!((C == 1 || C == 3) && sw * C >= 16 && dw >= sw && (int64_t)dw * C * dh <= kI32Max && (int64_t)sw * C * sh <= kI32Max)) return false;
    if (C == 3 && !nearest_up_rgb_spans_fit(CV, sw, dw, a.scale_x)) return false;
    const bool dword_ok = (dw * C) % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (batch <= 1 || ds % 4 == 0);   // buffer stores need dword-aligned rows
    a.plain = dword_ok ? plain_row_stores((int64_t)dw * C, dst, ds, batch) : 2;
    a.tiles = xcd_tiles(cdiv(dw * C, 256 * 16), cdiv(dh, kNuRows), (unsigned)batch, cdiv(dw * C, 256 * 16) * 8);
    if (a.tiles.total == 0) return false;
    if (C == 1) hipLaunchKernelGGL((nearest_up_gray_kernel<1, CV>), xcd_grid(a.tiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((nearest_up_gray_kernel<3, CV>), xcd_grid(a.tiles), dim3(256), 0, st, a);
    return true;
}

// horizontal pass: the LDS-staged kernel when its tile fits 64 KiB, else (or with test option resize_u8_gather = 1) the gather
int32_t launch_sep_h(hipStream_t st, const void* src, int sw, int sh, int dw, int dh, int channels, int batch, int64_t ss,
                     int16_t* hbuf, const SepTab& tx, const char* what) {
    const int opt = dev_opt(kOptResizeU8Gather);   // test option: 1 = the per-tap gather kernels for both passes (the fallback for windows beyond the LDS budgets)
    // plane pitch classes (a template constant of the kernel): the smallest that holds the widest tile span of this geometry
    const int pitchp = tx.span8 + 8 <= 320 ? 320 : (tx.span8 + 8 <= 640 ? 640 : (tx.span8 + 8 <= 1280 ? 1280 : 0));
    const size_t lds8 = (size_t)pitchp * channels * kSepRows + (size_t)(tx.kp8 / 8) * kSepTX * 16;   // staged rows + the tile's weights
    if (opt < 1 && tx.span8 > 0 && pitchp > 0 && lds8 <= 64 * 1024 && channels != 2 && sw >= 4) {
        Rz ah = make_rz(src, nullptr, sw, sh, dw, dh, ss, 0, batch, dw, sh);
        ah.tiles = xcd_tiles(cdiv(dw, kSepTX), cdiv(sh, kSepRows), (unsigned)batch, cdiv(dw, kSepTX) * 8);
        if (ah.tiles.total == 0) return fail(KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
        const dim3 grid = xcd_grid(ah.tiles), blk(64 * (kSepRows / kSepRpt));
#define KH_SEPH(CC, PP)                                                                                                                     \
    do {                                                                                                                                    \
        if (lds8 > 48 * 1024) KH_HIP(hipFuncSetAttribute((const void*)sep_h_u8_dot4_kernel<CC, PP, kSepRpt>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
        hipLaunchKernelGGL((sep_h_u8_dot4_kernel<CC, PP, kSepRpt>), grid, blk, lds8, st, ah, hbuf, tx);                                              \
    } while (0)
#define KH_SEPH_C(CC) do { if (pitchp == 320) KH_SEPH(CC, 320); else if (pitchp == 640) KH_SEPH(CC, 640); else KH_SEPH(CC, 1280); } while (0)
        if (channels == 1) KH_SEPH_C(1); else if (channels == 3) KH_SEPH_C(3); else KH_SEPH_C(4);
#undef KH_SEPH_C
#undef KH_SEPH
        return KH_OK;
    }
    Rz ah = make_rz(src, nullptr, sw, sh, dw, dh, ss, 0, batch, dw, sh);
    if (ah.tiles.total == 0) return fail(KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
    const dim3 blk(kBx, kBy);
    switch (channels) {
        case 1: hipLaunchKernelGGL(sep_h_u8_kernel<1>, xcd_grid(ah.tiles), blk, 0, st, ah, hbuf, tx); break;
        case 3: hipLaunchKernelGGL(sep_h_u8_kernel<3>, xcd_grid(ah.tiles), blk, 0, st, ah, hbuf, tx); break;
        default: hipLaunchKernelGGL(sep_h_u8_kernel<4>, xcd_grid(ah.tiles), blk, 0, st, ah, hbuf, tx); break;
    }
    return KH_OK;
}

#define KH_RZ_LAUNCH_C(KERNEL, channels, a, st)                                                           \
    do {                                                                                                  \
        const dim3 blk(kBx, kBy), grid = xcd_grid((a).tiles);                                             \
        switch (channels) {                                                                               \
            case 1: hipLaunchKernelGGL((KERNEL<1>), grid, blk, 0, st, a); break;                          \
            case 2: hipLaunchKernelGGL((KERNEL<2>), grid, blk, 0, st, a); break;                          \
            case 3: hipLaunchKernelGGL((KERNEL<3>), grid, blk, 0, st, a); break;                          \
            default: hipLaunchKernelGGL((KERNEL<4>), grid, blk, 0, st, a); break;                         \
        }                                                                                                 \
    } while (0)

}  // namespace

extern "C" {

int32_t kh_resize_fast_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t sw, int32_t sh, int32_t dw,
                          int32_t dh, int32_t channels, int32_t mode, int32_t antialias, int32_t batch,
                          int64_t src_stride, int64_t dst_stride) {
    const char* what = "kh_resize_fast_u8";
    if (int32_t rc = check_rz(what, src, dst, sw, sh, dw, dh, channels, batch, src_stride, dst_stride, 1)) return rc;
    KH_REQUIRE(mode >= KH_INTERP_NEAREST && mode <= KH_INTERP_LANCZOS, KH_ERR_UNSUPPORTED,
               "%s: unknown interpolation mode %d", what, mode);
    // resize_u8_path (P/resize/mod.rs:283-340): errors are decided before anything is launched
    const bool down2 = mode == KH_INTERP_BILINEAR && channels == 3 && sw == dw * 2 && sh == dh * 2 && sw >= 2 && sh >= 2;
    const bool up2 = mode == KH_INTERP_BILINEAR && channels == 3 && dw == sw * 2 && dh == sh * 2 && sw >= 2 && sh >= 2;
    if (!down2 && !up2 && mode != KH_INTERP_NEAREST) {
        KH_REQUIRE(channels == 1 || channels == 3 || channels == 4, KH_ERR_UNSUPPORTED,
                   "%s: unsupported channel count %d (1, 3 or 4)", what, channels);
        if (mode == KH_INTERP_BILINEAR)
            KH_REQUIRE(sw >= 2 && sh >= 2, KH_ERR_INVALID_ARG, "%s: bilinear needs a source of at least 2x2 (got %dx%d)",
                       what, sw, sh);
    }
    if (batch == 0) return KH_OK;
    // Exact 2x bilinear upscale (round 6): the rolling pyrup kernels with this resize's arithmetic — the reference's rounding-halving chains
    // for RGB, the generic Q14 weights (9 3 3 1) / 16 for one / four channels — instead of the per-pixel forms below, which ran at 0.09-0.24
    // of peak (1080p -> 4K RGB 0.73 -> 0.1 ms per 16 images, profiles/r06zz3_resize_up2.txt).  Test option resize_u8_px = 1 / 2 keeps them.
    // (nearest at this scale: column X >> 1, the same walk with no arithmetic — label maps are upscaled this way: 0.085 -> 0.03 ms on one channel)
    if ((mode == KH_INTERP_BILINEAR || mode == KH_INTERP_NEAREST) && dw == sw * 2 && dh == sh * 2 && sw >= 2 && sh >= 2 &&
        (channels == 1 || channels == 3 || (channels == 4 && mode == KH_INTERP_BILINEAR)) &&   // (RGBA nearest: the quad kernel is already at 0.87 of peak, 0.096 vs 0.109 ms)
        dev_opt(kOptResizeU8Px) != 1 && dev_opt(kOptResizeU8Px) != 2) {
        int32_t rc = KH_OK;
        if (resize_up2_u8_rolling(stream, src, dst, sw, sh, channels, batch, src_stride, dst_stride, what, rc, mode == KH_INTERP_NEAREST)) return rc;
    }
    hipStream_t st = as_hip(stream);
    Rz a = make_rz(src, dst, sw, sh, dw, dh, src_stride, dst_stride, batch, dw, dh);
    KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
    const dim3 blk(kBx, kBy);
    // the simple paths: four pixels per lane with dword stores where the destination rows are whole quads on 4-byte-aligned images
    // (test option resize_u8_px = 1: one pixel per thread, byte stores)
    // Q14 bilinear only up to a 2x horizontal downscale: beyond it a lane's four tap pairs — and the 64 lanes of a load — spread over so
    // many lines that the one-pixel mapping (adjacent lanes = adjacent taps) wins again (1080p -> 640 x 360: 0.257 vs 0.321 ms per 256
    // images; -> 1280 x 720: 0.896 vs 0.714; profiles/r06zg_resize_u8_quads.txt).  resize_u8_px = 4: quads wherever they are possible.
    const int px_opt = dev_opt(kOptResizeU8Px);
    const bool quads = dw % 4 == 0 && px_opt != 1 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (batch <= 1 || dst_stride % 4 == 0) &&
                       (int64_t)dw * channels <= kI32Max && (px_opt == 4 || down2 || up2 || mode == KH_INTERP_NEAREST || sw <= 2 * dw);
    if (quads) a.tiles = xcd_tiles(cdiv(dw, kBx * 4), cdiv(dh, kBy), (unsigned)batch, cdiv(dw, kBx * 4) * 8);
    if (quads) a.plain = plain_row_stores((int64_t)dw * channels, dst, dst_stride, batch);
#define KH_RZ_OP(CC, OP) do { if (quads) hipLaunchKernelGGL((resize_u8_quads_kernel<CC, OP>), xcd_grid(a.tiles), blk, 0, st, a); \
                              else hipLaunchKernelGGL((resize_u8_px_kernel<CC, OP>), xcd_grid(a.tiles), blk, 0, st, a); } while (0)
#define KH_RZ_OP_C(OP) do { switch (channels) { case 1: KH_RZ_OP(1, OP); break; case 2: KH_RZ_OP(2, OP); break; case 3: KH_RZ_OP(3, OP); break; default: KH_RZ_OP(4, OP); break; } } while (0)
    // exact 2x downscale of 1 / 4 channels on whole-quad rows: the generic Q14 bilinear equals the 2 x 2 box there (see resize_u8_quads_kernel)
    const bool down2_any = mode == KH_INTERP_BILINEAR && (channels == 1 || channels == 4) && sw == dw * 2 && sh == dh * 2 && quads && px_opt != 2 &&
                           (int64_t)sw * sh * channels <= kI32Max;
    if (down2) {
        KH_RZ_OP(3, kRzDown2);
    } else if (down2_any) {
        if (channels == 1) hipLaunchKernelGGL((resize_u8_quads_kernel<1, kRzDown2>), xcd_grid(a.tiles), blk, 0, st, a);
        else hipLaunchKernelGGL((resize_u8_quads_kernel<4, kRzDown2>), xcd_grid(a.tiles), blk, 0, st, a);
    } else if (up2) {
        KH_RZ_OP(3, kRzUp2);
    } else if (mode == KH_INTERP_NEAREST) {
        // one channel, an upscale: sixteen destination columns per lane with the column selectors computed once (test option resize_u8_px = 1 / 2: the quad / per-pixel kernels)
        if ((channels == 1 || channels == 3) && px_opt != 1 && px_opt != 2 && launch_nearest_up_gray<false>(st, a, sw, sh, dw, dh, channels, batch, dst, dst_stride)) return check_launch(what);
        KH_RZ_OP_C(kRzNearest);
    } else if (mode == KH_INTERP_BILINEAR) {
        KH_RZ_OP_C(kRzBilinear);
    } else {
        const int filt = mode == KH_INTERP_BICUBIC ? 0 : 1;
        SepTab tx, ty;
        TableLease lx, ly;  // keep both tables alive until the launches that read them are enqueued (and recorded below)
        if (int32_t rc = get_tab(sw, dw, filt, antialias != 0, st, what, tx, lx)) return rc;
        if (int32_t rc = get_tab(sh, dh, filt, antialias != 0, st, what, ty, ly)) return rc;
        const int hrow = dw * channels;
        Scratch scratch;  // dst_w x src_h i16 intermediate (P/resize/cuda.rs:262): caller workspace or stream-ordered pool
        if (int32_t rc = get_scratch(stream, sizeof(int16_t) * (size_t)hrow * sh * batch, what, scratch)) return rc;
        int16_t* hbuf = scratch.as<int16_t>();
        Rz av = make_rz(src, dst, sw, sh, dw, dh, src_stride, dst_stride, batch, hrow, dh);
        if (av.tiles.total == 0) return fail(KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
        if (int32_t rc = launch_sep_h(st, src, sw, sh, dw, dh, channels, batch, src_stride, hbuf, tx, what)) return rc;
        if (ty.vty > 0 && dev_opt(kOptResizeU8Gather) < 1) {   // LDS-staged vertical pass; the per-tap kernel for windows beyond kSepVRows rows
            Rz al = av;
            al.tiles = xcd_tiles(cdiv(hrow, 128), cdiv(dh, ty.vty), (unsigned)batch, cdiv(hrow, 128) * 4);
            if (al.tiles.total == 0) return fail(KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
            const size_t vlds = (size_t)(((ty.vrows + 1) >> 1) + 2) * 512 + (size_t)ty.vty * ((ty.kp >> 1) + 1) * 4;   // row pairs + two spare, weight rows
            if (vlds > 48 * 1024) KH_HIP(hipFuncSetAttribute((const void*)sep_v_u8_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            hipLaunchKernelGGL(sep_v_u8_lds_kernel, xcd_grid(al.tiles), dim3(256), vlds, st, al, (const int16_t*)hbuf, ty, hrow);
        } else {
            hipLaunchKernelGGL(sep_v_u8_kernel, xcd_grid(av.tiles), blk, 0, st, av, (const int16_t*)hbuf, ty, hrow);
        }
        const int32_t rc = check_launch(what);
        lx->used_on(st); ly->used_on(st);
        return rc;
    }
    return check_launch(what);
}

int32_t kh_resize_normalize_to_chw_u8_f32(kh_stream_t stream, const uint8_t* src, float* dst, int32_t sw, int32_t sh,
                                          int32_t dw, int32_t dh, const float* scale, const float* bias, int32_t mode,
                                          int32_t antialias, int32_t batch, int64_t src_stride, int64_t dst_stride) {
    const char* what = "kh_resize_normalize_to_chw_u8_f32";
    if (int32_t rc = check_rz(what, src, dst, sw, sh, dw, dh, 3, batch, src_stride, dst_stride, 1)) return rc;
    KH_REQUIRE((int64_t)dw * dh * 3 <= kI32Max / 4, KH_ERR_TOO_LARGE, "%s: output exceeds 32-bit indexing", what);
    KH_REQUIRE(scale && bias, KH_ERR_INVALID_ARG, "%s: null scale / bias", what);
    KH_REQUIRE(mode >= KH_INTERP_NEAREST && mode <= KH_INTERP_LANCZOS, KH_ERR_UNSUPPORTED,
               "%s: unknown interpolation mode %d", what, mode);
    if (batch == 0) return KH_OK;
    hipStream_t st = as_hip(stream);
    Norm3 n;
    for (int c = 0; c < 3; ++c) { n.scale[c] = scale[c]; n.bias[c] = bias[c]; }
    Rz a = make_rz(src, dst, sw, sh, dw, dh, src_stride, dst_stride, batch, dw, dh);
    KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
    const dim3 blk(kBx, kBy), grid = xcd_grid(a.tiles);
    const float sxf = (float)sw / (float)dw, syf = (float)sh / (float)dh;
    if (mode == KH_INTERP_NEAREST) {
        hipLaunchKernelGGL(fused_rgb_chw_kernel<0>, grid, blk, 0, st, a, n, sxf, syf);
    } else if (mode == KH_INTERP_BILINEAR) {
        if (sw == 2 * dw && sh == 2 * dh) hipLaunchKernelGGL(fused_rgb_chw_kernel<2>, grid, blk, 0, st, a, n, sxf, syf);
        else hipLaunchKernelGGL(fused_rgb_chw_kernel<1>, grid, blk, 0, st, a, n, sxf, syf);
    } else {
        const int filt = mode == KH_INTERP_BICUBIC ? 0 : 1;
        SepTab tx, ty;
        TableLease lx, ly;  // keep both tables alive until the launches that read them are enqueued (and recorded below)
        if (int32_t rc = get_tab(sw, dw, filt, antialias != 0, st, what, tx, lx)) return rc;
        if (int32_t rc = get_tab(sh, dh, filt, antialias != 0, st, what, ty, ly)) return rc;
        Scratch scratch;
        if (int32_t rc = get_scratch(stream, sizeof(int16_t) * (size_t)dw * 3 * sh * batch, what, scratch)) return rc;
        int16_t* hbuf = scratch.as<int16_t>();
        if (int32_t rc = launch_sep_h(st, src, sw, sh, dw, dh, 3, batch, src_stride, hbuf, tx, what)) return rc;
        hipLaunchKernelGGL(fused_sep_v_kernel, grid, blk, 0, st, a, (const int16_t*)hbuf, ty, n);
        const int32_t rc = check_launch(what);
        lx->used_on(st); ly->used_on(st);
        return rc;
    }
    return check_launch(what);
}

static int32_t resize_opencv(const char* what, kh_stream_t stream, const void* src, void* dst, int sw, int sh, int dw,
                             int dh, int channels, int mode, int batch, int64_t ss, int64_t ds, int elem) {
    if (int32_t rc = check_rz(what, src, dst, sw, sh, dw, dh, channels, batch, ss, ds, elem)) return rc;
    // opencv_compat.rs:95-101: nearest and linear only
    KH_REQUIRE(mode == KH_INTERP_NEAREST || mode == KH_INTERP_BILINEAR, KH_ERR_UNSUPPORTED,
               "%s: unsupported interpolation mode %d (nearest, bilinear)", what, mode);
    if (batch == 0) return KH_OK;
    hipStream_t st = as_hip(stream);
    Rz a = make_rz(src, dst, sw, sh, dw, dh, ss * elem, ds * elem, batch, dw, dh);
    // linear_axis / nearest_axis use scale = 1 / (dst / src), not src / dst
    a.scale_x = 1.0 / ((double)dw / (double)sw);
    a.scale_y = 1.0 / ((double)dh / (double)sh);
    KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
    // u8: four pixels per lane with dword stores under the conditions of kh_resize_fast_u8's simple paths (linear up to a 2x downscale)
    const int px_opt = dev_opt(kOptResizeU8Px);
    // INTER_NEAREST at an exact 2x upscale: floor(i * 0.5) = i >> 1 on both axes — the rolling walk of kh_resize_fast_u8's nearest (gray / RGB)
    // INTER_LINEAR at the same scale: the rolling walk with the reference's fixed-point arithmetic (kh_pyramid_morph.hip::kUpCv), 1 / 3 / 4 channels
    const bool cv_nn = mode == KH_INTERP_NEAREST && (channels == 1 || channels == 3), cv_lin = mode == KH_INTERP_BILINEAR && (channels == 1 || channels == 3 || channels == 4);
    if (elem == 1 && (cv_nn || cv_lin) && dw == 2 * sw && dh == 2 * sh && sw >= 2 && sh >= 2 && px_opt != 1 && px_opt != 2) {
        int32_t rc = KH_OK;
        if (resize_up2_u8_rolling(stream, static_cast<const uint8_t*>(src), static_cast<uint8_t*>(dst), sw, sh, channels, batch, ss, ds, what, rc, cv_nn, cv_lin)) return rc;
    }
    // INTER_NEAREST upscales of one channel: the column selectors once per lane (nearest_up_gray_kernel)
    if (elem == 1 && mode == KH_INTERP_NEAREST && (channels == 1 || channels == 3) && px_opt != 1 && px_opt != 2 && launch_nearest_up_gray<true>(st, a, sw, sh, dw, dh, channels, batch, dst, ds)) return check_launch(what);
    if (elem == 1 && dw % 4 == 0 && px_opt != 1 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (batch <= 1 || ds % 4 == 0) && (int64_t)dw * channels <= kI32Max &&
        (px_opt == 4 || mode == KH_INTERP_NEAREST || sw <= 2 * dw)) {
        a.tiles = xcd_tiles(cdiv(dw, kBx * 4), cdiv(dh, kBy), (unsigned)batch, cdiv(dw, kBx * 4) * 8);
        a.plain = plain_row_stores((int64_t)dw * channels, dst, ds, batch);
        const dim3 qblk(kBx, kBy), qgrid = xcd_grid(a.tiles);
        // INTER_LINEAR at an exact 2x downscale (round 6): every coefficient is 1024 of 2048, no border column / row is reached, and
        // ((1024 ((1024 (p00 + p01)) >> 4)) >> 16) + (the same of the lower row) + 2) >> 2 == (p00 + p01 + p10 + p11 + 2) >> 2 exactly (the shifts
        // drop only zero bits): the 2 x 2 box of resize_u8_quads_kernel on packed bytes, 0.076 -> 0.025 ms per 16 4K gray planes.
        if (mode == KH_INTERP_BILINEAR && sw == 2 * dw && sh == 2 * dh && (channels == 1 || channels == 3 || channels == 4) && px_opt != 2 &&
            (int64_t)sw * sh * channels <= kI32Max) {
            if (channels == 1) hipLaunchKernelGGL((resize_u8_quads_kernel<1, kRzDown2>), qgrid, qblk, 0, st, a);
            else if (channels == 3) hipLaunchKernelGGL((resize_u8_quads_kernel<3, kRzDown2>), qgrid, qblk, 0, st, a);
            else hipLaunchKernelGGL((resize_u8_quads_kernel<4, kRzDown2>), qgrid, qblk, 0, st, a);
            return check_launch(what);
        }
#define KH_CVQ(CC) do { if (mode == KH_INTERP_NEAREST) hipLaunchKernelGGL((cv_u8_quads_kernel<CC, false>), qgrid, qblk, 0, st, a); \
                        else hipLaunchKernelGGL((cv_u8_quads_kernel<CC, true>), qgrid, qblk, 0, st, a); } while (0)
        switch (channels) { case 1: KH_CVQ(1); break; case 2: KH_CVQ(2); break; case 3: KH_CVQ(3); break; default: KH_CVQ(4); break; }
#undef KH_CVQ
        return check_launch(what);
    }
    const dim3 blk(kBx, kBy), grid = xcd_grid(a.tiles);
    if (mode == KH_INTERP_NEAREST) {
        if (elem == 1) {
            switch (channels) {
                case 1: hipLaunchKernelGGL((cv_nearest_kernel<uint8_t, 1>), grid, blk, 0, st, a); break;
                case 2: hipLaunchKernelGGL((cv_nearest_kernel<uint8_t, 2>), grid, blk, 0, st, a); break;
                case 3: hipLaunchKernelGGL((cv_nearest_kernel<uint8_t, 3>), grid, blk, 0, st, a); break;
                default: hipLaunchKernelGGL((cv_nearest_kernel<uint8_t, 4>), grid, blk, 0, st, a); break;
            }
        } else {
            switch (channels) {
                case 1: hipLaunchKernelGGL((cv_nearest_kernel<float, 1>), grid, blk, 0, st, a); break;
                case 2: hipLaunchKernelGGL((cv_nearest_kernel<float, 2>), grid, blk, 0, st, a); break;
                case 3: hipLaunchKernelGGL((cv_nearest_kernel<float, 3>), grid, blk, 0, st, a); break;
                default: hipLaunchKernelGGL((cv_nearest_kernel<float, 4>), grid, blk, 0, st, a); break;
            }
        }
    } else if (elem == 1) {
        KH_RZ_LAUNCH_C(cv_linear_u8_kernel, channels, a, st);
    } else {
        KH_RZ_LAUNCH_C(cv_linear_f32_kernel, channels, a, st);
    }
    return check_launch(what);
}

int32_t kh_resize_opencv_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t sw, int32_t sh, int32_t dw,
                            int32_t dh, int32_t channels, int32_t mode, int32_t batch, int64_t src_stride,
                            int64_t dst_stride) {
    return resize_opencv("kh_resize_opencv_u8", stream, src, dst, sw, sh, dw, dh, channels, mode, batch, src_stride,
                         dst_stride, 1);
}

int32_t kh_resize_opencv_f32(kh_stream_t stream, const float* src, float* dst, int32_t sw, int32_t sh, int32_t dw,
                             int32_t dh, int32_t channels, int32_t mode, int32_t batch, int64_t src_stride,
                             int64_t dst_stride) {
    return resize_opencv("kh_resize_opencv_f32", stream, src, dst, sw, sh, dw, dh, channels, mode, batch, src_stride,
                         dst_stride, 4);
}

}  // extern "C"
