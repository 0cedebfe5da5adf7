// Gaussian pyramid (pyrdown / pyrup, f32 and u8) and u8 morphology (dilate / erode) for gfx950.
//
// Device twins of P/cuda/pyramid.rs:84-362 and P/cuda/morphology.rs (adapters P/pyramid.rs
// `cuda_adapters`, P/morphology/cuda.rs:52-200); arithmetic of the CPU ops pyrdown_f32 / pyrup_f32 /
// pyrdown_u8 / pyrup_u8 (P/pyramid.rs:210,312,469,804) and dilate / erode (P/morphology/ops.rs:22,125)
// over PaddingMode::map_index (P/padding.rs:32-80).  Results are bit-identical
// (tests/test_pyramid_morph_gpu.py).
//
// The reference runs the pyramids as an H pass into an intermediate image and a V pass; here each
// destination pixel evaluates both passes itself — the intermediate values it needs are recomputed
// from at most 5x5 (down) / 3x3 (up) source taps that sit in L1/L2, with exactly the reference's
// per-pass expressions and roundings (u16 / u8 / f32 intermediates), so no scratch image is written.
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "kh_common.h"

using namespace kh;

namespace {

constexpr int kBx = 64, kBy = 4;

template <typename T>
struct Pyr {
    const T* src;
    T* dst;
    int sw, sh, dw, dh;
    long long ss, ds;  // elements between consecutive images
    XcdTiles tiles;
};

#define KH_PYR_PROLOGUE(T)                                      \
    unsigned bx_, by_, bz_;                                     \
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;              \
    const int X = bx_ * kBx + threadIdx.x;                      \
    const int Y = by_ * kBy + threadIdx.y;                      \
    if (X >= a.dw || Y >= a.dh) return;                         \
    const T* __restrict__ src = a.src + (long long)bz_ * a.ss;  \
    T* __restrict__ o = a.dst + (long long)bz_ * a.ds + ((long long)Y * a.dw + X) * C;

__device__ __forceinline__ int reflect_101(int p, int len) {  // pyramid.rs:252-270
    if ((unsigned)p < (unsigned)len) return p;  // the common case costs one compare, not a modulo
    if (len == 1) return 0;
    if (p < 0) p = -p;
    const int period = 2 * (len - 1);
    p %= period;
    return p >= len ? period - p : p;
}

// pyrdown_f32 (:312-430): 5x5 outer-product taps, ky-major accumulation, reflect-101 border.
// (Round 3 tried sharing source pixels along the wave — a lane loads only its own pair and takes the other three pixels from the
// lanes either side by DPP shifts, 10 full-wave loads per pixel instead of 25: 2.53 ms against this kernel's 2.22 on one box,
// profiles/r03zc — the row-by-row load / shift dependency costs more than the loads it saves.  The kernel is not addresser-bound.)
template <int C>
__global__ __launch_bounds__(kBx* kBy) void pyrdown_f32_kernel(Pyr<float> a) {
    KH_PYR_PROLOGUE(float)
    const float k1[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    int sx[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) sx[t] = reflect_101(2 * X + t - 2, a.sw) * C;
    float sum[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sum[c] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        const float* row = src + (long long)reflect_101(2 * Y + ky - 2, a.sh) * a.sw * C;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const float w = k1[ky] * k1[kx];  // exact products of powers of two and 3
#pragma unroll
            for (int c = 0; c < C; ++c) sum[c] += row[sx[kx] + c] * w;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = sum[c];
}

// ---- pyrdown_f32, streaming wave (round 4) ------------------------------------------------------------------------------------------
// Limiter of the per-pixel kernel above (profiles/r04a_limiter_pyrdown_f32.txt): 25 strided 12-byte loads per destination pixel put
// 25x the source through the vector L1 — 51.8 M wave-loads per 64 4K images at ~24 address / data cycles each keep every CU's
// texture addresser busy for the whole 2.0 ms (TA_BUSY = kernel time, 60 % of the wave-cycles are VMEM issue stalls) while HBM moves
// 8.3 GB at 4.1 TB/s.  The round-3 attempts cut the load COUNT but paid for it in registers (146 VGPRs, 3 waves per SIMD) or in
// dependent DPP chains.  Here a wave owns 64 adjacent flat destination floats and walks DOWN a strip, streaming the source rows once:
//   * per source row the wave loads only its span — the pixels 2 px0 - 2 .. 2 px1 + 2 its 64 outputs tap (at most 160 floats: up to three
//     coalesced dword loads per lane, reflect-101 applied to the loaded element's pixel index, so borders need no other code) — into a
//     wave-private LDS row; no block barrier (DS operations of one wave execute in order);
//   * a lane reads its 5 horizontal taps of that row from LDS and adds their products straight into the (up to) three destination rows
//     the source row belongs to: rows arrive in increasing ky for each of them, so every output still sees the reference's 25 products
//     in the reference's order (ky-major, `sum += v * (k1[ky] * k1[kx])`, starting from +0) — bit-identical — with THREE running sums
//     per lane instead of a 5 x 5 register window: 10 LDS reads per output instead of 25 vector-memory loads, ~40 VGPRs.
// A 256-thread block = 4 independent waves = 256 flat destination floats; kPdfQ source rows of loads are in flight per lane.
constexpr int kPdfQ = 6, kPdfSpan = 160, kPdfStripMax = 360;
struct PyrF32Roll { const float* src; float* dst; int sw, sh, dw, dh, th, C; long long ss, ds; XcdTiles tiles; int plain; };   // plain: kh_common.h::plain_row_stores

template <int C>
__global__ __launch_bounds__(256) void pyrdown_f32_roll_kernel(PyrF32Roll a) {
    __shared__ float rowbuf[4][kPdfSpan + 32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int rowlen_d = a.dw * C, rowlen_s = a.sw * C;
    const int gx0 = tx * 256 + wv * 64;   // first flat destination float of this wave
    if (gx0 >= rowlen_d) return;          // whole wave idle (no block barrier below)
    const int Y0 = ty * a.th, nrows = min(a.th, a.dh - Y0);
    const float* __restrict__ src = a.src + (long long)bz * a.ss;
    float* buf = rowbuf[wv];

    const int gx = gx0 + lane;
    const bool gx_ok = gx < rowlen_d;
    const int px0 = gx0 / C;                                  // first destination pixel this wave touches
    const int pxl = min(gx, rowlen_d - 1) / C, ch = min(gx, rowlen_d - 1) - pxl * C;
    const int tap0 = (2 * (pxl - px0)) * C + ch;             // LDS index of this lane's kx = 0 tap; tap kx sits kx * C further
    // the three span elements this lane loads per source row: e = lane + 64 j <-> source pixel 2 px0 - 2 + e / C, channel e % C
    int soff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int e = min(lane + 64 * j, kPdfSpan - 1);
        soff[j] = reflect_101(2 * px0 - 2 + e / C, a.sw) * C + e % C;   // always inside the row
    }
    const bool third = lane < kPdfSpan - 128;

    float q[kPdfQ][3];
    int pf = 0;   // source walk step of the next prefetch: source row 2 Y0 - 2 + pf
    auto prefetch = [&](float (&d)[3]) {
        const int base = reflect_101(2 * Y0 - 2 + pf, a.sh) * rowlen_s;   // 32-bit: host-checked
        d[0] = src[base + soff[0]];
        d[1] = src[base + soff[1]];
        d[2] = src[base + soff[2]];   // lanes >= 32 re-load element 159's neighbourhood (an L1 hit): no exec mask, no vmcnt(0) drain
        ++pf;
    };
#pragma unroll
    for (int p = 0; p < kPdfQ; ++p) prefetch(q[p]);

    const float k1[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;   // running sums of destination rows m, m - 1, m - 2 of the walk
    const __amdgpu_buffer_rsrc_t ow = stream_window(a.dst + (long long)bz * a.ds + (long long)Y0 * rowlen_d, (long long)(a.dh - Y0) * rowlen_d * 4);
    int out_off = (gx - 2 * rowlen_d) * 4;   // byte offset of destination row m - 2 inside the window (negative during the warm-up: never stored)

    auto stage = [&](const float (&d)[3]) {
        buf[lane] = d[0];
        buf[lane + 64] = d[1];
        if (third) buf[lane + 128] = d[2];
        __builtin_amdgcn_wave_barrier();   // one wave: DS operations execute in issue order; this only pins the compiler's order
    };
    // one walk step = one destination row = source rows 2 m (even: ky = 0 / 2 / 4 of rows m, m - 1, m - 2) and 2 m + 1 (odd: ky = 1 / 3)
    const int steps = nrows + 2;
    for (int mb = 0; mb < steps; mb += kPdfQ / 2) {
#pragma unroll
        for (int u = 0; u < kPdfQ / 2; ++u) {
            const int m = mb + u;
            float t[5];
            // even source row
            stage(q[2 * u]);
            prefetch(q[2 * u]);
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) t[kx] = buf[tap0 + kx * C];
            __builtin_amdgcn_wave_barrier();   // every lane has read the row before it is overwritten
            s0 = 0.0f;
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                s2 += t[kx] * (k1[4] * k1[kx]);
                s1 += t[kx] * (k1[2] * k1[kx]);
                s0 += t[kx] * (k1[0] * k1[kx]);
            }
            if (gx_ok && m >= 2 && m < steps) { const uint32_t bits = __float_as_uint(s2); row_store<1>(ow, out_off, &bits, a.plain); }
            out_off += rowlen_d * 4;
            // odd source row
            stage(q[2 * u + 1]);
            prefetch(q[2 * u + 1]);
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) t[kx] = buf[tap0 + kx * C];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                s1 += t[kx] * (k1[3] * k1[kx]);
                s0 += t[kx] * (k1[1] * k1[kx]);
            }
            s2 = s1;
            s1 = s0;
        }
    }
}

// pyrup_horizontal_pass_f32 (:22-96) at column X of one source row
template <int C>
__device__ __forceinline__ float pyrup_h(const float* __restrict__ row, int sw, int X, int c) {
    if (sw == 1) return row[c];
    const int x = X >> 1;
    const bool odd = X & 1;
    if (x == 0) {
        const float l = row[c], r = row[C + c];
        return odd ? (l + r) * 0.5f : (6.0f * l + 2.0f * r) * 0.125f;
    }
    if (x == sw - 1) {
        const float prev = row[(x - 1) * C + c], curr = row[x * C + c];
        return odd ? curr : (1.0f * prev + 7.0f * curr) * 0.125f;
    }
    const float prev = row[(x - 1) * C + c], curr = row[x * C + c], next = row[(x + 1) * C + c];
    return odd ? (curr + next) * 0.5f : (1.0f * prev + 6.0f * curr + 1.0f * next) * 0.125f;
}

template <int C>
__global__ __launch_bounds__(kBx* kBy) void pyrup_f32_kernel(Pyr<float> a) {  // + vertical pass (:98-170)
    KH_PYR_PROLOGUE(float)
    const int y = Y >> 1;
    const bool odd = Y & 1;
    int rt, rc, rb;
    if (a.sh == 1) { rt = rc = rb = 0; }
    else if (y == 0) { rt = 0; rc = 0; rb = 1; }
    else if (y == a.sh - 1) { rt = a.sh - 2; rc = a.sh - 1; rb = a.sh - 1; }
    else { rt = y - 1; rc = y; rb = y + 1; }
    const long long stride = (long long)a.sw * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float top = pyrup_h<C>(src + rt * stride, a.sw, X, c);
        const float cen = pyrup_h<C>(src + rc * stride, a.sw, X, c);
        const float bot = pyrup_h<C>(src + rb * stride, a.sw, X, c);
        float v;
        if (y == 0) v = odd ? (cen + bot) * 0.5f : (6.0f * cen + 2.0f * bot) * 0.125f;
        else if (y == a.sh - 1) v = odd ? cen : (1.0f * top + 7.0f * cen) * 0.125f;
        else v = odd ? (cen + bot) * 0.5f : (1.0f * top + 6.0f * cen + 1.0f * bot) * 0.125f;
        o[c] = v;
    }
}

// pyrup_f32 by SOURCE pixel (round 2).  The kernel above works per destination pixel: every thread re-derives three horizontal
// passes (each with its own edge cases) for one output, and odd / even columns and rows diverge inside every wave — 6.9 ms per 64
// 1080p -> 4K RGB images, 0.14 of the roofline (r02zg).  Here a thread owns one source pixel and writes its 2 x 2 destination
// block: nine dwordx3 loads (3 x 3 neighbourhood, clamped addresses), the horizontal pass of the three rows once for the even and
// the odd column, the vertical pass for the even and the odd row, two 2-pixel stores.  The expressions are the per-pixel kernel's,
// operand for operand (including the multiplications by 1.0f), so the floats are identical.
template <int C>
__global__ __launch_bounds__(kBx* kBy) void pyrup_f32_block_kernel(Pyr<float> a) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    // a block is 64 x 4 threads: a wave = 64 consecutive source pixels of one row.  Lanes past the image stay alive (clamped to the
    // last pixel / row) because their neighbours take values from them; they do not store.
    const int lane = threadIdx.x;
    const int xr = bx_ * kBx + lane, yr = by_ * kBy + threadIdx.y;
    const int x = min(xr, a.sw - 1), y = min(yr, a.sh - 1);  // SOURCE pixel
    const float* __restrict__ src = a.src + (long long)bz_ * a.ss;
    float* __restrict__ dst = a.dst + (long long)bz_ * a.ds;
    int rt, rc, rb;
    if (a.sh == 1) { rt = rc = rb = 0; }
    else if (y == 0) { rt = 0; rc = 0; rb = 1; }
    else if (y == a.sh - 1) { rt = a.sh - 2; rc = a.sh - 1; rb = a.sh - 1; }
    else { rt = y - 1; rc = y; rb = y + 1; }
    const int xm = max(x - 1, 0), xp = min(x + 1, a.sw - 1);
    const long long stride = (long long)a.sw * C;
    const int rows[3] = {rt, rc, rb};
    // Each lane loads ITS pixel of the three rows (lane-consecutive: a wave-load is 768 contiguous bytes) and takes the left / right
    // neighbours from the adjacent lanes; only lanes 0 and 63 load a neighbour themselves.  Loading all nine pixels per lane cost
    // ~35 L1 accesses per wave-load x 9 and the L1's access rate was 1.8 of the kernel's 2.7 ms (r02zq).  The values are pinned
    // (empty asm) so that the edge selects below cannot pull a load into a branch.
    float win[3][3][C];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float* row = src + rows[r] * stride;
#pragma unroll
        for (int c = 0; c < C; ++c) win[r][1][c] = row[x * C + c];
    }
    float edge[3][C];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c) edge[r][c] = 0.0f;
    if (lane == 0 || lane == kBx - 1) {
        const int xe = lane == 0 ? xm : xp;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float* row = src + rows[r] * stride;
#pragma unroll
            for (int c = 0; c < C; ++c) edge[r][c] = row[xe * C + c];
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            asm volatile("" : "+v"(win[r][1][c]));
            const int bits = __float_as_uint(win[r][1][c]);
            const float left = __uint_as_float((uint32_t)__shfl(bits, max(lane - 1, 0)));
            const float right = __uint_as_float((uint32_t)__shfl(bits, min(lane + 1, kBx - 1)));
            win[r][0][c] = lane == 0 ? edge[r][c] : left;        // pixel xm: lane - 1 holds x - 1 (x >= 1 here)
            win[r][2][c] = lane == kBx - 1 ? edge[r][c] : right;  // pixel xp: lane + 1 holds min(x + 1, sw - 1)
        }
    float he[3][C], ho[3][C];  // horizontal pass at columns 2x and 2x + 1 of rows rt, rc, rb
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float prev = win[r][0][c], curr = win[r][1][c], next = win[r][2][c];
            const float e_in = (1.0f * prev + 6.0f * curr + 1.0f * next) * 0.125f, e_first = (6.0f * curr + 2.0f * next) * 0.125f;
            const float e_last = (1.0f * prev + 7.0f * curr) * 0.125f, o_in = (curr + next) * 0.5f;
            he[r][c] = x == 0 ? e_first : x == a.sw - 1 ? e_last : e_in;   // sw >= 2 here (the host sends 1-pixel-wide images
            ho[r][c] = x == a.sw - 1 ? curr : o_in;                           // to the per-destination-pixel kernel)
        }
    }
    float out[2][2 * C];  // [row 2y, 2y + 1][pixel 2x channels, pixel 2x + 1 channels]
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // k = 0: column 2x (he), 1: column 2x + 1 (ho)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float top = k ? ho[0][c] : he[0][c], cen = k ? ho[1][c] : he[1][c], bot = k ? ho[2][c] : he[2][c];
            const float v_in = (1.0f * top + 6.0f * cen + 1.0f * bot) * 0.125f, v_first = (6.0f * cen + 2.0f * bot) * 0.125f;
            const float v_last = (1.0f * top + 7.0f * cen) * 0.125f, v_odd = (cen + bot) * 0.5f;
            out[0][k * C + c] = y == 0 ? v_first : y == a.sh - 1 ? v_last : v_in;
            out[1][k * C + c] = (y != 0 && y == a.sh - 1) ? cen : v_odd;   // sh == 1: y == 0 wins, as in the per-pixel kernel
        }
    }
    // A lane holds 2 * C contiguous floats of each of its two destination rows.  Stored straight from the lane (C = 3: a dwordx4 and a
    // dwordx2 at a 24-byte lane stride) every store instruction touches every cache line of the wave's 1.5 KB row segment with part of
    // its bytes; the same pattern cost pyrup_u8 a third of its time (r03r -> r03s).  So the rows go through a wave-private LDS row
    // and leave as contiguous 16-byte chunks: lane j stores chunk j (and chunk 64 + j) of the segment.
    __shared__ __attribute__((aligned(16))) float xpose[kBy][2][2 * C * kBx];
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const int wv = threadIdx.y;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 2 * C; ++i) xpose[wv][k][2 * C * lane + i] = out[k][i];
    __builtin_amdgcn_wave_barrier();
    if (yr >= a.sh) return;                                            // wave-uniform (one source row per wave)
    const int x0 = bx_ * kBx, nlive = min(kBx, a.sw - x0);            // live lanes of this wave: their floats are the valid segment
    const int seg = 2 * C * nlive;                                     // floats per destination row
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float* o = dst + ((long long)(2 * y + k) * a.dw + 2 * x0) * C;
        const float* xr = xpose[wv][k];
#pragma unroll
        for (int t = 0; t < (2 * C * kBx + 255) / 256; ++t) {
            const int f = 4 * (lane + kBx * t);
            typedef float f32x4a __attribute__((ext_vector_type(4)));   // LDS side: 16-byte aligned (ds_read_b128)
            if (f + 4 <= seg) { const f32x4a v = *reinterpret_cast<const f32x4a*>(xr + f); *reinterpret_cast<f32x4u*>(o + f) = f32x4u{v.x, v.y, v.z, v.w}; }
            else for (int i = f; i < seg; ++i) o[i] = xr[i];         // the segment's last, partial chunk (2 * C * nlive is even: at most 2 floats)
        }
    }
}

// pyrdown_u8 (:469-655): [1 4 6 4 1] rows into u16, then columns, (sum + 128) >> 8, min 255
template <int C>
__global__ __launch_bounds__(kBx* kBy) void pyrdown_u8_kernel(Pyr<uint8_t> a) {
    KH_PYR_PROLOGUE(uint8_t)
    int sx[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) sx[t] = reflect_101(2 * X + t - 2, a.sw) * C;
    const uint32_t wv[5] = {1, 4, 6, 4, 1};
    uint32_t sum[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sum[c] = 0;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        const uint8_t* row = src + (long long)reflect_101(2 * Y + ky - 2, a.sh) * a.sw * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const uint32_t h = (uint32_t)row[sx[0] + c] + 4u * row[sx[1] + c] + 6u * row[sx[2] + c] + 4u * row[sx[3] + c] + row[sx[4] + c];
            sum[c] += wv[ky] * h;  // h <= 4080 fits the reference's u16 intermediate
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = (uint8_t)min((sum[c] + 128u) >> 8, 255u);
}

// pyrdown_u8, tiled (round 2).  The per-pixel kernel above spends ~940 lane-operations per destination pixel (ten reflect_101
// modulos, 25 x C byte loads and their addresses): 12.8 ms per 256 4K RGB images, 0.08 of the HBM roofline.  Here a 256-thread block
// owns kPdTW x kPdTH destination pixels and
//   1. runs the reference's horizontal pass ([1 4 6 4 1] into u16) once per (source row, destination column) of the tile's
//      2 * kPdTH + 3 source rows, straight from global memory into LDS.  A thread takes two neighbouring destination pixels:
//      their windows overlap (7 source pixels = ceil(7C / 4) dwords, taken out of rows that interior tiles stage in LDS with
//      aligned dword loads; border tiles assemble the same dwords from reflected bytes), v_perm_b32 puts tap t of both pixels into the two 16-bit lanes of a register and the weights are
//      applied with packed 16-bit adds / shifts / multiplies (the sums are <= 4080).  LDS holds one plane per channel, a dword
//      = the pixel pair, so the writes are conflict-free (r02zb: the first version's three 2-byte writes per pixel at a 6-byte
//      lane stride kept the LDS busy 2.2 of 3.3 ms);
//   2. runs the vertical pass from LDS with the same packed arithmetic (<= 65280 + 128 fits a lane), four destination pixels per
//      thread, and stores C dwords.
// Same integers as the per-pixel kernel (exact sums, same final rounding), so the bytes are identical.
constexpr int kPdTW = 64, kPdTH = 16, kPdRows = 2 * kPdTH + 3, kPdPairs = kPdTW / 2;

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2_t as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2_t, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2_t v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2_t binomial5(u16x2_t t0, u16x2_t t1, u16x2_t t2, u16x2_t t3, u16x2_t t4) {
    const u16x2_t two = {2, 2}, six = {6, 6};
    return (t0 + t4) + ((t1 + t3) << two) + t2 * six;
}

constexpr bool pd_div_exact(int d, uint32_t m, int limit) {
    for (int i = 0; i < limit; ++i)
        if ((int)(((uint32_t)i * m) >> 20) != i / d) return false;
    return true;
}

template <int C>
__global__ __launch_bounds__(256) void pyrdown_u8_tile_kernel(Pyr<uint8_t> a) {
    // One LDS block, used twice: first the staged source rows S[kPdRows][kPitch] (bytes), then — after every thread has taken its
    // windows out of it — the horizontal pass H[kPdRows][C][kPdPairs] (a dword = a pixel pair's two 16-bit sums).
    constexpr int kWin = (2 * kPdTW + 3) * C;                                // bytes of a tile row's source window
    constexpr int kNdw = (kWin + 3 + 3) / 4 + 1;                              // dwords staged per row (any alignment, + the windows' read-ahead)
    constexpr int kPitch = 4 * kNdw + 8;
    constexpr int kLdsBytes = kPdRows * kPitch > kPdRows * C * kPdPairs * 4 ? kPdRows * kPitch : kPdRows * C * kPdPairs * 4;
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLdsBytes / 4];
    uint32_t (*H)[C][kPdPairs] = reinterpret_cast<uint32_t (*)[C][kPdPairs]>(lds);
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int tid = threadIdx.x;
    const int X0 = bx_ * kPdTW, Y0 = by_ * kPdTH;
    const int sx0 = 2 * X0 - 2, sy0 = 2 * Y0 - 2;                           // source coordinates of the tile's first tap
    const uint8_t* __restrict__ src = a.src + (long long)bz_ * a.ss;
    uint8_t* __restrict__ dst = a.dst + (long long)bz_ * a.ds;
    constexpr int ND = (7 * C + 3) / 4;                                      // dwords covering a pixel pair's 7-pixel window
    // interior (block-uniform): every tap inside the image, with a margin for the staged dwords that start up to 3 bytes before
    // and end up to 10 bytes after a row's window
    const bool interior = sx0 >= 4 && sy0 >= 0 && sx0 + 2 * kPdTW + 3 + (10 + C - 1) / C <= a.sw && sy0 + kPdRows <= a.sh;
    const int rows_needed = 2 * min(kPdTH, a.dh - Y0) + 3;
    const int p = tid & (kPdPairs - 1), slot = tid >> 5;                     // pixel pair, row slot (8 rows per trip)
    const bool pair_live = X0 + 2 * p < a.dw;
    constexpr int kBatch = 5;                                                // 8 * 5 >= kPdRows
    uint32_t wv[kBatch][ND];
    if (interior) {
        // Stage the rows with ALIGNED, lane-consecutive dword loads.  Loading each pair's window straight from global memory
        // (unaligned dwords, 4 * C bytes apart across lanes) cost ~52 L1 accesses per wave-load, and the L1's access rate was the
        // kernel's limit (r02zq: 1.7e9 TCP accesses = 2.8 of its 2.9 ms); an aligned 256-byte wave-load costs four.
        // Addresses: one 64-bit window base per block (wave-uniform) + 32-bit offsets, and the item -> (row, dword) split by an
        // exact 24-bit multiply-shift.  Round 2 did both in 64-bit / with a run-time division per load: ~8 quarter-rate
        // multiplies per staged dword in a kernel that is latency-bound to begin with (r03 ISA review).
        const uint8_t* g0 = src + ((long long)sy0 * a.sw + sx0) * C;      // first byte of the window's first row
        const uint32_t g0lo = (uint32_t)(uintptr_t)g0 & 3u, rowb = (uint32_t)(a.sw * C);   // row bytes < 2^24 (host-checked)
        auto row_mis = [&](int r) -> int { return (int)((g0lo + __umul24((uint32_t)r, rowb)) & 3u); };
        constexpr int kItems = kPdRows * kNdw, kTrips = (kItems + 255) / 256;
        constexpr uint32_t kDivM = ((1u << 20) + kNdw - 1) / kNdw;        // i / kNdw == (i * kDivM) >> 20 on the staging range (checked below)
        static_assert(pd_div_exact(kNdw, kDivM, kTrips * 256), "multiply-shift division by kNdw is not exact on the staging range");
        uint32_t v[kTrips];
#pragma unroll
        for (int k = 0; k < kTrips; ++k) {   // every load unconditional, from a clamped (row, dword): all in flight together
            const int i = min(tid + 256 * k, kItems - 1), ri = (int)(__umul24((uint32_t)i, kDivM) >> 20);
            const int r = min(ri, rows_needed - 1), d = i - ri * kNdw;
            const int off = (int)__umul24((uint32_t)r, rowb) - row_mis(r) + 4 * d;   // >= -3: the aligned dword holding the row's first byte
            v[k] = *reinterpret_cast<const uint32_t*>(g0 + off);
        }
#pragma unroll
        for (int k = 0; k < kTrips; ++k) {
            const int i = tid + 256 * k, ri = (int)(__umul24((uint32_t)i, kDivM) >> 20);
            if (i < kItems) lds[ri * (kPitch / 4) + (i - ri * kNdw)] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int r = min(slot + 8 * k, rows_needed - 1);
            const int mis = row_mis(r);
            const uint32_t* q = lds + r * (kPitch / 4) + p * C;              // the pair's window starts mis bytes into this dword
            uint32_t raw[ND + 1];
#pragma unroll
            for (int j = 0; j <= ND; ++j) raw[j] = q[j];
#pragma unroll
            for (int j = 0; j < ND; ++j) wv[k][j] = __builtin_amdgcn_alignbyte(raw[j + 1], raw[j], (uint32_t)mis);
        }
    } else {
        // every tap's address is valid after reflection, so these byte loads are unconditional too (dead pairs read pixel 0)
        int sx[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) sx[t] = pair_live ? reflect_101(sx0 + 4 * p + t, a.sw) * C : 0;
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int r = min(slot + 8 * k, rows_needed - 1);
            const uint8_t* row = src + (long long)reflect_101(sy0 + r, a.sh) * a.sw * C;
            uint32_t bytes[7 * C];
#pragma unroll
            for (int t = 0; t < 7; ++t)
#pragma unroll
                for (int c = 0; c < C; ++c) bytes[t * C + c] = row[sx[t] + c];
#pragma unroll
            for (int j = 0; j < ND; ++j) wv[k][j] = 0;
#pragma unroll
            for (int n = 0; n < 7 * C; ++n) wv[k][n >> 2] |= bytes[n] << (8 * (n & 3));
        }
    }
    uint32_t hv[kBatch][C];
#pragma unroll
    for (int k = 0; k < kBatch; ++k)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            u16x2_t t[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {  // lanes: tap j of the pair's first pixel, tap j of its second (two source pixels on)
                constexpr uint32_t kZero = 0x0c000c00u;
                const int n1 = j * C + c, n2 = (j + 2) * C + c;
                t[j] = as_u16x2(__builtin_amdgcn_perm(wv[k][n2 >> 2], wv[k][n1 >> 2], kZero | (uint32_t)(n1 & 3) | ((uint32_t)(4 + (n2 & 3)) << 16)));
            }
            hv[k][c] = as_u32(binomial5(t[0], t[1], t[2], t[3], t[4]));  // <= 4080: the reference's u16 intermediate
        }
    __syncthreads();   // the staged rows have been consumed: H takes their place
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        const int r = slot + 8 * k;
        if (r < rows_needed && pair_live) {
#pragma unroll
            for (int c = 0; c < C; ++c) H[r][c][p] = hv[k][c];
        }
    }
    __syncthreads();
    // vertical pass: thread = destination row ry, pixels 4q .. 4q + 3 (two pairs)
    const int ry = tid >> 4, q4 = tid & 15;
    const int Y = Y0 + ry, X = X0 + 4 * q4;
    if (Y >= a.dh || X >= a.dw) return;
    uint32_t o[4][C];  // [pixel][channel]
#pragma unroll
    for (int c = 0; c < C; ++c) {
        u16x2_t lo[5], hi[5];
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {
            const uint32_t* hp = &H[2 * ry + ky][c][2 * q4];
            lo[ky] = as_u16x2(hp[0]);
            hi[ky] = as_u16x2(hp[1]);
        }
        const u16x2_t half = {128, 128}, eight = {8, 8}, top = {255, 255};
        const u16x2_t v01 = __builtin_elementwise_min((binomial5(lo[0], lo[1], lo[2], lo[3], lo[4]) + half) >> eight, top);
        const u16x2_t v23 = __builtin_elementwise_min((binomial5(hi[0], hi[1], hi[2], hi[3], hi[4]) + half) >> eight, top);
        o[0][c] = v01[0]; o[1][c] = v01[1]; o[2][c] = v23[0]; o[3][c] = v23[1];
    }
    uint8_t* op = dst + ((long long)Y * a.dw + X) * C;
    if (X + 4 <= a.dw) {
        uint32_t w[C];
#pragma unroll
        for (int j = 0; j < C; ++j) w[j] = 0;
#pragma unroll
        for (int n = 0; n < 4 * C; ++n) w[n >> 2] |= o[n / C][n % C] << (8 * (n & 3));
#pragma unroll
        for (int j = 0; j < C; ++j) reinterpret_cast<u32_unaligned*>(op)[j] = w[j];
    } else {
        for (int px = 0; X + px < a.dw; ++px)
#pragma unroll
            for (int c = 0; c < C; ++c) op[px * C + c] = (uint8_t)o[px][c];
    }
}


// ---- pyrdown_u8 for RGB8, rolling wave, planar in registers (round 3) -----------------------------------------------------------
// pyrdown_u8_tile_kernel above is latency-bound (r02zp: vector ALUs 34 - 40 % busy, LDS 11 %, every wave slot taken, ~3 TB/s): its
// blocks load, barrier, compute, barrier, store.  This kernel has the structure of the round-3 u8 blur (kh_u8.hip,
// blur_u8_rgb_kernel): a WAVE walks down a strip with five source rows of loads in flight and no barrier at all; a lane owns EIGHT
// source pixels (24 bytes = two 12-byte quads; a 1.5 KB contiguous wave-load per row), de-interleaves them into two dwords per
// channel, takes the two border pixels of its neighbours by wave shifts, and runs the reference's [1 4 6 4 1] row pass on ADJACENT
// bytes with v_dot4_u32_u8 — 8 dot4 + 2 alignbyte per channel for its four destination pixels — into 16-bit lanes (<= 4080, the
// reference's u16 intermediate); the column pass is the same packed 16-bit arithmetic as the tile kernel (binomial5) on a five-row
// register ring, every second source row; four destination pixels leave as one 12-byte store.  ALL 64 lanes store — a wave's
// destination row segment is 256 pixels = 768 bytes = whole 128-byte lines (profiles/r03za: 744-byte segments, with lanes 0 / 63
// as halo lanes, split a line between two waves at every boundary and cost 19 % on a pure copy) — and the pixels either side of
// the wave come from one more quad load per row: the lower half's lanes load the quad before the wave's first pixel, the upper's
// the quad after its last, de-interleaved the same way and handed to the end lanes as the fill value of the DPP wave shifts.
// (Tried and dropped: chunked loads and stores through LDS, r03v / r03x — the L1 absorbs the 24-byte-stride loads.)
// reflect-101 borders: rows by reflecting the row index, columns (edge waves only) by loading from a clamped position and
// re-indexing with per-lane byte selectors.  Same integers as the per-pixel kernel: byte-identical (tests run both).  RGB8, sw >= 8.
constexpr int kPdRollWaveDst = 256;                  // destination pixels per wave (64 lanes x 4)
constexpr int kPdRollTileDst = 4 * kPdRollWaveDst;   // per 256-thread block
struct PyrRoll {
    const uint8_t* src;
    uint8_t* dst;
    int sw, sh, dw, dh, th;   // th = destination rows (pyrdown) / source rows (pyrup) per strip
    long long ss, ds;
    XcdTiles tiles;
    int plain;                // write-back instead of streaming stores (kh_common.h::plain_row_stores)
};

// C = 4 (round 6): RGBA images — eight source pixels are two 16-byte loads, four destination pixels one 16-byte store; 4 x 4 byte transposes
// (kh_common.h::deinterleave_quad) around the same per-channel code.  plain = 2: a destination off a dword (C = 4 only).
template <int C>
__global__ __launch_bounds__(256, C == 4 ? 3 : 4) void pyrdown_u8_rgb_roll_kernel(PyrRoll a) {   // (C = 4 at four waves per SIMD spills 28 dwords and runs 9 % slower)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int X0 = (int)tx * kPdRollTileDst + wv * kPdRollWaveDst;   // first destination pixel of this wave
    if (X0 >= a.dw) return;                                           // whole wave idle (no block barrier below)
    const int Y0 = ty * a.th, thr = min(a.th, a.dh - Y0);
    const uint8_t* __restrict__ src = a.src + (long long)bz * a.ss;
    uint8_t* __restrict__ dst = a.dst + (long long)bz * a.ds;
    // block-uniform: 12-byte quad offsets are dword-aligned and the image fits the V#'s 2 GiB window -> streaming stores (kh_common.h)
    const bool stream_ok = ((a.dw * C) & 3) == 0 && (long long)a.dw * a.dh * C <= 0x7fffffffLL && (C == 3 || a.plain != 2);
    const __amdgpu_buffer_rsrc_t out_win = stream_window(dst, (long long)a.dw * a.dh * C);
    const int p = 2 * X0 + 8 * lane;                                  // this lane's source pixels p .. p + 7
    const int ph = lane < 32 ? 2 * X0 - 4 : 2 * X0 + 2 * kPdRollWaveDst;   // the wave's halo quads: left in the lower half's lanes, right in the upper's
    const bool edge = 2 * X0 < 4 || 2 * X0 + 2 * kPdRollWaveDst + 4 > a.sw;   // wave-uniform: some lane's pixels need re-indexing
    const int pc = min(p, a.sw - 8), phc = min(max(ph, 0), a.sw - 4);  // where the pixels are loaded from (sw >= 8: host-checked)
    uint32_t selA = 0x03020100u, selB = 0x07060504u, selH = 0x03020100u;   // pixel j <- loaded pixel reflect_101(p + j) - pc (identity inside)
    if (edge) {
        selA = selB = selH = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            selA |= (uint32_t)min(max(reflect_101(p + j, a.sw) - pc, 0), 7) << (8 * j);
            selB |= (uint32_t)min(max(reflect_101(p + 4 + j, a.sw) - pc, 0), 7) << (8 * j);
            selH |= (uint32_t)min(max(reflect_101(ph + j, a.sw) - phc, 0), 3) << (8 * j);
        }
    }
    const int seg_bytes = C * min(kPdRollWaveDst, a.dw - X0);         // destination bytes of this wave per row (wave-uniform)
    const int rowb = a.sw * C;
    const int n = 2 * thr + 3;                                        // source rows walked: 2 Y0 - 2 .. 2 (Y0 + thr - 1) + 2
    int pf = 2 * Y0 - 2;

    uint32_t q[5][3 * C];  // five rows of raw loads in flight per lane: its eight pixels (2 C dwords) and its half-wave's halo quad (C dwords)
    auto prefetch = [&](uint32_t (&d)[3 * C]) {
        const uint8_t* row = src + (long long)reflect_101(pf, a.sh) * rowb;
        const uint8_t *rp = row + C * pc, *rh = row + C * phc;
        if constexpr (C == 4) {
            const u32x4_t v0 = *reinterpret_cast<const u32x4_unaligned*>(rp), v1 = *reinterpret_cast<const u32x4_unaligned*>(rp + 16), hq = *reinterpret_cast<const u32x4_unaligned*>(rh);
            d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
            d[8] = hq.x; d[9] = hq.y; d[10] = hq.z; d[11] = hq.w;
        } else {
#pragma unroll
            for (int k = 0; k < 2 * C; ++k) d[k] = *reinterpret_cast<const u32_unaligned*>(rp + 4 * k);
#pragma unroll
            for (int k = 0; k < C; ++k) d[2 * C + k] = *reinterpret_cast<const u32_unaligned*>(rh + 4 * k);
        }
        ++pf;
    };
#pragma unroll
    for (int i = 0; i < 5; ++i) prefetch(q[i]);

    u16x2_t ring[5][C][2];   // [row][channel][destination pixel pair]: the row pass, 16-bit lanes
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) { ring[i][c][0] = u16x2_t{0, 0}; ring[i][c][1] = u16x2_t{0, 0}; }

    long long out_off = (long long)Y0 * a.dw * C + C * (long long)X0;   // destination pixel X0 of destination row Y0
    for (int ib = 0; ib < n; ib += 10) {   // 10 = lcm(ring depth, row parity): ring slots and the emit test are compile-time
#pragma unroll
        for (int s = 0; s < 10; ++s) {
            const int i = ib + s, slot = s % 5;
            uint32_t d[3 * C];
#pragma unroll
            for (int k = 0; k < 3 * C; ++k) d[k] = q[slot][k];
            prefetch(q[slot]);
            // de-interleave: quad [R0 G0 B0 R1][G1 B1 R2 G2][B2 R3 G3 B3] -> one dword per channel (pixel j = byte j), three times
            uint32_t As[C], Bs[C], Hs[C];
            deinterleave_quad<C>(d, As);
            deinterleave_quad<C>(d + C, Bs);
            deinterleave_quad<C>(d + 2 * C, Hs);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                uint32_t A = As[c], B = Bs[c], Hq = Hs[c];
                if (edge) {   // wave-uniform
                    const uint32_t a0 = A, b0 = B;
                    A = __builtin_amdgcn_perm(b0, a0, selA);
                    B = __builtin_amdgcn_perm(b0, a0, selB);
                    Hq = __builtin_amdgcn_perm(0u, Hq, selH);
                }
                const uint32_t prevB = from_lane_below(B, Hq), nextA = from_lane_above(A, Hq);
                constexpr uint32_t kW = 0x04060401u;   // taps 1 4 6 4 on four adjacent bytes; the fifth tap (1) is a second dot4
                const uint32_t h0 = __builtin_amdgcn_udot4(A, 0x00010000u, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(A, prevB, 2), kW, 0u, false), false);
                const uint32_t h1 = __builtin_amdgcn_udot4(B, 0x00000001u, __builtin_amdgcn_udot4(A, kW, 0u, false), false);
                const uint32_t h2 = __builtin_amdgcn_udot4(B, 0x00010000u, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 2), kW, 0u, false), false);
                const uint32_t h3 = __builtin_amdgcn_udot4(nextA, 0x00000001u, __builtin_amdgcn_udot4(B, kW, 0u, false), false);
                ring[slot][c][0] = as_u16x2(h0 | (h1 << 16));   // <= 4080 each
                ring[slot][c][1] = as_u16x2(h2 | (h3 << 16));
            }
            if ((s & 1) == 0 && i >= 4 && i < n) {   // source row 2 Y + 2 is in: destination row Y = Y0 + (i - 4) / 2
                uint32_t v[C][2];
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u16x2_t half = {128, 128}, eight = {8, 8}, top = {255, 255};
                        const u16x2_t b5 = binomial5(ring[(s + 1) % 5][c][h], ring[(s + 2) % 5][c][h], ring[(s + 3) % 5][c][h], ring[(s + 4) % 5][c][h], ring[slot][c][h]);
                        v[c][h] = as_u32(__builtin_elementwise_min((b5 + half) >> eight, top));   // pixels (2h, 2h + 1) in bytes 0 and 2
                    }
                // re-interleave; 4 C bytes straight from the owning lane
                uint32_t w[C];
                if constexpr (C == 3) {
                    const uint32_t rg01 = __builtin_amdgcn_perm(v[1][0], v[0][0], 0x06020400u);   // R0 G0 R1 G1
                    const uint32_t rg23 = __builtin_amdgcn_perm(v[1][1], v[0][1], 0x06020400u);   // R2 G2 R3 G3
                    w[0] = __builtin_amdgcn_perm(v[2][0], rg01, 0x02040100u);        // R0 G0 B0 R1
                    w[1] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(v[2][0], rg01, 0x0c0c0603u), rg23, 0x01000504u);   // G1 B1 | R2 G2
                    w[2] = __builtin_amdgcn_perm(v[2][1], rg23, 0x06030204u);        // B2 R3 G3 B3
                } else {
                    uint32_t pl[C];   // one dword per channel (pixel j = byte j) from the pairs in bytes 0 and 2
#pragma unroll
                    for (int c = 0; c < C; ++c) pl[c] = __builtin_amdgcn_perm(v[c][1], v[c][0], 0x06040200u);
                    interleave_quad<C>(pl, w);
                }
                uint8_t* o = dst + out_off;
                const int off = 4 * C * lane;
                if (off + 4 * C <= seg_bytes && stream_ok) {
                    row_store<C>(out_win, (int)out_off + off, w, a.plain);
                } else if (off + 4 * C <= seg_bytes) {
#pragma unroll
                    for (int k = 0; k < C; ++k) *reinterpret_cast<u32_unaligned*>(o + off + 4 * k) = w[k];
                } else {
#pragma unroll
                    for (int b = 0; b < 3 * C; ++b)   // at most three pixels of a quad that reaches past the last destination column
                        if (off + b < seg_bytes) o[off + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
                }
                out_off += (long long)a.dw * C;
            }
        }
    }
}


// ---- pyrdown_u8 for single-channel images, rolling wave (round 6) ---------------------------------------------------------------------
// Gray pyramids (optical flow, feature detection) took pyrdown_u8_tile_kernel at 0.32 of peak against 0.54 for the RGB kernel above.
// The same walk on one plane: a lane owns SIXTEEN source pixels of a row (one 16-byte load, 1 KiB per wave and row) and their EIGHT
// destination pixels (one 8-byte store, 512 bytes = four whole lines per wave and row); its dword pairs (0, 1) and (2, 3) are the
// RGB kernel's (A, B) of one channel — the same [1 4 6 4 1] row pass on adjacent bytes with v_dot4_u32_u8, the same packed 16-bit
// column pass on a five-row register ring — with the dword before the lane's first and after its last from the neighbouring lanes
// by wave shifts and, at the ends of the wave, from one halo dword per half-wave.  reflect-101 borders: rows by reflecting the row
// index; columns -2, -1 and sw by re-indexing the halo dword / the row's last dword with one v_perm_b32 (edge waves only).  For
// source widths that are multiples of 16; byte-identical to the other kernels (same integers).
constexpr int kPgRollWaveDst = 512;                  // destination pixels per wave (64 lanes x 8)
constexpr int kPgRollTileDst = 4 * kPgRollWaveDst;   // per 256-thread block
// RAGGED (round 6): any source width >= 16 and any alignment (remap16_* above); plain = 2: unaligned global stores.
template <bool RAGGED>
__global__ __launch_bounds__(256, 4) void pyrdown_u8_gray_roll_kernel(PyrRoll a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int X0 = (int)tx * kPgRollTileDst + wv * kPgRollWaveDst;   // first destination pixel of this wave
    if (X0 >= a.dw) return;                                           // whole wave idle (no block barrier below)
    const int Y0 = ty * a.th, thr = min(a.th, a.dh - Y0);
    const uint8_t* __restrict__ src = a.src + (long long)bz * a.ss;
    uint8_t* __restrict__ dst = a.dst + (long long)bz * a.ds;
    const __amdgpu_buffer_rsrc_t out_win = stream_window(dst, (long long)a.dw * a.dh);   // (dw * dh < 2^31: host-checked)
    const int p = 2 * X0 + 16 * lane;                                 // this lane's source pixels p .. p + 15
    const int nvalid = min(max(a.dw - (X0 + 8 * lane), 0), 8);        // destination pixels of this lane (RAGGED: 0 .. 8; otherwise 0 or 8, sw % 16 == 0)
    const bool inside = nvalid > 0;
    const int ph = lane < 32 ? 2 * X0 - 4 : 2 * X0 + 2 * kPgRollWaveDst;   // the wave's halo dwords: left in the lower half's lanes, right in the upper's
    const bool edge = 2 * X0 < 4 || 2 * X0 + 2 * kPgRollWaveDst + 4 > a.sw;   // wave-uniform
    const int pc = min(p, a.sw - 16), phc = min(max(ph, 0), a.sw - 4);
    // the first lane past the row end supplies pixel sw of its inside neighbour's last window: its first dword <- the row's last dword re-indexed
    uint32_t esel = 0x03020100u, selH = 0x03020100u;
    if (edge) {
        esel = selH = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            esel |= (uint32_t)min(max(reflect_101(p + j, a.sw) - (a.sw - 4), 0), 3) << (8 * j);
            selH |= (uint32_t)min(max(reflect_101(ph + j, a.sw) - phc, 0), 3) << (8 * j);
        }
    }
    Remap16 rm{};
    if (RAGGED && edge) rm = remap16_setup(p, pc, [&](int x) { return reflect_101(x, a.sw); });
    const int n = 2 * thr + 3;                                        // source rows walked: 2 Y0 - 2 .. 2 (Y0 + thr - 1) + 2
    int pf = 2 * Y0 - 2;

    uint32_t q[5][5];  // five rows of raw loads in flight per lane: its sixteen pixels and its half-wave's halo dword
    auto prefetch = [&](uint32_t (&d)[5]) {
        const uint8_t* row = src + (long long)reflect_101(pf, a.sh) * a.sw;
        const u32x4_t v = *reinterpret_cast<const u32x4_unaligned*>(row + pc);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        d[4] = *reinterpret_cast<const u32_unaligned*>(row + phc);
        ++pf;
    };
#pragma unroll
    for (int i = 0; i < 5; ++i) prefetch(q[i]);

    u16x2_t ring[5][2][2];   // [row][dword pair][destination pixel pair]: the row pass, 16-bit lanes
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) { ring[i][c][0] = u16x2_t{0, 0}; ring[i][c][1] = u16x2_t{0, 0}; }

    int out_off = Y0 * a.dw + X0 + 8 * lane;
    for (int ib = 0; ib < n; ib += 10) {   // 10 = lcm(ring depth, row parity): ring slots and the emit test are compile-time
#pragma unroll
        for (int s = 0; s < 10; ++s) {
            const int i = ib + s, slot = s % 5;
            uint32_t cur[4] = {q[slot][0], q[slot][1], q[slot][2], q[slot][3]}, halo = q[slot][4];
            prefetch(q[slot]);
            if (edge) {   // wave-uniform
                if constexpr (RAGGED) remap16_apply(rm, cur, 0u);
                else {
                    const uint32_t beyond = __builtin_amdgcn_perm(0u, cur[3], esel);   // (of a lane past the row end: the loaded sixteen are the row's last)
                    cur[0] = inside ? cur[0] : beyond;
                }
                halo = __builtin_amdgcn_perm(0u, halo, selH);
            }
            const uint32_t prevd = from_lane_below(cur[3], halo), nextd = from_lane_above(cur[0], halo);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint32_t A = cur[2 * c], B = cur[2 * c + 1], prevB = c == 0 ? prevd : cur[1], nextA = c == 0 ? cur[2] : nextd;
                constexpr uint32_t kW = 0x04060401u;   // taps 1 4 6 4 on four adjacent bytes; the fifth tap (1) is a second dot4
                const uint32_t h0 = __builtin_amdgcn_udot4(A, 0x00010000u, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(A, prevB, 2), kW, 0u, false), false);
                const uint32_t h1 = __builtin_amdgcn_udot4(B, 0x00000001u, __builtin_amdgcn_udot4(A, kW, 0u, false), false);
                const uint32_t h2 = __builtin_amdgcn_udot4(B, 0x00010000u, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 2), kW, 0u, false), false);
                const uint32_t h3 = __builtin_amdgcn_udot4(nextA, 0x00000001u, __builtin_amdgcn_udot4(B, kW, 0u, false), false);
                ring[slot][c][0] = as_u16x2(h0 | (h1 << 16));   // <= 4080 each
                ring[slot][c][1] = as_u16x2(h2 | (h3 << 16));
            }
            if ((s & 1) == 0 && i >= 4 && i < n) {   // source row 2 Y + 2 is in: destination row Y = Y0 + (i - 4) / 2
                uint32_t w[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    uint32_t v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u16x2_t half = {128, 128}, eight = {8, 8}, top = {255, 255};
                        const u16x2_t b5 = binomial5(ring[(s + 1) % 5][c][h], ring[(s + 2) % 5][c][h], ring[(s + 3) % 5][c][h], ring[(s + 4) % 5][c][h], ring[slot][c][h]);
                        v[h] = as_u32(__builtin_elementwise_min((b5 + half) >> eight, top));   // pixels (2h, 2h + 1) in bytes 0 and 2
                    }
                    w[c] = __builtin_amdgcn_perm(v[1], v[0], 0x06040200u);   // four destination pixels
                }
                if constexpr (RAGGED) {
                    if (nvalid == 8 && a.plain != 2) row_store<2>(out_win, out_off, w, a.plain);
                    else if (nvalid == 8) *reinterpret_cast<u64_unaligned*>(dst + out_off) = ((uint64_t)w[1] << 32) | w[0];
                    else if (nvalid > 0) { const uint32_t w4[4] = {w[0], w[1], 0u, 0u}; store_head_bytes(dst + out_off, w4, nvalid); }
                } else if (inside) row_store<2>(out_win, out_off, w, a.plain);
                out_off += a.dw;
            }
        }
    }
}

// ---- pyrup_u8 for RGB8, rolling wave, planar in registers (round 3) ---------------------------------------------------------------
// pyrup_u8_pair_kernel above is VALU-bound (r02zp: 71 % busy, 1.06 G instructions per 256 4K outputs).  Same structure as the rolling
// pyrdown above: a WAVE walks down a strip of SOURCE rows with three rows of loads in flight; a lane owns four source pixels (12
// bytes), de-interleaves them into one dword per channel and takes the neighbouring pixels by wave shifts.  Row pass per channel:
//   even destination columns  (p[x-1] + 6 p[x] + p[x+1] + 4) >> 3 : v_dot4_u32_u8 of the four-byte window with the taps pre-multiplied by
//                             32 — (32 t) has t >> 3 in byte 1 for t < 2048 — so the four results are gathered by three v_perm_b32;
//   odd destination columns   (p[x] + p[x+1] + 1) >> 1            : a bytewise rounding average of two dwords, (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7f),
//                             four pixels in five instructions;
// column pass on the three-row ring: odd destination rows are the same bytewise average of two packed rows, even rows the [1 6 1] sum
// in 16-bit lanes (rows kept unpacked in the ring).  Two destination rows per source row; a wave's destination row segment is 512 pixels
// = 1536 bytes = whole 128-byte lines, written as 96 contiguous 16-byte chunks through a wave-private LDS row (below).  All 64 lanes
// produce output: the source pixels either side of the wave come from one more quad load per row (lower half's lanes: the quad before
// the wave's first pixel, upper half's: the quad after its last) handed to the end lanes as the fill value of the DPP wave shifts
// (the first version kept lanes 0 / 63 as halo lanes: 1488-byte segments that split a line between two waves at every boundary).
// reflect-101 columns on edge waves by re-indexing clamped quads (one byte selector per lane).  Same integers as the per-pixel kernel:
// byte-identical (tests run both).  RGB8, sw >= 4.
constexpr int kPuRollWaveSrc = 256;                  // source pixels per wave (64 lanes x 4)
constexpr int kPuRollTileSrc = 4 * kPuRollWaveSrc;

__device__ __forceinline__ uint32_t avg_round_u8x4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }   // per byte (a + b + 1) >> 1
// AR (round 6): the same walks serve resize_fast_u8's EXACT 2x bilinear upscale, which ran per pixel with byte loads at 0.09 of peak
// (1080p -> 4K RGB 0.73 ms per 16 images).  A destination pixel pair (2x, 2x + 1) of that resize is centred on source pixel x exactly as
// in pyrup; what differs is the arithmetic and the border (replicate instead of reflect-101, which reproduces the reference's first /
// last column and row rules):
//   kUpRh  — the reference's RGB special case (P/resize/kernels.rs:166-183, blend_75_25_row): rounding-halving chains,
//            even = rh(c, rh(prev, c)), odd = rh(c, rh(c, next)), on packed bytes (four pixels per instruction), both axes;
//   kUpQ14 — every other channel count takes the generic Q14 bilinear, whose weights at this scale are 4096 / 12288 for every pixel:
//            ((3 a + b) 4096 x 12288-or-4096 ... + 2^27) >> 28 == (9 a + 3 b + 3 c + d + 8) >> 4 exactly; the horizontal 3 c + n stays
//            unrounded in 16-bit lanes (<= 1020), the vertical pass adds, rounds and shifts once.
//   kUpNearest — the nearest resize at this scale: column floor((X + 0.5) / 2) = X >> 1, i.e. both pixels of a pair and both rows ARE the source pixel.
//   kUpCv — resize_opencv_u8's INTER_LINEAR at this scale (opencv_compat.rs:139-195): coefficients 512 / 1536 of 2048, the horizontal sums as in
//            kUpQ14 (h = 3 c + n: `s >> 4` drops zero bits), the vertical ((h_far >> 2) + ((3 h_near) >> 2) + 2) >> 2 with the reference's truncations,
//            and (h + 2) >> 2 on the first / last destination row, where its coefficients are (2048, 0).
enum { kUpPyr = 0, kUpRh = 1, kUpQ14 = 2, kUpNearest = 3, kUpCv = 4 };
template <int AR> __device__ __forceinline__ int up_index(int i, int len) { return AR == kUpPyr ? reflect_101(i, len) : min(max(i, 0), len - 1); }

// C = 4 (round 6): RGBA images — a 16-byte source quad per lane, 32 destination bytes per lane and row through the same LDS transposition.
template <int C, int AR = kUpPyr>
__global__ __launch_bounds__(256, C == 4 ? 3 : 4) void pyrup_u8_rgb_roll_kernel(PyrRoll a) {   // th = SOURCE rows per strip here
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int x0 = (int)tx * kPuRollTileSrc + wv * kPuRollWaveSrc;   // first source pixel of this wave
    if (x0 >= a.sw) return;
    const int y0 = ty * a.th, thr = min(a.th, a.sh - y0);
    const uint8_t* __restrict__ src = a.src + (long long)bz * a.ss;
    uint8_t* __restrict__ dst = a.dst + (long long)bz * a.ds;
    const bool stream_ok = ((a.dw * C) & 3) == 0 && (long long)a.dw * a.dh * C <= 0x7fffffffLL && (C == 3 || a.plain != 2);   // block-uniform (see pyrdown above)
    const __amdgpu_buffer_rsrc_t out_win = stream_window(dst, (long long)a.dw * a.dh * C);
    const int p = x0 + 4 * lane;                                      // this lane's source pixels p .. p + 3
    const int ph = lane < 32 ? x0 - 4 : x0 + kPuRollWaveSrc;          // the wave's halo quads: left in the lower half's lanes, right in the upper's
    const bool edge = x0 < 4 || x0 + kPuRollWaveSrc + 4 > a.sw;       // wave-uniform
    const int pc = min(p, a.sw - 4), phc = min(max(ph, 0), a.sw - 4); // sw >= 4: host-checked
    uint32_t esel = 0x03020100u, hsel = 0x03020100u;
    if (edge) {
        esel = hsel = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            esel |= (uint32_t)min(max(up_index<AR>(p + j, a.sw) - pc, 0), 3) << (8 * j);
            hsel |= (uint32_t)min(max(up_index<AR>(ph + j, a.sw) - phc, 0), 3) << (8 * j);
        }
    }
    const int rowb = a.sw * C;
    const long long drow = (long long)a.dw * C;
    const int seg_bytes = 2 * C * min(kPuRollWaveSrc, a.sw - x0);    // destination bytes of this wave per row (wave-uniform)
    __shared__ __attribute__((aligned(16))) uint32_t xpose[4][2][2 * C * 64];   // per wave, per destination row of a step: 64 lanes x 8 C bytes
    const int n = thr + 2;                                            // source rows walked: y0 - 1 .. y0 + thr
    int pf = y0 - 1;

    uint32_t q[3][2 * C];   // the lane's quad and its half-wave's halo quad
    auto prefetch = [&](uint32_t (&d)[2 * C]) {
        const uint8_t* row = src + (long long)up_index<AR>(pf, a.sh) * rowb;
        const uint8_t *rp = row + C * pc, *rh = row + C * phc;
        if constexpr (C == 4) {
            const u32x4_t v = *reinterpret_cast<const u32x4_unaligned*>(rp), hq = *reinterpret_cast<const u32x4_unaligned*>(rh);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; d[4] = hq.x; d[5] = hq.y; d[6] = hq.z; d[7] = hq.w;
        } else {
            d[0] = *reinterpret_cast<const u32_unaligned*>(rp); d[1] = *reinterpret_cast<const u32_unaligned*>(rp + 4); d[2] = *reinterpret_cast<const u32_unaligned*>(rp + 8);
            d[3] = *reinterpret_cast<const u32_unaligned*>(rh); d[4] = *reinterpret_cast<const u32_unaligned*>(rh + 4); d[5] = *reinterpret_cast<const u32_unaligned*>(rh + 8);
        }
        ++pf;
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) prefetch(q[i]);

    // the row pass of three source rows: [row][channel][even / odd destination columns], packed bytes, and unpacked 16-bit lanes
    uint32_t hp_[3][C][2], hl[3][C][2], hh[3][C][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int e = 0; e < 2; ++e) { hp_[i][c][e] = 0; hl[i][c][e] = 0; hh[i][c][e] = 0; }

    long long row_off = (long long)(2 * y0) * drow + 2 * C * (long long)x0;   // destination pixel 2 x0 of destination row 2 y0
    for (int ib = 0; ib < n; ib += 3) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int i = ib + s;
            uint32_t dq[2 * C];
#pragma unroll
            for (int k = 0; k < 2 * C; ++k) dq[k] = q[s][k];
            prefetch(q[s]);
            uint32_t As[C], Hs[C];
            deinterleave_quad<C>(dq, As);       // each channel's four pixels
            deinterleave_quad<C>(dq + C, Hs);   // and the halo quad's
#pragma unroll
            for (int c = 0; c < C; ++c) {
                uint32_t A = As[c], Hq = Hs[c];
                if (edge) { A = __builtin_amdgcn_perm(0u, A, esel); Hq = __builtin_amdgcn_perm(0u, Hq, hsel); }
                const uint32_t prev = from_lane_below(A, Hq), next = from_lane_above(A, Hq);
                const uint32_t w2 = __builtin_amdgcn_alignbyte(next, A, 1);                 // p[x+1] for the four pixels
                if constexpr (AR != kUpPyr) {
                    const uint32_t wm = __builtin_amdgcn_alignbyte(A, prev, 3);             // p[x-1]
                    if constexpr (AR == kUpNearest) {
                        hp_[s][c][0] = A; hp_[s][c][1] = A;
                    } else if constexpr (AR == kUpRh) {
                        hp_[s][c][0] = avg_round_u8x4(A, avg_round_u8x4(wm, A)); hp_[s][c][1] = avg_round_u8x4(A, avg_round_u8x4(A, w2));
                    } else {   // 3 c + neighbour, unrounded, even / odd bytes in 16-bit lanes
                        const uint32_t al = A & 0x00ff00ffu, ah = (A >> 8) & 0x00ff00ffu;
                        hl[s][c][0] = mad24(al, 3u, wm & 0x00ff00ffu); hh[s][c][0] = mad24(ah, 3u, (wm >> 8) & 0x00ff00ffu);
                        hl[s][c][1] = mad24(al, 3u, w2 & 0x00ff00ffu); hh[s][c][1] = mad24(ah, 3u, (w2 >> 8) & 0x00ff00ffu);
                    }
                    continue;
                }
                constexpr uint32_t kT = 0x0020c020u;   // taps (1, 6, 1, 0) x 32; accumulator 4 x 32
                const uint32_t t0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(A, prev, 3), kT, 128u, false);
                const uint32_t t1 = __builtin_amdgcn_udot4(A, kT, 128u, false);
                const uint32_t t2 = __builtin_amdgcn_udot4(w2, kT, 128u, false);
                const uint32_t t3 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(next, A, 2), kT, 128u, false);
                const uint32_t ev = __builtin_amdgcn_perm(__builtin_amdgcn_perm(t3, t2, 0x0c0c0501u), __builtin_amdgcn_perm(t1, t0, 0x0c0c0501u), 0x05040100u);
                const uint32_t od = avg_round_u8x4(A, w2);
                hp_[s][c][0] = ev; hp_[s][c][1] = od;
                hl[s][c][0] = ev & 0x00ff00ffu; hh[s][c][0] = (ev >> 8) & 0x00ff00ffu;
                hl[s][c][1] = od & 0x00ff00ffu; hh[s][c][1] = (od >> 8) & 0x00ff00ffu;
            }
            if (i >= 2 && i < n) {   // rows y - 1, y, y + 1 are in: destination rows 2 y and 2 y + 1 (y = y0 + i - 2)
                const int sp = (s + 1) % 3, sc = (s + 2) % 3, sn = s;   // compile-time after unrolling
                [[maybe_unused]] const int yy = y0 + i - 2;   // the source row these two destination rows are centred on
                uint32_t ve[C][2], vo[C][2];
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if constexpr (AR == kUpNearest) {
                            ve[c][e] = hp_[sc][c][e]; vo[c][e] = hp_[sc][c][e];
                            continue;
                        } else if constexpr (AR == kUpRh) {
                            ve[c][e] = avg_round_u8x4(hp_[sc][c][e], avg_round_u8x4(hp_[sp][c][e], hp_[sc][c][e]));
                            vo[c][e] = avg_round_u8x4(hp_[sc][c][e], avg_round_u8x4(hp_[sc][c][e], hp_[sn][c][e]));
                            continue;
                        } else if constexpr (AR == kUpQ14) {
                            const uint32_t cl3 = hl[sc][c][e] + (hl[sc][c][e] << 1), ch3 = hh[sc][c][e] + (hh[sc][c][e] << 1);   // (10-bit lanes: beyond mad24's 24-bit operands)
                            const uint32_t el = ((cl3 + hl[sp][c][e] + 0x00080008u) >> 4) & 0x00ff00ffu, eh = ((ch3 + hh[sp][c][e] + 0x00080008u) >> 4) & 0x00ff00ffu;
                            const uint32_t ol = ((cl3 + hl[sn][c][e] + 0x00080008u) >> 4) & 0x00ff00ffu, oh = ((ch3 + hh[sn][c][e] + 0x00080008u) >> 4) & 0x00ff00ffu;
                            ve[c][e] = el | (eh << 8); vo[c][e] = ol | (oh << 8);
                            continue;
                        } else if constexpr (AR == kUpCv) {
                            constexpr uint32_t kS = 0x3fff3fffu, kB = 0x00ff00ffu, kR = 0x00020002u;   // per-lane >> 2, byte mask, rounding
                            const uint32_t cl = hl[sc][c][e], ch = hh[sc][c][e];
                            const uint32_t nl = ((cl + (cl << 1)) >> 2) & kS, nh = ((ch + (ch << 1)) >> 2) & kS;   // (3 h_near) >> 2
                            const bool first = yy == 0, last = yy == a.sh - 1;   // wave-uniform
                            const uint32_t el = first ? cl + kR : ((hl[sp][c][e] >> 2) & kS) + nl + kR, eh = first ? ch + kR : ((hh[sp][c][e] >> 2) & kS) + nh + kR;
                            const uint32_t ol = last ? cl + kR : ((hl[sn][c][e] >> 2) & kS) + nl + kR, oh = last ? ch + kR : ((hh[sn][c][e] >> 2) & kS) + nh + kR;
                            ve[c][e] = ((el >> 2) & kB) | (((eh >> 2) & kB) << 8); vo[c][e] = ((ol >> 2) & kB) | (((oh >> 2) & kB) << 8);
                            continue;
                        }
                        const uint32_t lo = ((mad24(hl[sc][c][e], 6u, hl[sp][c][e]) + hl[sn][c][e] + 0x00040004u) >> 3) & 0x00ff00ffu;
                        const uint32_t hi = ((mad24(hh[sc][c][e], 6u, hh[sp][c][e]) + hh[sn][c][e] + 0x00040004u) >> 3) & 0x00ff00ffu;
                        ve[c][e] = lo | (hi << 8);
                        vo[c][e] = avg_round_u8x4(hp_[sc][c][e], hp_[sn][c][e]);
                    }
                // Every lane re-interleaves its eight destination pixels (24 bytes) of both rows; the bytes then go through a wave-private
                // LDS row so that a store INSTRUCTION covers contiguous memory — lane j writes 16-byte chunk j of the wave's row segment.
                // Written straight from the owning lanes (three 8-byte pieces at a 24-byte lane stride) every store touched every cache line
                // of the segment with a third of its bytes: 2.63 ms per 256 4K outputs, the vector ALUs 28 % busy (r03r).
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    uint32_t pl[C][2];   // per channel: destination pixels 0..3 and 4..7 (even / odd columns merged)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const uint32_t ee = r ? vo[c][0] : ve[c][0], oo = r ? vo[c][1] : ve[c][1];
                        pl[c][0] = __builtin_amdgcn_perm(oo, ee, 0x05010400u);   // e0 o0 e1 o1
                        pl[c][1] = __builtin_amdgcn_perm(oo, ee, 0x07030602u);   // e2 o2 e3 o3
                    }
                    uint32_t* xr = xpose[wv][r];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {   // re-interleave four pixels: [R0 G0 B0 R1][G1 B1 R2 G2][B2 R3 G3 B3]
                        uint32_t ph_[C], wq[C];
#pragma unroll
                        for (int c = 0; c < C; ++c) ph_[c] = pl[c][h];
                        interleave_quad<C>(ph_, wq);
#pragma unroll
                        for (int k = 0; k < C; ++k) xr[2 * C * lane + C * h + k] = wq[k];
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint8_t* xb = reinterpret_cast<const uint8_t*>(xpose[wv][r]);        // lane 0's first byte = destination pixel 2 x0
                    uint8_t* o = dst + row_off + r * drow;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int off = 16 * (lane + 64 * t);
                        if (off + 16 <= seg_bytes) {
                            const uint64_t lo = *reinterpret_cast<const uint64_t*>(xb + off), hi = *reinterpret_cast<const uint64_t*>(xb + off + 8);
                            if (stream_ok) {
                                const uint32_t w[4] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
                                row_store<4>(out_win, (int)(row_off + r * drow) + off, w, a.plain);
                            } else {
                                *reinterpret_cast<u64_unaligned*>(o + off) = lo; *reinterpret_cast<u64_unaligned*>(o + off + 8) = hi;
                            }
                        } else if (off < seg_bytes) {   // the segment's last, partial chunk (one lane)
                            for (int b = off; b < seg_bytes; ++b) o[b] = xb[b];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();   // the rows are out before the next step overwrites them (DS operations of a wave are ordered)
                row_off += 2 * drow;
            }
        }
    }
}

// ---- pyrup_u8 for single-channel images, rolling wave (round 6) -----------------------------------------------------------------------
// The gray twin of the kernel above (the pair kernel ran at 0.29 of peak): a lane owns EIGHT source pixels of a row (one 8-byte load)
// and their sixteen destination pixels of both destination rows (one 16-byte store each, 1 KiB = whole lines per wave and row —
// straight from the owning lanes, no LDS transposition: one plane needs no re-interleaving).  Its two dwords go through the RGB
// kernel's per-channel code; the dword before the lane's first and after its last come from the neighbouring lanes by wave shifts
// and, at the ends of the wave, from one halo dword per half-wave; reflect-101 columns -1 and sw by one v_perm_b32 (edge waves).
// For source widths that are multiples of 8; byte-identical to the other kernels.
constexpr int kPuGrayWaveSrc = 512;                  // source pixels per wave (64 lanes x 8)
constexpr int kPuGrayTileSrc = 4 * kPuGrayWaveSrc;
// RAGGED (round 6): any source width >= 8 and any alignment — a lane's eight loaded bytes re-indexed by ONE v_perm_b32 per dword; plain = 2: unaligned stores.
template <bool RAGGED, int AR = kUpPyr>
__global__ __launch_bounds__(256, 4) void pyrup_u8_gray_roll_kernel(PyrRoll a) {   // th = SOURCE rows per strip
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int x0 = (int)tx * kPuGrayTileSrc + wv * kPuGrayWaveSrc;   // first source pixel of this wave
    if (x0 >= a.sw) return;
    const int y0 = ty * a.th, thr = min(a.th, a.sh - y0);
    const uint8_t* __restrict__ src = a.src + (long long)bz * a.ss;
    uint8_t* __restrict__ dst = a.dst + (long long)bz * a.ds;
    const __amdgpu_buffer_rsrc_t out_win = stream_window(dst, (long long)a.dw * a.dh);   // (dw * dh < 2^31: host-checked)
    const int p = x0 + 8 * lane;                                      // this lane's source pixels p .. p + 7
    const int nvalid = min(max(a.sw - p, 0), 8);                      // source pixels of this lane (RAGGED: 0 .. 8; otherwise 0 or 8, sw % 8 == 0)
    const bool inside = nvalid > 0;
    const int ph = lane < 32 ? x0 - 4 : x0 + kPuGrayWaveSrc;          // the wave's halo dwords: left in the lower half's lanes, right in the upper's
    const bool edge = x0 < 4 || x0 + kPuGrayWaveSrc + 4 > a.sw;       // wave-uniform
    const int pc = min(p, a.sw - 8), phc = min(max(ph, 0), a.sw - 4);
    uint32_t esel = 0x03020100u, hsel = 0x03020100u;   // (esel: the first lane past the row end re-indexes the row's last dword into its first)
    if (edge) {
        esel = hsel = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            esel |= (uint32_t)min(max(up_index<AR>(p + j, a.sw) - (a.sw - 4), 0), 3) << (8 * j);
            hsel |= (uint32_t)min(max(up_index<AR>(ph + j, a.sw) - phc, 0), 3) << (8 * j);
        }
    }
    uint32_t rsel[2] = {0x03020100u, 0x07060504u};   // RAGGED: lane byte j <- loaded byte reflect_101(p + j) - pc of the eight
    if (RAGGED && edge) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            rsel[c] = 0;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) rsel[c] |= (uint32_t)min(max(up_index<AR>(p + 4 * c + j2, a.sw) - pc, 0), 7) << (8 * j2);
        }
    }
    const int n = thr + 2;                                            // source rows walked: y0 - 1 .. y0 + thr
    int pf = y0 - 1;

    uint32_t q[3][3];   // the lane's eight pixels and its half-wave's halo dword
    auto prefetch = [&](uint32_t (&d)[3]) {
        const uint8_t* row = src + (long long)up_index<AR>(pf, a.sh) * a.sw;
        d[0] = *reinterpret_cast<const u32_unaligned*>(row + pc); d[1] = *reinterpret_cast<const u32_unaligned*>(row + pc + 4);
        d[2] = *reinterpret_cast<const u32_unaligned*>(row + phc);
        ++pf;
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) prefetch(q[i]);

    // the row pass of three source rows: [row][dword][even / odd destination columns], packed bytes, and unpacked 16-bit lanes
    uint32_t hp_[3][2][2], hl[3][2][2], hh[3][2][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 2; ++e) { hp_[i][c][e] = 0; hl[i][c][e] = 0; hh[i][c][e] = 0; }

    int row_off = 2 * y0 * a.dw + 2 * p;   // this lane's sixteen destination pixels of destination row 2 y0
    for (int ib = 0; ib < n; ib += 3) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int i = ib + s;
            uint32_t cur[2] = {q[s][0], q[s][1]}, halo = q[s][2];
            prefetch(q[s]);
            if (edge) {   // wave-uniform
                if constexpr (RAGGED) {
                    const uint32_t l0 = cur[0], l1 = cur[1];
                    cur[0] = __builtin_amdgcn_perm(l1, l0, rsel[0]); cur[1] = __builtin_amdgcn_perm(l1, l0, rsel[1]);
                } else {
                    const uint32_t beyond = __builtin_amdgcn_perm(0u, cur[1], esel);   // (of a lane past the row end: the loaded eight are the row's last)
                    cur[0] = inside ? cur[0] : beyond;
                }
                halo = __builtin_amdgcn_perm(0u, halo, hsel);
            }
            const uint32_t prevd = from_lane_below(cur[1], halo), nextd = from_lane_above(cur[0], halo);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint32_t A = cur[c], prev = c == 0 ? prevd : cur[0], next = c == 0 ? cur[1] : nextd;
                const uint32_t w2 = __builtin_amdgcn_alignbyte(next, A, 1);                 // p[x+1] for the four pixels
                if constexpr (AR != kUpPyr) {   // (see kUpRh / kUpQ14 above)
                    const uint32_t wm = __builtin_amdgcn_alignbyte(A, prev, 3);             // p[x-1]
                    if constexpr (AR == kUpNearest) {
                        hp_[s][c][0] = A; hp_[s][c][1] = A;
                    } else if constexpr (AR == kUpRh) {
                        hp_[s][c][0] = avg_round_u8x4(A, avg_round_u8x4(wm, A)); hp_[s][c][1] = avg_round_u8x4(A, avg_round_u8x4(A, w2));
                    } else {
                        const uint32_t al = A & 0x00ff00ffu, ah = (A >> 8) & 0x00ff00ffu;
                        hl[s][c][0] = mad24(al, 3u, wm & 0x00ff00ffu); hh[s][c][0] = mad24(ah, 3u, (wm >> 8) & 0x00ff00ffu);
                        hl[s][c][1] = mad24(al, 3u, w2 & 0x00ff00ffu); hh[s][c][1] = mad24(ah, 3u, (w2 >> 8) & 0x00ff00ffu);
                    }
                    continue;
                }
                constexpr uint32_t kT = 0x0020c020u;   // taps (1, 6, 1, 0) x 32; accumulator 4 x 32
                const uint32_t t0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(A, prev, 3), kT, 128u, false);
                const uint32_t t1 = __builtin_amdgcn_udot4(A, kT, 128u, false);
                const uint32_t t2 = __builtin_amdgcn_udot4(w2, kT, 128u, false);
                const uint32_t t3 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(next, A, 2), kT, 128u, false);
                const uint32_t ev = __builtin_amdgcn_perm(__builtin_amdgcn_perm(t3, t2, 0x0c0c0501u), __builtin_amdgcn_perm(t1, t0, 0x0c0c0501u), 0x05040100u);
                const uint32_t od = avg_round_u8x4(A, w2);
                hp_[s][c][0] = ev; hp_[s][c][1] = od;
                hl[s][c][0] = ev & 0x00ff00ffu; hh[s][c][0] = (ev >> 8) & 0x00ff00ffu;
                hl[s][c][1] = od & 0x00ff00ffu; hh[s][c][1] = (od >> 8) & 0x00ff00ffu;
            }
            if (i >= 2 && i < n) {   // rows y - 1, y, y + 1 are in: destination rows 2 y and 2 y + 1 (y = y0 + i - 2)
                const int sp = (s + 1) % 3, sc = (s + 2) % 3, sn = s;   // compile-time after unrolling
                [[maybe_unused]] const int yy = y0 + i - 2;   // the source row these two destination rows are centred on
                uint32_t w[2][4];   // [destination row][dword]
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    uint32_t ve[2], vo[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if constexpr (AR == kUpNearest) {
                            ve[e] = hp_[sc][c][e]; vo[e] = hp_[sc][c][e];
                            continue;
                        } else if constexpr (AR == kUpRh) {
                            ve[e] = avg_round_u8x4(hp_[sc][c][e], avg_round_u8x4(hp_[sp][c][e], hp_[sc][c][e]));
                            vo[e] = avg_round_u8x4(hp_[sc][c][e], avg_round_u8x4(hp_[sc][c][e], hp_[sn][c][e]));
                            continue;
                        } else if constexpr (AR == kUpQ14) {
                            const uint32_t cl3 = hl[sc][c][e] + (hl[sc][c][e] << 1), ch3 = hh[sc][c][e] + (hh[sc][c][e] << 1);   // (10-bit lanes: beyond mad24's 24-bit operands)
                            const uint32_t el = ((cl3 + hl[sp][c][e] + 0x00080008u) >> 4) & 0x00ff00ffu, eh = ((ch3 + hh[sp][c][e] + 0x00080008u) >> 4) & 0x00ff00ffu;
                            const uint32_t ol = ((cl3 + hl[sn][c][e] + 0x00080008u) >> 4) & 0x00ff00ffu, oh = ((ch3 + hh[sn][c][e] + 0x00080008u) >> 4) & 0x00ff00ffu;
                            ve[e] = el | (eh << 8); vo[e] = ol | (oh << 8);
                            continue;
                        } else if constexpr (AR == kUpCv) {
                            constexpr uint32_t kS = 0x3fff3fffu, kB = 0x00ff00ffu, kR = 0x00020002u;   // per-lane >> 2, byte mask, rounding
                            const uint32_t cl = hl[sc][c][e], ch = hh[sc][c][e];
                            const uint32_t nl = ((cl + (cl << 1)) >> 2) & kS, nh = ((ch + (ch << 1)) >> 2) & kS;   // (3 h_near) >> 2
                            const bool first = yy == 0, last = yy == a.sh - 1;   // wave-uniform
                            const uint32_t el = first ? cl + kR : ((hl[sp][c][e] >> 2) & kS) + nl + kR, eh = first ? ch + kR : ((hh[sp][c][e] >> 2) & kS) + nh + kR;
                            const uint32_t ol = last ? cl + kR : ((hl[sn][c][e] >> 2) & kS) + nl + kR, oh = last ? ch + kR : ((hh[sn][c][e] >> 2) & kS) + nh + kR;
                            ve[e] = ((el >> 2) & kB) | (((eh >> 2) & kB) << 8); vo[e] = ((ol >> 2) & kB) | (((oh >> 2) & kB) << 8);
                            continue;
                        }
                        const uint32_t lo = ((mad24(hl[sc][c][e], 6u, hl[sp][c][e]) + hl[sn][c][e] + 0x00040004u) >> 3) & 0x00ff00ffu;
                        const uint32_t hi = ((mad24(hh[sc][c][e], 6u, hh[sp][c][e]) + hh[sn][c][e] + 0x00040004u) >> 3) & 0x00ff00ffu;
                        ve[e] = lo | (hi << 8);
                        vo[e] = avg_round_u8x4(hp_[sc][c][e], hp_[sn][c][e]);
                    }
                    w[0][2 * c] = __builtin_amdgcn_perm(ve[1], ve[0], 0x05010400u); w[0][2 * c + 1] = __builtin_amdgcn_perm(ve[1], ve[0], 0x07030602u);   // e0 o0 e1 o1 | e2 o2 e3 o3
                    w[1][2 * c] = __builtin_amdgcn_perm(vo[1], vo[0], 0x05010400u); w[1][2 * c + 1] = __builtin_amdgcn_perm(vo[1], vo[0], 0x07030602u);
                }
                if constexpr (RAGGED) {
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int off = row_off + r * a.dw;
                        if (nvalid == 8 && a.plain != 2) row_store<4>(out_win, off, w[r], a.plain);
                        else if (nvalid == 8) *reinterpret_cast<u32x4_unaligned*>(dst + off) = u32x4_t{w[r][0], w[r][1], w[r][2], w[r][3]};
                        else if (nvalid > 0) store_head_bytes(dst + off, w[r], 2 * nvalid);
                    }
                } else if (inside) {
                    row_store<4>(out_win, row_off, w[0], a.plain);
                    row_store<4>(out_win, row_off + a.dw, w[1], a.plain);
                }
                row_off += 2 * a.dw;
            }
        }
    }
}

// pyrup_u8 (:656-840): horizontal pass to a u8 intermediate, then the same taps vertically
template <int C>
__device__ __forceinline__ uint32_t pyrup_h_u8(const uint8_t* __restrict__ row, int sw, int X, int c) {
    const int x = X >> 1;
    const uint32_t pc = row[x * C + c], pn = row[reflect_101(x + 1, sw) * C + c];
    if (X & 1) return (pc + pn + 1u) >> 1;
    const uint32_t pp = row[reflect_101(x - 1, sw) * C + c];
    return ((pp + 6u * pc + pn + 4u) >> 3) & 0xffu;
}
template <int C>
__global__ __launch_bounds__(kBx* kBy) void pyrup_u8_kernel(Pyr<uint8_t> a) {
    KH_PYR_PROLOGUE(uint8_t)
    const int y = Y >> 1;
    const long long stride = (long long)a.sw * C;
    const uint8_t* rc = src + y * stride;
    const uint8_t* rn = src + reflect_101(y + 1, a.sh) * stride;
    const uint8_t* rp = src + reflect_101(y - 1, a.sh) * stride;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const uint32_t pc = pyrup_h_u8<C>(rc, a.sw, X, c), pn = pyrup_h_u8<C>(rn, a.sw, X, c);
        uint32_t v;
        if (Y & 1) v = (pc + pn + 1u) >> 1;
        else v = (pyrup_h_u8<C>(rp, a.sw, X, c) + 6u * pc + pn + 4u) >> 3;
        o[c] = (uint8_t)v;
    }
}

// pyrup_u8 by SOURCE pixel pair (round 2).  The per-pixel kernel above costs ~380 lane-operations per destination pixel (three
// horizontal passes per channel, reflections, divergent odd / even lanes): 20.6 ms per 256 1080p -> 4K RGB images, 0.05 of the
// roofline (r02zg).  Here a thread owns two neighbouring source pixels of one row and writes their 4 x 2 destination block: the
// 4-pixel windows of rows y - 1, y, y + 1 (reflect-101) come in as C unaligned dwords each (border threads assemble the same
// dwords from reflected bytes), v_perm_b32 puts the prev / curr / next samples of both pixels into 16-bit lanes, both passes are
// packed 16-bit arithmetic (<= 2044), and each destination row leaves as C dwords.  Same integer expressions: identical bytes.
template <int C>
__global__ __launch_bounds__(kBx* kBy) void pyrup_u8_pair_kernel(Pyr<uint8_t> a) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int x0 = 2 * (bx_ * kBx + threadIdx.x), y = by_ * kBy + threadIdx.y;  // SOURCE pixels x0, x0 + 1 of row y
    if (x0 >= a.sw || y >= a.sh) return;
    const uint8_t* __restrict__ src = a.src + (long long)bz_ * a.ss;
    uint8_t* __restrict__ dst = a.dst + (long long)bz_ * a.ds;
    const int rows[3] = {reflect_101(y - 1, a.sh), y, reflect_101(y + 1, a.sh)};
    const long long stride = (long long)a.sw * C;
    uint32_t w[3][C];  // per row: the bytes of pixels x0 - 1, x0, x0 + 1, x0 + 2
    if (x0 >= 1 && x0 + 2 < a.sw) {  // the whole window is inside the row (wave-uniform except in the edge columns)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint8_t* q = src + rows[r] * stride + (x0 - 1) * C;
#pragma unroll
            for (int j = 0; j < C; ++j) w[r][j] = *reinterpret_cast<const u32_unaligned*>(q + 4 * j);
        }
    } else {
        int sx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) sx[t] = reflect_101(min(x0 - 1 + t, a.sw), a.sw) * C;  // x0 + 1 == sw (odd widths): a dead lane
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint8_t* row = src + rows[r] * stride;
#pragma unroll
            for (int j = 0; j < C; ++j) w[r][j] = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int c = 0; c < C; ++c) w[r][(t * C + c) >> 2] |= (uint32_t)row[sx[t] + c] << (8 * ((t * C + c) & 3));
        }
    }
    const u16x2_t one = {1, 1}, three = {3, 3}, four = {4, 4}, six = {6, 6}, ff = {255, 255};
    u16x2_t he[3][C], ho[3][C];  // lanes: source pixel x0, source pixel x0 + 1
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            u16x2_t t[3];  // prev, curr, next of both pixels
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n1 = j * C + c, n2 = (j + 1) * C + c;
                t[j] = as_u16x2(__builtin_amdgcn_perm(w[r][n2 >> 2], w[r][n1 >> 2], 0x0c000c00u | (uint32_t)(n1 & 3) | ((uint32_t)(4 + (n2 & 3)) << 16)));
            }
            he[r][c] = ((t[0] + t[1] * six + t[2] + four) >> three) & ff;
            ho[r][c] = (t[1] + t[2] + one) >> one;
        }
    const bool second = x0 + 1 < a.sw;
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // destination rows 2y (k = 0) and 2y + 1
        uint32_t o[4][C];           // [destination pixel 2 x0 + i][channel]
#pragma unroll
        for (int c = 0; c < C; ++c) {
            u16x2_t ve, vo;
            if (k == 0) {
                ve = ((he[0][c] + he[1][c] * six + he[2][c] + four) >> three) & ff;
                vo = ((ho[0][c] + ho[1][c] * six + ho[2][c] + four) >> three) & ff;
            } else {
                ve = (he[1][c] + he[2][c] + one) >> one;
                vo = (ho[1][c] + ho[2][c] + one) >> one;
            }
            o[0][c] = ve[0]; o[1][c] = vo[0]; o[2][c] = ve[1]; o[3][c] = vo[1];
        }
        uint8_t* op = dst + ((long long)(2 * y + k) * a.dw + 2 * x0) * C;
        if (second) {
            uint32_t d[C];
#pragma unroll
            for (int j = 0; j < C; ++j) d[j] = 0;
#pragma unroll
            for (int n = 0; n < 4 * C; ++n) d[n >> 2] |= o[n / C][n % C] << (8 * (n & 3));
#pragma unroll
            for (int j = 0; j < C; ++j) reinterpret_cast<u32_unaligned*>(op)[j] = d[j];
        } else {
#pragma unroll
            for (int n = 0; n < 2 * C; ++n) op[n] = (uint8_t)o[n / C][n % C];
        }
    }
}

// ---- morphology ------------------------------------------------------------------------------------------
struct Morph {
    const uint8_t* src;
    uint8_t* dst;
    int w, h, kw, kh, border, op;
    long long ss, ds;
    uint32_t rows[32];  // bit kx of rows[ky] = tap (ky, kx) active
    uint32_t cval[4];
    XcdTiles tiles;
};

__device__ __forceinline__ int map_index(int mode, int i, int len) {  // PaddingMode::map_index, padding.rs:32-80
    if (i >= 0 && i < len) return i;
    switch (mode) {
        case 1: return i < 0 ? 0 : len - 1;
        case 2:
            if (len == 1) return 0;
            while (i < 0 || i >= len) i = i < 0 ? -i : 2 * len - i - 2;
            return i;
        case 3:
            if (len == 1) return 0;
            while (i < 0 || i >= len) i = i < 0 ? -i - 1 : 2 * len - i - 1;
            return i;
        case 4: return ((i % len) + len) % len;
        default: return -1;  // constant
    }
}


// ---- box dilate / erode for RGB8, rolling wave, planar in registers (round 3) --------------------------------------------------
// morphology_u8_tile_kernel below is vector-ALU-bound (r02zp: 74 % busy; ~19 instructions per byte for a 5 x 5 box): with interleaved
// RGB every (byte pair, tap) costs a v_perm_b32 to gather it and a packed max, in both passes, plus the LDS tile traffic and three
// barriers.  Here, as in the round-3 u8 blur and pyramids: a WAVE walks down a strip with K rows of loads in flight and no barrier;
// a lane owns four pixels (12 bytes), de-interleaves them into one dword per channel and takes its neighbours' by wave shifts.
// ALL 64 lanes store, so a wave's row segment is 768 bytes = whole 128-byte lines: with 62 storing lanes (744 bytes, the shape of
// the round-3 blur and pyramid kernels) every segment boundary splits a line between two waves, and a pure copy in that shape
// runs at 3.05 ms against 2.47 ms for this one (profiles/r03za: stores alone 1.59 vs 1.16 ms).  The pixels either side of the
// wave come from one extra load per row — the lower half's lanes all load the quad before the wave's first, the upper's the quad
// after its last — de-interleaved the same way and handed to the end lanes as the DPP wave shift's fill value.  The
// byte pair (b[i], b[i+2]) of a channel's twelve bytes (prev | cur | next) in 16-bit lanes is ONE v_perm_b32 of two neighbouring
// dwords (K + 1 of them per channel), and the K-wide row maxima of the four pixels are K + 1 packed max (the pairs (0, 2) and
// (1, 3) share all but one term).  Column pass: on pair maxima of consecutive rows, 1 + K / 2 packed max per register.  Re-interleave, one
// 12-byte store.  Borders: the row index through map_index (constant: the whole row is the border value); columns on edge waves by
// loading the quad from a clamped position and re-indexing it — and substituting the border value — with ONE v_perm_b32 per channel
// whose per-lane selector is computed once.  max / min are exact and order-independent: byte-identical to the other kernels.
// For square all-ones masks of 3 / 5 / 7, 3 channels, every border mode but wrap, images at least 4 pixels wide.
template <bool DILATE>
__device__ __forceinline__ uint32_t pk_minmax(uint32_t a, uint32_t b) {
    const u16x2_t x = __builtin_bit_cast(u16x2_t, a), y = __builtin_bit_cast(u16x2_t, b);
    return __builtin_bit_cast(uint32_t, DILATE ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y));
}
// Cross and "ellipse" structuring elements of 3 / 5 / 7 (round 6).  They took the LDS-tile kernel's mask scan at 2-6x the time of the
// box (profiles/r06zo_morph_shapes.txt).  Each of them is a union of at most two RECTANGLES of the K x K window — the cross its middle row
// and its middle column; the reference's ellipse (kernels.rs:163-190: centre (K / 2.0, K / 2.0), so the disc sits half a pixel down /
// right of the anchor) is the 2 x 2 block [1, 2]^2 for K = 3, the 4 x 4 block [1, 4]^2 for K = 5, and rows 2..5 x columns 1..6 joined
// with rows 1..6 x columns 2..5 for K = 7 — and max / min over a union is the max / min of the parts, each of which is separable.  The
// rolling kernels below keep, per rectangle, the last K rows' maxima over the rectangle's column run and combine the rows of its band.
enum { kMsBox = 0, kMsCross = 1, kMsEllipse = 2 };
struct MRect { int xl, xh, yt, yb; };   // columns xl..xh, rows yt..yb of the window (tap (ky, kx) reads pixel (y + ky - K / 2, x + kx - K / 2))
template <int K, int SHAPE> __host__ __device__ constexpr int morph_nrects() { return SHAPE == kMsCross ? 2 : (SHAPE == kMsEllipse && K == 7 ? 2 : 1); }
template <int K, int SHAPE> __host__ __device__ constexpr MRect morph_rect(int j) {
    if (SHAPE == kMsCross) return j == 0 ? MRect{0, K - 1, K / 2, K / 2} : MRect{K / 2, K / 2, 0, K - 1};
    if (SHAPE == kMsEllipse) return K == 3 ? MRect{1, 2, 1, 2} : K == 5 ? MRect{1, 4, 1, 4} : (j == 0 ? MRect{1, 6, 2, 5} : MRect{2, 5, 1, 6});
    return MRect{0, K - 1, 0, K - 1};
}
// One loaded row of one register column (four pixels of a plane in `P`'s twelve-byte string): the row's maxima over the rectangle's
// run for the pixel pairs (0, 2) and (1, 3) go into slot `s` of the rectangle's ring; returned are the maxima over its band of rows,
// the newest loaded row being the window's LAST.  `s` is a compile-time constant after unrolling.
template <int K, int XL, int XH, int YT, int YB, bool DILATE, class PF>
__device__ __forceinline__ void morph_rect_step(uint32_t (&hist)[K][2], int s, PF P, uint32_t& ve, uint32_t& vo) {
    constexpr int H = K / 2;
    uint32_t re, ro;
    if constexpr (XL == XH) { re = P(4 - H + XL); ro = P(5 - H + XL); }
    else {   // pixels (0, 2) take P(4 - H + XL .. 4 - H + XH), pixels (1, 3) the same range one up: all but one term in common
        uint32_t m = P(5 - H + XL);
#pragma unroll
        for (int i = 6 - H + XL; i <= 4 - H + XH; ++i) m = pk_minmax<DILATE>(m, P(i));
        re = pk_minmax<DILATE>(m, P(4 - H + XL)); ro = pk_minmax<DILATE>(m, P(5 - H + XH));
    }
    hist[s][0] = re; hist[s][1] = ro;
    constexpr int a0 = K - 1 - YB, a1 = K - 1 - YT;   // ages of the band's rows (0 = the row just loaded)
    ve = hist[(s + K - a0) % K][0]; vo = hist[(s + K - a0) % K][1];
#pragma unroll
    for (int d = a0 + 1; d <= a1; ++d) {
        ve = pk_minmax<DILATE>(ve, hist[(s + K - d) % K][0]);
        vo = pk_minmax<DILATE>(vo, hist[(s + K - d) % K][1]);
    }
}
template <int K, int SHAPE, bool DILATE, class PF>
__device__ __forceinline__ void morph_shape_step(uint32_t (&hist)[2][K][2], int s, PF P, uint32_t& ve, uint32_t& vo) {
    constexpr MRect r0 = morph_rect<K, SHAPE>(0);
    morph_rect_step<K, r0.xl, r0.xh, r0.yt, r0.yb, DILATE>(hist[0], s, P, ve, vo);
    if constexpr (morph_nrects<K, SHAPE>() == 2) {
        constexpr MRect r1 = morph_rect<K, SHAPE>(1);
        uint32_t ve1, vo1;
        morph_rect_step<K, r1.xl, r1.xh, r1.yt, r1.yb, DILATE>(hist[1], s, P, ve1, vo1);
        ve = pk_minmax<DILATE>(ve, ve1); vo = pk_minmax<DILATE>(vo, vo1);
    }
}
// register targets for the scheduler (blocks of 4 waves per CU = waves per SIMD): the box kernels 64 / 96 / 128 VGPRs; a cross or the
// two-rectangle ellipse keeps a second ring per register column and gets the next step down
__host__ __device__ constexpr int morph_roll_blocks(int K, int shape, bool ragged = false) {   // (ragged: twelve more registers of selectors)
    const int b = shape == kMsBox || (shape == kMsEllipse && K < 7) ? (K <= 3 ? 8 : (K <= 5 ? 5 : 4)) : (K <= 3 ? 6 : (K <= 5 ? 4 : 3));
    return !ragged ? b : (b >= 8 ? 6 : (b >= 3 ? b - 1 : 2));
}
struct MorphRoll {
    const uint8_t* src;
    uint8_t* dst;
    int w, h, th, border;
    long long ss, ds;
    uint32_t cval[4];
    XcdTiles tiles;
    int plain;                // write-back instead of streaming stores (kh_common.h::plain_row_stores)
};
constexpr int kMrWavePx = 256, kMrTilePx = 4 * kMrWavePx;

// C = 4 (round 6): RGBA / BGRA images took the LDS-tile kernel at 0.40 of peak.  The same kernel with a 16-byte quad per lane (one load and
// one store of whole pixels, 1 KiB per wave and row), de-interleaved into four channel dwords by a 4 x 4 byte transpose (eight v_perm_b32,
// and eight back) instead of RGB's six / nine; everything between is per channel and unchanged.
template <int K, bool DILATE, int SHAPE = kMsBox, int C = 3>
__global__ __launch_bounds__(256, morph_roll_blocks(K, SHAPE, C == 4)) void morph_u8_rgb_roll_kernel(MorphRoll a) {   // a register target for the scheduler: 5 x 5 105 -> <= 96, 7 x 7 137 -> <= 128 VGPRs
    constexpr int H = K / 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int p0 = (int)tx * kMrTilePx + wv * kMrWavePx;    // first output pixel of this wave
    if (p0 >= a.w) return;                                  // whole wave idle (no block barrier below)
    const int y0 = ty * a.th;
    const uint8_t* __restrict__ src = a.src + (long long)bz * a.ss;
    uint8_t* __restrict__ dst = a.dst + (long long)bz * a.ds;
    const bool stream_ok = ((a.w * C) & 3) == 0 && (long long)a.w * a.h * C <= 0x7fffffffLL && (C == 3 || a.plain != 2);   // block-uniform: streaming stores (kh_common.h); C = 4, plain = 2: a destination off a dword
    const __amdgpu_buffer_rsrc_t out_win = stream_window(dst, (long long)a.w * a.h * C);
    const int p = p0 + 4 * lane;                            // this lane's quad
    const int ph = lane < 32 ? p0 - 4 : p0 + kMrWavePx;     // the wave's halo quads: left in the lower half's lanes, right in the upper's
    const bool edge = p0 < 4 || p0 + kMrWavePx + 4 > a.w;   // wave-uniform
    const int pc = min(p, a.w - 4), phc = min(max(ph, 0), a.w - 4);
    uint32_t esel = 0x03020100u, hsel = 0x03020100u;   // byte j: 0..3 = loaded pixel, 4 = the constant border value
    if (edge) {
        esel = hsel = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = map_index(a.border, p + j, a.w), mh = map_index(a.border, ph + j, a.w);
            esel |= (uint32_t)(m < 0 ? 4 : min(max(m - pc, 0), 3)) << (8 * j);
            hsel |= (uint32_t)(mh < 0 ? 4 : min(max(mh - phc, 0), 3)) << (8 * j);
        }
    }
    const bool writer = p < a.w;
    const bool full = p + 3 < a.w;
    const int rowb = a.w * C;
    const int nrows = min(a.th, a.h - y0) + 2 * H;
    int pf_row = y0 - H;
    uint32_t cv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) cv[c] = a.cval[c] * 0x01010101u;

    uint32_t q[K][2 * C];   // the lane's quad and its half-wave's halo quad
    auto prefetch = [&](uint32_t (&d)[2 * C]) {
        const uint8_t* rp = src + (long long)max(map_index(a.border, pf_row, a.h), 0) * rowb;
        const uint8_t *rq = rp + C * pc, *rh = rp + C * phc;
        if constexpr (C == 4) {
            const u32x4_t v = *reinterpret_cast<const u32x4_unaligned*>(rq), hq = *reinterpret_cast<const u32x4_unaligned*>(rh);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; d[4] = hq.x; d[5] = hq.y; d[6] = hq.z; d[7] = hq.w;
        } else {
            d[0] = *reinterpret_cast<const u32_unaligned*>(rq); d[1] = *reinterpret_cast<const u32_unaligned*>(rq + 4); d[2] = *reinterpret_cast<const u32_unaligned*>(rq + 8);
            d[3] = *reinterpret_cast<const u32_unaligned*>(rh); d[4] = *reinterpret_cast<const u32_unaligned*>(rh + 4); d[5] = *reinterpret_cast<const u32_unaligned*>(rh + 8);
        }
        ++pf_row;
    };
#pragma unroll
    for (int i = 0; i < K; ++i) prefetch(q[i]);

    constexpr uint32_t kInit = DILATE ? 0u : 0x00ff00ffu;
    // column pass on pair maxima: pr[t] = max(row t - 1, row t), so the K-row maximum ending at row t is row t with pr[t - 1], pr[t - 3] ...:
    // 1 + K / 2 packed max per register instead of K - 1
    uint32_t pr[K][C][2], last[C][2];
    uint32_t hist[C][2][K][2];   // (cross / ellipse: per channel and rectangle, the last K rows' run maxima)
#pragma unroll
    for (int c = 0; c < C; ++c) {
        last[c][0] = kInit; last[c][1] = kInit;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            pr[i][c][0] = kInit; pr[i][c][1] = kInit;
            hist[c][0][i][0] = kInit; hist[c][0][i][1] = kInit; hist[c][1][i][0] = kInit; hist[c][1][i][1] = kInit;
        }
    }

    long long out_off = (long long)(y0 - 2 * H) * rowb + C * p;
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const int r = rb + s, row = y0 - H + r;
            const bool row_out = a.border == KH_BORDER_CONSTANT && (row < 0 || row >= a.h);   // wave-uniform: the whole row is the border value
            uint32_t dq[2 * C];
#pragma unroll
            for (int k = 0; k < 2 * C; ++k) dq[k] = q[s][k];
            prefetch(q[s]);
            uint32_t curs[C], halos[C], pl[C];
            deinterleave_quad<C>(dq, curs);
            deinterleave_quad<C>(dq + C, halos);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                uint32_t cur = curs[c], halo = halos[c];
                if (edge) { cur = __builtin_amdgcn_perm(cv[c], cur, esel); halo = __builtin_amdgcn_perm(cv[c], halo, hsel); }
                if (row_out) { cur = cv[c]; halo = cv[c]; }
                const uint32_t prev = from_lane_below(cur, halo), next = from_lane_above(cur, halo);
                // P(i) = (b[i], b[i + 2]) of the 12-byte string prev | cur | next in 16-bit lanes: one v_perm_b32 of two neighbouring dwords
                auto P = [&](int i) -> uint32_t {   // i is a compile-time constant after unrolling
                    return i <= 5 ? __builtin_amdgcn_perm(cur, prev, 0x0c000c00u | (uint32_t)i | ((uint32_t)(i + 2) << 16))
                                  : __builtin_amdgcn_perm(next, cur, 0x0c000c00u | (uint32_t)(i - 4) | ((uint32_t)(i - 2) << 16));
                };
                uint32_t ve, vo;
                if constexpr (SHAPE != kMsBox) morph_shape_step<K, SHAPE, DILATE>(hist[c], s, P, ve, vo);
                else {
                    // pixels (0, 2) take P(4 - H .. 4 + H), pixels (1, 3) P(5 - H .. 5 + H): K - 1 terms in common
                    uint32_t m = P(5 - H);
#pragma unroll
                    for (int i = 6 - H; i <= 4 + H; ++i) m = pk_minmax<DILATE>(m, P(i));
                    const uint32_t re = pk_minmax<DILATE>(m, P(4 - H)), ro = pk_minmax<DILATE>(m, P(5 + H));
                    ve = re; vo = ro;
#pragma unroll
                    for (int j = 1; j <= H; ++j) {
                        ve = pk_minmax<DILATE>(ve, pr[(s + K - (2 * j - 1)) % K][c][0]);
                        vo = pk_minmax<DILATE>(vo, pr[(s + K - (2 * j - 1)) % K][c][1]);
                    }
                    pr[s][c][0] = pk_minmax<DILATE>(re, last[c][0]); pr[s][c][1] = pk_minmax<DILATE>(ro, last[c][1]);
                    last[c][0] = re; last[c][1] = ro;
                }
                pl[c] = __builtin_amdgcn_perm(vo, ve, 0x06020400u);   // pixels 0, 1, 2, 3 of this channel
            }
            if (writer && r >= 2 * H && r < nrows) {
                uint32_t w[C];
                interleave_quad<C>(pl, w);
                uint8_t* o = dst + out_off;
                if (full && stream_ok) {
                    row_store<C>(out_win, (int)out_off, w, a.plain);
                } else if (full) {
#pragma unroll
                    for (int k = 0; k < C; ++k) *reinterpret_cast<u32_unaligned*>(o + 4 * k) = w[k];
                } else {
#pragma unroll
                    for (int b = 0; b < 3 * C; ++b)   // at most three pixels of a quad that reaches past the last column
                        if (p + b / C < a.w) o[b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
                }
            }
            out_off += rowb;
        }
    }
}

// ---- box dilate / erode for single-channel u8, rolling wave (round 6) -------------------------------------------------------------------
// Masks and gray images are what morphology mostly runs on, and they took the LDS-tile kernel at 0.23-0.31 of peak against 0.57-0.66 for
// the planar RGB kernel above.  The same walk on one plane: a lane owns SIXTEEN pixels of a row (one 16-byte load and store, 1 KiB per
// wave and row), its four dwords go through exactly the per-channel code of the RGB kernel — the twelve-byte string (previous dword |
// this dword | next dword), byte pairs in 16-bit lanes, K + 1 packed max / min per dword, the pair-maxima column pass — with the dword
// before the lane's first and after its last taken from the neighbouring lanes by wave shifts and, at the ends of the wave, from one
// halo dword per half-wave.  Borders as in the RGB kernel: rows through map_index (constant: the whole row is the border value); the
// three pixels beyond a row end that a 7-tap window can reach are re-indexed (or replaced by the border value) with one v_perm_b32
// whose per-lane selector is computed once.  For square all-ones masks of 3 / 5 / 7, widths that are multiples of 16, every border mode
// but wrap; byte-identical to the other kernels (max / min are exact and order-independent).
constexpr int kMgWavePx = 1024, kMgTilePx = 4 * kMgWavePx;
// RAGGED (round 6): any width >= 16 and any alignment (remap16_* above); plain = 2: rows or images off a dword, unaligned global stores.
template <int K, bool DILATE, int SHAPE = kMsBox, bool RAGGED = false>
__global__ __launch_bounds__(256, morph_roll_blocks(K, SHAPE, RAGGED)) void morph_u8_gray_roll_kernel(MorphRoll a) {
    constexpr int H = K / 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int p0 = (int)tx * kMgTilePx + wv * kMgWavePx;    // first output pixel of this wave
    if (p0 >= a.w) return;                                  // whole wave idle (no block barrier below)
    const int y0 = ty * a.th;
    const uint8_t* __restrict__ src = a.src + (long long)bz * a.ss;
    uint8_t* __restrict__ dst = a.dst + (long long)bz * a.ds;
    const __amdgpu_buffer_rsrc_t out_win = stream_window(dst, (long long)a.w * a.h);   // (w * h < 2^31: host-checked)
    const int p = p0 + 16 * lane;                           // this lane's sixteen pixels
    const int nvalid = min(max(a.w - p, 0), 16);            // RAGGED: 0 .. 16; otherwise all sixteen or none (w % 16 == 0: host-checked)
    const bool inside = nvalid > 0;
    const int ph = lane < 32 ? p0 - 4 : p0 + kMgWavePx;     // the wave's halo dwords: left in the lower half's lanes, right in the upper's
    const bool edge = p0 < 4 || p0 + kMgWavePx + 4 > a.w;   // wave-uniform
    const int pc = min(p, a.w - 16), phc = min(max(ph, 0), a.w - 4);
    // a lane past the row end supplies the pixels its inside neighbour's window reaches: its first dword <- the row's last dword re-indexed
    uint32_t esel = 0x03020100u, hsel = 0x03020100u;   // byte j: 0..3 = loaded pixel, 4 = the constant border value
    if (edge) {
        esel = hsel = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = map_index(a.border, p + j, a.w), mh = map_index(a.border, ph + j, a.w);
            esel |= (uint32_t)(m < 0 ? 4 : min(max(m - (a.w - 4), 0), 3)) << (8 * j);
            hsel |= (uint32_t)(mh < 0 ? 4 : min(max(mh - phc, 0), 3)) << (8 * j);
        }
    }
    Remap16 rm{};
    if (RAGGED && edge) rm = remap16_setup(p, pc, [&](int x) { return map_index(a.border, x, a.w); });
    const int nrows = min(a.th, a.h - y0) + 2 * H;
    int pf_row = y0 - H;
    const uint32_t cv = a.cval[0] * 0x01010101u;

    uint32_t q[K][5];   // the lane's sixteen pixels and its half-wave's halo dword
    auto prefetch = [&](uint32_t (&d)[5]) {
        const uint8_t* rp = src + (long long)max(map_index(a.border, pf_row, a.h), 0) * a.w;
        const u32x4_t v = *reinterpret_cast<const u32x4_unaligned*>(rp + pc);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        d[4] = *reinterpret_cast<const u32_unaligned*>(rp + phc);
        ++pf_row;
    };
#pragma unroll
    for (int i = 0; i < K; ++i) prefetch(q[i]);

    constexpr uint32_t kInit = DILATE ? 0u : 0x00ff00ffu;
    uint32_t pr[K][4][2], last[4][2];   // pair maxima of consecutive rows, as in the RGB kernel
    uint32_t hist[4][2][K][2];          // (cross / ellipse: per dword and rectangle, the last K rows' run maxima)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        last[c][0] = kInit; last[c][1] = kInit;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            pr[i][c][0] = kInit; pr[i][c][1] = kInit;
            hist[c][0][i][0] = kInit; hist[c][0][i][1] = kInit; hist[c][1][i][0] = kInit; hist[c][1][i][1] = kInit;
        }
    }

    int out_off = (y0 - 2 * H) * a.w + p;
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const int r = rb + s, row = y0 - H + r;
            const bool row_out = a.border == KH_BORDER_CONSTANT && (row < 0 || row >= a.h);   // wave-uniform: the whole row is the border value
            uint32_t cur[4] = {q[s][0], q[s][1], q[s][2], q[s][3]}, halo = q[s][4];
            prefetch(q[s]);
            if (edge) {
                if constexpr (RAGGED) remap16_apply(rm, cur, cv);
                else {
                    const uint32_t beyond = __builtin_amdgcn_perm(cv, cur[3], esel);   // (of a lane past the row end: the loaded sixteen are the row's last)
                    cur[0] = inside ? cur[0] : beyond;
                }
                halo = __builtin_amdgcn_perm(cv, halo, hsel);
            }
            if (row_out) { cur[0] = cv; cur[1] = cv; cur[2] = cv; cur[3] = cv; halo = cv; }
            const uint32_t prevd = from_lane_below(cur[3], halo), nextd = from_lane_above(cur[0], halo);
            uint32_t pl[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t prev = c == 0 ? prevd : cur[c - 1], next = c == 3 ? nextd : cur[c + 1], mid = cur[c];
                auto P = [&](int i) -> uint32_t {   // (b[i], b[i + 2]) of prev | mid | next in 16-bit lanes; i is a compile-time constant after unrolling
                    return i <= 5 ? __builtin_amdgcn_perm(mid, prev, 0x0c000c00u | (uint32_t)i | ((uint32_t)(i + 2) << 16))
                                  : __builtin_amdgcn_perm(next, mid, 0x0c000c00u | (uint32_t)(i - 4) | ((uint32_t)(i - 2) << 16));
                };
                uint32_t ve, vo;
                if constexpr (SHAPE != kMsBox) morph_shape_step<K, SHAPE, DILATE>(hist[c], s, P, ve, vo);
                else {
                    uint32_t m = P(5 - H);
#pragma unroll
                    for (int i = 6 - H; i <= 4 + H; ++i) m = pk_minmax<DILATE>(m, P(i));
                    const uint32_t re = pk_minmax<DILATE>(m, P(4 - H)), ro = pk_minmax<DILATE>(m, P(5 + H));
                    ve = re; vo = ro;
#pragma unroll
                    for (int j = 1; j <= H; ++j) {
                        ve = pk_minmax<DILATE>(ve, pr[(s + K - (2 * j - 1)) % K][c][0]);
                        vo = pk_minmax<DILATE>(vo, pr[(s + K - (2 * j - 1)) % K][c][1]);
                    }
                    pr[s][c][0] = pk_minmax<DILATE>(re, last[c][0]); pr[s][c][1] = pk_minmax<DILATE>(ro, last[c][1]);
                    last[c][0] = re; last[c][1] = ro;
                }
                pl[c] = __builtin_amdgcn_perm(vo, ve, 0x06020400u);   // pixels 0, 1, 2, 3 of this dword
            }
            if constexpr (RAGGED) {
                if (r >= 2 * H && r < nrows) {
                    if (nvalid == 16 && a.plain != 2) row_store<4>(out_win, out_off, pl, a.plain);
                    else if (nvalid == 16) *reinterpret_cast<u32x4_unaligned*>(dst + out_off) = u32x4_t{pl[0], pl[1], pl[2], pl[3]};
                    else if (nvalid > 0) store_head_bytes(dst + out_off, pl, nvalid);
                }
            } else if (inside && r >= 2 * H && r < nrows) row_store<4>(out_win, out_off, pl, a.plain);
            out_off += a.w;
        }
    }
}

template <int C>
__global__ __launch_bounds__(kBx* kBy) void morphology_u8_kernel(Morph a) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int x = bx_ * kBx + threadIdx.x, y = by_ * kBy + threadIdx.y;
    if (x >= a.w || y >= a.h) return;
    const uint8_t* __restrict__ src = a.src + (long long)bz_ * a.ss;
    const int pad_h = a.kh / 2, pad_w = a.kw / 2;
    const bool dilate = a.op == 0;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = dilate ? 0 : 256;  // dilate starts from T::default(), ops.rs:88
    bool any = false;
    for (int ky = 0; ky < a.kh; ++ky) {
        const uint32_t bits = a.rows[ky];
        if (!bits) continue;
        const int sy = map_index(a.border, y + ky - pad_h, a.h);
        for (int kx = 0; kx < a.kw; ++kx) {
            if (!((bits >> kx) & 1u)) continue;
            any = true;
            const int sx = map_index(a.border, x + kx - pad_w, a.w);
            const bool outside = sy < 0 || sx < 0;
            const uint8_t* p = src + ((long long)max(sy, 0) * a.w + max(sx, 0)) * C;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int v = outside ? (int)a.cval[c] : (int)p[c];
                acc[c] = dilate ? max(acc[c], v) : min(acc[c], v);
            }
        }
    }
    uint8_t* o = a.dst + (long long)bz_ * a.ds + ((long long)y * a.w + x) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = (uint8_t)(any ? acc[c] : 0);  // erode with no active tap: unwrap_or_default
}


// ---- tiled morphology (round 2) -------------------------------------------------------------------------------------------------
// The per-pixel kernel above issues kw * kh * C byte loads and two border mappings per tap: 69.9 ms per 256 4K RGB images for a
// 5x5 box (0.02 of the HBM roofline).  Here a 256-thread block owns an output tile of kMorphFW flat bytes x kMorphTH rows and
//   1. stages the source window (tile + (kw - 1) x (kh - 1) halo) in LDS as flat byte rows; the border mode is resolved ONCE per
//      staged pixel (constant mode writes the border value), interior tiles copy dwords;
//   2. for an all-ones (box) structuring element: a horizontal pass S -> H (max / min over kw taps) and a vertical pass H ->
//      destination: kw + kh taps per byte instead of kw * kh; for any other mask: the active taps straight from S.
// The kernel is bound by vector-ALU issue, not memory (r02w: the first version spent ~190 lane-operations per four output bytes
// and ran at exactly that rate), so taps work on two bytes per instruction: v_perm_b32 pulls the even / odd bytes of a tap's
// unaligned four-byte window out of two LDS dwords straight into 16-bit lanes (the selector depends on the tap only, not on the
// lane) and v_pk_max_u16 / v_pk_min_u16 accumulates them; H holds the lanes unpacked, so a vertical tap is one ds_read_b64 and
// two packed max.  max / min are order-independent and exact: the bytes equal the per-pixel kernel's.
constexpr int kMorphFW = 384;   // flat bytes per tile row: a whole number of pixels for C = 1, 2, 3, 4, and of dwords
constexpr int kMorphTH = 32;    // output rows per tile
extern __shared__ __attribute__((aligned(16))) uint8_t kh_morph_lds[];

// one tap on the four output bytes whose window starts `o` bytes into LDS row `row32` (o & 3 is the same for every lane)
template <bool DILATE>
__device__ __forceinline__ void tap_pair(const uint32_t* row32, int o, uint32_t& acc_e, uint32_t& acc_o) {
    const uint32_t lo = row32[o >> 2], hi = row32[(o >> 2) + 1], s = (uint32_t)(o & 3);
    const uint32_t sel_e = 0x0c000c00u | s | ((s + 2u) << 16);   // {byte s, 0, byte s + 2, 0}
    acc_e = pk_minmax<DILATE>(acc_e, __builtin_amdgcn_perm(hi, lo, sel_e));
    acc_o = pk_minmax<DILATE>(acc_o, __builtin_amdgcn_perm(hi, lo, sel_e + 0x00010001u));
}

// K > 0: a K x K box known at compile time (3, 5, 7): the tap loops unroll, a window's dwords are read once and the tile
// geometry folds into constants (r02zb: the run-time loops cost 128 lane-operations per four output bytes, 62 % VALU-busy).
template <int C, bool DILATE, bool BOX, int K>
__global__ __launch_bounds__(256) void morphology_u8_tile_kernel(Morph a, int sp_, int srows_) {
    static_assert(K == 0 || BOX, "a compile-time size is a box");
    const int srows = K ? kMorphTH + K - 1 : srows_;
    const int sp = K ? ((kMorphFW + (K - 1) * C + 3) & ~3) + 4 : sp_;
    if (K) { a.kw = K; a.kh = K; }
    uint8_t* S = kh_morph_lds;                                              // [srows][sp]: source window, flat bytes
    uint32_t* H = reinterpret_cast<uint32_t*>(kh_morph_lds + srows * sp);   // [srows][kDW][2]: horizontal pass, 16-bit lanes (BOX only)
    unsigned bx_, by_, bz_;
    if (!xcd_tile(a.tiles, bx_, by_, bz_)) return;
    const int tid = threadIdx.x;
    const int pad_h = a.kh / 2, pad_w = a.kw / 2;
    const int x0 = bx_ * (kMorphFW / C), y0 = by_ * kMorphTH;            // first output pixel / row of the tile
    const uint8_t* __restrict__ src = a.src + (long long)bz_ * a.ss;
    uint8_t* __restrict__ dst = a.dst + (long long)bz_ * a.ds;
    const int spx = kMorphFW / C + a.kw - 1;                                // staged pixels per row
    const int wx0 = x0 - pad_w, wy0 = y0 - pad_h;                           // source coordinates of staged cell (0, 0)

    // 1. stage the window
    const bool interior = wx0 >= 0 && wy0 >= 0 && wx0 + spx <= a.w && wy0 + srows <= a.h;   // block-uniform
    // Both staging loops issue a batch of independent global loads before the first LDS write: one load per loop trip left the
    // block waiting a full memory round trip per trip (r02z: 18 trips, 12 ms per 256 4K images, slower than the VALU work).
    if (interior) {
        const int dpr = (spx * C + 3) >> 2;                                  // dwords per staged row (sp >= 4 * dpr, dpr <= 128)
        // the last dword of a row may read up to 3 bytes past the window: still inside the image except at its very end
        const long long img_bytes = (long long)a.w * a.h * C;
        // Branch-free: every thread loads from a clamped (row, dword) of the window — a load inside its own branch is waited for
        // before the next one issues — and the duplicate lanes / rows store the same value to the same LDS cell.  The last dword
        // of the image is fetched from img_bytes - 4 and shifted down.
        const int d = min(tid & 127, dpr - 1);
        constexpr int kBatch = 10;                                           // rows per thread in flight (the two 128-thread halves interleave rows)
        // Addresses: ONE 64-bit window base per block (wave-uniform, scalar registers) plus a 32-bit offset per load — the offset of a
        // staged cell from the window's first byte is < 64 rows x 2^24 bytes.  Round 2 evaluated `((wy0 + r) * w + wx0) * C` in 64 bits
        // per load: three quarter-rate v_mad_u64_u32 each, about a quarter of the kernel's vector-ALU time (r03 ISA review).
        const uint8_t* wbase = src + ((long long)wy0 * a.w + wx0) * C;
        const uint32_t rowb = (uint32_t)(a.w * C);                           // < 2^24 for any image this kernel takes (host-checked)
        const long long lim64 = img_bytes - 4 - ((long long)wy0 * a.w + wx0) * C;
        const uint32_t lim = (uint32_t)(lim64 > 0x7fffffffLL ? 0x7fffffffLL : lim64);   // last in-image dword, relative to the window (>= 0)
        for (int r0 = tid >> 7; r0 < srows; r0 += 2 * kBatch) {
            uint32_t v[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int r = min(r0 + 2 * k, srows - 1);
                const uint32_t off = __umul24((uint32_t)r, rowb) + 4u * (uint32_t)d, offc = min(off, lim);
                v[k] = *reinterpret_cast<const u32_unaligned*>(wbase + offc) >> (8 * (int)(off - offc));
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) *reinterpret_cast<uint32_t*>(S + __umul24((uint32_t)min(r0 + 2 * k, srows - 1), (uint32_t)sp) + 4 * d) = v[k];
        }
    } else {
        constexpr int kBatch = 4;
        const int n = spx * srows;
        for (int i0 = tid; i0 < n; i0 += 256 * kBatch) {
            uint8_t px[kBatch][C];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int i = i0 + 256 * k;
                const int r = i / spx, c = i - r * spx;
                // cells right of / below every output's window (ragged last tiles) are never consumed: skip their border walk
                const bool unused = i >= n || wx0 + c >= a.w + pad_w || wy0 + r >= a.h + pad_h;
                const int sy = unused ? -1 : map_index(a.border, wy0 + r, a.h), sx = unused ? -1 : map_index(a.border, wx0 + c, a.w);
                const bool outside = sy < 0 || sx < 0;
                const uint8_t* p = src + ((long long)max(sy, 0) * a.w + max(sx, 0)) * C;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) px[k][ch] = outside ? (uint8_t)a.cval[ch] : p[ch];
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int i = i0 + 256 * k;
                if (i >= n) break;
                const int r = i / spx, c = i - r * spx;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) S[r * sp + c * C + ch] = px[k][ch];
            }
        }
    }
    __syncthreads();

    constexpr int kDW = kMorphFW / 4;   // four-byte items per row
    const uint32_t init = DILATE ? 0u : 0x00ff00ffu;
    if constexpr (BOX) {
        // 2a. horizontal pass over every staged row
        if constexpr (K > 0) {
            // eight output bytes per item: the two windows share all but one of their dwords and the index arithmetic
            static_assert(kDW / 2 == 48, "the multiply-shift below divides by 48");
            for (int i = tid; i < (kDW / 2) * srows; i += 256) {
                const int r = (int)(__umul24((uint32_t)i, 1366u) >> 16), j = i - r * (kDW / 2);   // i / 48, exact for i < 2048 (srows <= 38)
                const uint32_t* row32 = reinterpret_cast<const uint32_t*>(S + __umul24((uint32_t)r, (uint32_t)sp)) + 2 * j;
                uint32_t e0 = init, o0 = init, e1 = init, o1 = init;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    tap_pair<DILATE>(row32, kx * C, e0, o0);
                    tap_pair<DILATE>(row32 + 1, kx * C, e1, o1);
                }
                uint32_t* h = H + 4 * i;
                h[0] = e0; h[1] = o0; h[2] = e1; h[3] = o1;
            }
        } else {
            for (int i = tid; i < kDW * srows; i += 256) {
                const int r = i / kDW, d = i - r * kDW;
                const uint32_t* row32 = reinterpret_cast<const uint32_t*>(S + r * sp);
                uint32_t acc_e = init, acc_o = init;
                for (int kx = 0; kx < a.kw; ++kx) tap_pair<DILATE>(row32, 4 * d + kx * C, acc_e, acc_o);
                H[2 * i] = acc_e;
                H[2 * i + 1] = acc_o;
            }
        }
        __syncthreads();
    }
    // 2b / 3. output rows
    const int row_bytes = a.w * C;
    if constexpr (K > 0) {
        // two output rows per item: they share K - 1 of their K + 1 rows of H
        static_assert(kDW == 96, "the multiply-shift below divides by 96");
        uint8_t* tile_dst = dst + (long long)y0 * row_bytes + (long long)x0 * C;   // wave-uniform 64-bit base; per-item offsets are 32-bit
        for (int i = tid; i < kDW * (kMorphTH / 2); i += 256) {
            const int rp = (int)(__umul24((uint32_t)i, 683u) >> 16), d = i - rp * kDW;   // i / 96, exact for i < 2048
            const int y = y0 + 2 * rp, fb = x0 * C + 4 * d;
            if (y >= a.h || fb >= row_bytes) continue;
            const uint32_t* h = H + 2 * (2 * rp * kDW + d);
            uint32_t he[K + 1], ho[K + 1];
#pragma unroll
            for (int ky = 0; ky <= K; ++ky) { he[ky] = h[2 * ky * kDW]; ho[ky] = h[2 * ky * kDW + 1]; }
            uint32_t me = he[1], mo = ho[1];   // rows 1 .. K - 1 are common to both outputs
#pragma unroll
            for (int ky = 2; ky < K; ++ky) { me = pk_minmax<DILATE>(me, he[ky]); mo = pk_minmax<DILATE>(mo, ho[ky]); }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (y + j >= a.h) break;
                const uint32_t out = __builtin_amdgcn_perm(pk_minmax<DILATE>(mo, ho[j ? K : 0]), pk_minmax<DILATE>(me, he[j ? K : 0]), 0x06020400u);
                uint8_t* o = tile_dst + (__umul24((uint32_t)(2 * rp + j), (uint32_t)row_bytes) + 4u * (uint32_t)d);
                if (fb + 4 <= row_bytes) *reinterpret_cast<u32_unaligned*>(o) = out;
                else for (int b = 0; fb + b < row_bytes; ++b) o[b] = (uint8_t)(out >> (8 * b));
            }
        }
        return;
    }
    for (int i = tid; i < kDW * kMorphTH; i += 256) {
        const int r = i / kDW, d = i - r * kDW;
        const int y = y0 + r;
        const int fb = x0 * C + 4 * d;                                       // flat byte inside the image row
        if (y >= a.h || fb >= row_bytes) continue;
        uint32_t acc_e = init, acc_o = init;
        if constexpr (BOX) {
            for (int ky = 0; ky < a.kh; ++ky) {
                acc_e = pk_minmax<DILATE>(acc_e, H[2 * (i + ky * kDW)]);
                acc_o = pk_minmax<DILATE>(acc_o, H[2 * (i + ky * kDW) + 1]);
            }
        } else {
            for (int ky = 0; ky < a.kh; ++ky) {
                uint32_t bits = a.rows[ky];
                const uint32_t* row32 = reinterpret_cast<const uint32_t*>(S + (r + ky) * sp);
                for (int kx = 0; bits; ++kx, bits >>= 1)
                    if (bits & 1u) tap_pair<DILATE>(row32, 4 * d + kx * C, acc_e, acc_o);
            }
        }
        const uint32_t out = __builtin_amdgcn_perm(acc_o, acc_e, 0x06020400u);   // {e.b0, o.b0, e.b2, o.b2}
        uint8_t* o = dst + (long long)y * row_bytes + fb;
        if (fb + 4 <= row_bytes) *reinterpret_cast<u32_unaligned*>(o) = out;
        else for (int b = 0; fb + b < row_bytes; ++b) o[b] = (uint8_t)(out >> (8 * b));
    }
}

template <int C>
int32_t launch_morph_tile(hipStream_t st, const Morph& a, bool box, int sp, int srows, size_t lds) {
    const dim3 grid = xcd_grid(a.tiles), blk(256);
#define KH_MORPH_TILE(D, B, KK)                                                                                                    \
    do {                                                                                                                          \
        if (lds > 48 * 1024)                                                                                                      \
            KH_HIP(hipFuncSetAttribute((const void*)morphology_u8_tile_kernel<C, D, B, KK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((morphology_u8_tile_kernel<C, D, B, KK>), grid, blk, lds, st, a, sp, srows);                            \
    } while (0)
#define KH_MORPH_BOX(D)                                                                                                            \
    do {                                                                                                                          \
        if (a.kw == a.kh && a.kw == 3) KH_MORPH_TILE(D, true, 3);                                                                  \
        else if (a.kw == a.kh && a.kw == 5) KH_MORPH_TILE(D, true, 5);                                                             \
        else if (a.kw == a.kh && a.kw == 7) KH_MORPH_TILE(D, true, 7);                                                             \
        else KH_MORPH_TILE(D, true, 0);                                                                                            \
    } while (0)
    if (a.op == 0) {
        if (box) KH_MORPH_BOX(true);
        else KH_MORPH_TILE(true, false, 0);
    } else {
        if (box) KH_MORPH_BOX(false);
        else KH_MORPH_TILE(false, false, 0);
    }
#undef KH_MORPH_BOX
#undef KH_MORPH_TILE
    return KH_OK;
}

template <typename T>
int32_t check_pyr(const char* what, const T* src, T* dst, int sw, int sh, int channels, int batch, int64_t ss, int64_t ds,
                  int dw, int dh) {
    KH_REQUIRE(sw > 0 && sh > 0, KH_ERR_INVALID_ARG, "%s: zero-sized image %dx%d", what, sw, sh);
    KH_REQUIRE(channels == 1 || channels == 3 || channels == 4, KH_ERR_UNSUPPORTED,
               "%s: no device kernel for %d channels (supported: 1, 3, 4)", what, channels);
    KH_REQUIRE(batch >= 0 && batch <= 65535, KH_ERR_TOO_LARGE, "%s: batch %d outside [0, 65535]", what, batch);
    KH_REQUIRE((int64_t)sw * sh * channels <= kI32Max && (int64_t)dw * dh * channels <= kI32Max, KH_ERR_TOO_LARGE,
               "%s: image exceeds 32-bit indexing", what);
    KH_REQUIRE(ss >= 0 && ds >= 0, KH_ERR_INVALID_ARG, "%s: negative batch stride", what);
    if (batch > 0) KH_REQUIRE(src && dst, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    return KH_OK;
}

#define KH_PYR_ENTRY(NAME, T, KERNEL, DW, DH)                                                                     \
    int32_t NAME(kh_stream_t stream, const T* src, T* dst, int32_t sw, int32_t sh, int32_t channels, int32_t batch, \
                 int64_t ss, int64_t ds) {                                                                        \
        const int dw = (DW), dh = (DH);                                                                           \
        if (int32_t rc = check_pyr(#NAME, src, dst, sw, sh, channels, batch, ss, ds, dw, dh)) return rc;          \
        if (batch == 0) return KH_OK;                                                                             \
        Pyr<T> a{src, dst, sw, sh, dw, dh, ss, ds,                                                                \
                 xcd_tiles(cdiv(dw, kBx), cdiv(dh, kBy), (unsigned)batch, cdiv(dw, kBx) * 8)};                    \
        KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, #NAME ": batch x tiles exceeds one launch");              \
        const dim3 blk(kBx, kBy), grid = xcd_grid(a.tiles);                                                       \
        hipStream_t st = as_hip(stream);                                                                          \
        if (channels == 1) hipLaunchKernelGGL(KERNEL<1>, grid, blk, 0, st, a);                                    \
        else if (channels == 3) hipLaunchKernelGGL(KERNEL<3>, grid, blk, 0, st, a);                               \
        else hipLaunchKernelGGL(KERNEL<4>, grid, blk, 0, st, a);                                                  \
        return check_launch(#NAME);                                                                               \
    }

KH_PYR_ENTRY(kh_pyrdown_u8_direct, uint8_t, pyrdown_u8_kernel, (sw + 1) / 2, (sh + 1) / 2)  // per-pixel kernel: test option pyr_direct = 1 only
KH_PYR_ENTRY(kh_pyrdown_f32_direct, float, pyrdown_f32_kernel, (sw + 1) / 2, (sh + 1) / 2)
KH_PYR_ENTRY(kh_pyrup_f32_direct, float, pyrup_f32_kernel, sw * 2, sh * 2)  // per destination pixel: test option pyr_direct = 1 only
KH_PYR_ENTRY(kh_pyrup_u8_direct, uint8_t, pyrup_u8_kernel, sw * 2, sh * 2)

}  // namespace

// The rolling 2x-upscale kernels (one source quad / eight source pixels per lane), shared by pyrup_u8 (AR = kUpPyr) and resize_fast_u8's
// exact 2x bilinear upscale (kUpRh: the reference's RGB case; kUpQ14: every other channel count).  Preconditions are the caller's:
// channels 1 (sw >= 8), 3 or 4 (sw >= 4), sw * 8 < 2^24, sw * sh * 4 * channels within 32 bits for one channel.
template <int AR>
static int32_t launch_up2_roll(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int sw, int sh, int channels, int batch, int64_t ss, int64_t ds,
                               const char* what) {
    const int dw = sw * 2, dh = sh * 2;
    const bool dword_dst = reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (batch <= 1 || ds % 4 == 0);
    PyrRoll r{src, dst, sw, sh, dw, dh, 0, ss, ds, XcdTiles{}, plain_row_stores((int64_t)dw * channels, dst, ds, batch)};
    const bool gray = channels == 1;
    const unsigned tiles_x = cdiv(sw, gray ? kPuGrayTileSrc : kPuRollTileSrc);
    const long long cols_blocks = (long long)tiles_x * batch;
    long long strips = (2048 + cols_blocks - 1) / cols_blocks;   // >= 8 blocks per CU
    const long long min_strips = cdiv(sh, 360), max_strips = cdiv(sh, 16);
    strips = strips < min_strips ? min_strips : (strips > max_strips ? max_strips : strips);
    r.th = (int)cdiv(sh, strips);
    r.tiles = xcd_tiles(tiles_x, cdiv(sh, r.th), (unsigned)batch, kXcdEighth);
    KH_REQUIRE(r.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
    const dim3 grid = xcd_grid(r.tiles);
    hipStream_t st = as_hip(stream);
    if (gray) {
        const bool dword_ok = sw % 2 == 0 && dword_dst;   // buffer stores need dword-aligned rows
        if (!dword_ok) r.plain = 2;
        if (sw % 8 != 0 || !dword_ok) hipLaunchKernelGGL((pyrup_u8_gray_roll_kernel<true, AR>), grid, dim3(256), 0, st, r);   // (any width / alignment: the RAGGED instantiation)
        else hipLaunchKernelGGL((pyrup_u8_gray_roll_kernel<false, AR>), grid, dim3(256), 0, st, r);
    } else if (channels == 4) {   // RGBA (round 6); plain = 2: a destination off a dword
        if (!dword_dst) r.plain = 2;
        hipLaunchKernelGGL((pyrup_u8_rgb_roll_kernel<4, AR>), grid, dim3(256), 0, st, r);
    } else hipLaunchKernelGGL((pyrup_u8_rgb_roll_kernel<3, AR>), grid, dim3(256), 0, st, r);
    return check_launch(what);
}
// resize_fast_u8's exact 2x bilinear / nearest upscale on the rolling kernels (kh_resize_u8.hip calls this; false = not taken: shapes the
// rolling kernels do not cover, or test option pyr_roll = 0)
namespace kh {
bool resize_up2_u8_rolling(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int sw, int sh, int channels, int batch, int64_t ss, int64_t ds,
                           const char* what, int32_t& rc, bool nearest, bool opencv) {
    const bool ok = (channels == 1 ? sw >= 8 && (int64_t)sw * sh * 4 <= kI32Max : (channels == 3 || channels == 4) && sw >= 4) && sh >= 2 &&
                    (int64_t)sw * 8 < (1 << 24) && (int64_t)sw * sh * 4 * channels <= kI32Max && dev_opt(kOptPyrRoll) != 0;
    if (!ok) return false;
    rc = nearest ? launch_up2_roll<kUpNearest>(stream, src, dst, sw, sh, channels, batch, ss, ds, what)
       : opencv ? launch_up2_roll<kUpCv>(stream, src, dst, sw, sh, channels, batch, ss, ds, what)
       : channels == 3 ? launch_up2_roll<kUpRh>(stream, src, dst, sw, sh, channels, batch, ss, ds, what)
                       : launch_up2_roll<kUpQ14>(stream, src, dst, sw, sh, channels, batch, ss, ds, what);
    return true;
}
}  // namespace kh

extern "C" {

int32_t kh_pyrdown_f32(kh_stream_t stream, const float* src, float* dst, int32_t sw, int32_t sh, int32_t channels, int32_t batch,
                       int64_t ss, int64_t ds) {
    if (dev_opt(kOptPyrDirect) == 1) return kh_pyrdown_f32_direct(stream, src, dst, sw, sh, channels, batch, ss, ds);   // test option: per-pixel kernel
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    if (int32_t rc = check_pyr("kh_pyrdown_f32", src, dst, sw, sh, channels, batch, ss, ds, dw, dh)) return rc;
    if (batch == 0) return KH_OK;
    PyrF32Roll r{src, dst, sw, sh, dw, dh, 0, channels, ss, ds, XcdTiles{}, plain_row_stores((int64_t)dw * channels * 4, dst, ds * 4, batch)};
    const unsigned tiles_x = cdiv(dw * channels, 256);
    const long long cols_blocks = (long long)tiles_x * batch;
    long long strips = (2048 + cols_blocks - 1) / cols_blocks;   // >= 8 blocks per CU
    const long long min_strips = cdiv(dh, kPdfStripMax), max_strips = cdiv(dh, 16);
    strips = strips < min_strips ? min_strips : (strips > max_strips ? max_strips : strips);
    r.th = (int)cdiv(dh, strips);
    r.tiles = xcd_tiles(tiles_x, cdiv(dh, r.th), (unsigned)batch, kXcdEighth);
    KH_REQUIRE(r.tiles.total > 0, KH_ERR_TOO_LARGE, "kh_pyrdown_f32: batch x tiles exceeds one launch");
    const dim3 grid = xcd_grid(r.tiles), blk(256);
    hipStream_t st = as_hip(stream);
    if (channels == 1) hipLaunchKernelGGL(pyrdown_f32_roll_kernel<1>, grid, blk, 0, st, r);
    else if (channels == 3) hipLaunchKernelGGL(pyrdown_f32_roll_kernel<3>, grid, blk, 0, st, r);
    else hipLaunchKernelGGL(pyrdown_f32_roll_kernel<4>, grid, blk, 0, st, r);
    return check_launch("kh_pyrdown_f32");
}
int32_t kh_pyrdown_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t sw, int32_t sh, int32_t channels, int32_t batch,
                      int64_t ss, int64_t ds) {
    const bool direct = dev_opt(kOptPyrDirect) == 1;
    // (rows of 2^24 bytes or more: the tile kernel forms its 32-bit offsets with 24-bit multiplies)
    if (direct || (int64_t)sw * channels >= (1 << 24)) return kh_pyrdown_u8_direct(stream, src, dst, sw, sh, channels, batch, ss, ds);
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    if (int32_t rc = check_pyr("kh_pyrdown_u8", src, dst, sw, sh, channels, batch, ss, ds, dw, dh)) return rc;
    if (batch == 0) return KH_OK;
    const bool no_roll = dev_opt(kOptPyrRoll) == 0;   // dev / test knob: the tile kernel
    if ((channels == 3 || channels == 4) && sw >= 8 && !no_roll) {   // RGB8 / RGBA8 (round 6): the rolling planar kernel
        const bool dword_ok = reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (batch <= 1 || ds % 4 == 0);
        PyrRoll r{src, dst, sw, sh, dw, dh, 0, ss, ds, XcdTiles{}, channels == 4 && !dword_ok ? 2 : plain_row_stores((int64_t)dw * channels, dst, ds, batch)};
        const unsigned tiles_x = cdiv(dw, kPdRollTileDst);
        const long long cols_blocks = (long long)tiles_x * batch;
        long long strips = (2048 + cols_blocks - 1) / cols_blocks;   // >= 8 blocks per CU
        const long long min_strips = cdiv(dh, 360), max_strips = cdiv(dh, 16);
        strips = strips < min_strips ? min_strips : (strips > max_strips ? max_strips : strips);
        r.th = (int)cdiv(dh, strips);
        r.tiles = xcd_tiles(tiles_x, cdiv(dh, r.th), (unsigned)batch, kXcdEighth);
        KH_REQUIRE(r.tiles.total > 0, KH_ERR_TOO_LARGE, "kh_pyrdown_u8: batch x tiles exceeds one launch");
        if (channels == 4) hipLaunchKernelGGL(pyrdown_u8_rgb_roll_kernel<4>, xcd_grid(r.tiles), dim3(256), 0, as_hip(stream), r);
        else hipLaunchKernelGGL(pyrdown_u8_rgb_roll_kernel<3>, xcd_grid(r.tiles), dim3(256), 0, as_hip(stream), r);
        return check_launch("kh_pyrdown_u8");
    }
    if (channels == 1 && sw >= 16 && !no_roll && (int64_t)dw * dh <= kI32Max) {   // one channel: the rolling gray kernel
        const bool dword_ok = dw % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (batch <= 1 || ds % 4 == 0);   // buffer stores need dword-aligned rows
        const bool ragged = sw % 16 != 0 || !dword_ok;   // (round 6: any width / alignment on the RAGGED instantiation)
        PyrRoll r{src, dst, sw, sh, dw, dh, 0, ss, ds, XcdTiles{}, dword_ok ? plain_row_stores((int64_t)dw, dst, ds, batch) : 2};
        const unsigned tiles_x = cdiv(dw, kPgRollTileDst);
        const long long cols_blocks = (long long)tiles_x * batch;
        long long strips = (2048 + cols_blocks - 1) / cols_blocks;   // >= 8 blocks per CU
        const long long min_strips = cdiv(dh, 360), max_strips = cdiv(dh, 16);
        strips = strips < min_strips ? min_strips : (strips > max_strips ? max_strips : strips);
        r.th = (int)cdiv(dh, strips);
        r.tiles = xcd_tiles(tiles_x, cdiv(dh, r.th), (unsigned)batch, kXcdEighth);
        KH_REQUIRE(r.tiles.total > 0, KH_ERR_TOO_LARGE, "kh_pyrdown_u8: batch x tiles exceeds one launch");
        if (ragged) hipLaunchKernelGGL(pyrdown_u8_gray_roll_kernel<true>, xcd_grid(r.tiles), dim3(256), 0, as_hip(stream), r);
        else hipLaunchKernelGGL(pyrdown_u8_gray_roll_kernel<false>, xcd_grid(r.tiles), dim3(256), 0, as_hip(stream), r);
        return check_launch("kh_pyrdown_u8");
    }
    Pyr<uint8_t> a{src, dst, sw, sh, dw, dh, ss, ds, xcd_tiles(cdiv(dw, kPdTW), cdiv(dh, kPdTH), (unsigned)batch, cdiv(dw, kPdTW) * 4)};
    KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, "kh_pyrdown_u8: batch x tiles exceeds one launch");
    const dim3 blk(256), grid = xcd_grid(a.tiles);
    hipStream_t st = as_hip(stream);
    if (channels == 1) hipLaunchKernelGGL(pyrdown_u8_tile_kernel<1>, grid, blk, 0, st, a);
    else if (channels == 3) hipLaunchKernelGGL(pyrdown_u8_tile_kernel<3>, grid, blk, 0, st, a);
    else hipLaunchKernelGGL(pyrdown_u8_tile_kernel<4>, grid, blk, 0, st, a);
    return check_launch("kh_pyrdown_u8");
}

// pyrup: one thread per source pixel (f32) / source pixel pair (u8) writes the 2 x 2 / 4 x 2 destination block
#define KH_PYRUP_ENTRY(NAME, T, KERNEL, PX)                                                                                         \
    int32_t NAME(kh_stream_t stream, const T* src, T* dst, int32_t sw, int32_t sh, int32_t channels, int32_t batch, int64_t ss,     \
                 int64_t ds) {                                                                                                     \
        const bool direct = dev_opt(kOptPyrDirect) == 1;                     \
        if (direct || sw < 2) return NAME##_direct(stream, src, dst, sw, sh, channels, batch, ss, ds); /* 1-pixel rows: own rule */ \
        const int dw = sw * 2, dh = sh * 2;                                                                                        \
        if (int32_t rc = check_pyr(#NAME, src, dst, sw, sh, channels, batch, ss, ds, dw, dh)) return rc;                           \
        if (batch == 0) return KH_OK;                                                                                              \
        const unsigned tx = cdiv(cdiv(sw, PX), kBx);                                                                               \
        Pyr<T> a{src, dst, sw, sh, dw, dh, ss, ds, xcd_tiles(tx, cdiv(sh, kBy), (unsigned)batch, tx * 8)};                         \
        KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, #NAME ": batch x tiles exceeds one launch");                               \
        const dim3 blk(kBx, kBy), grid = xcd_grid(a.tiles);                                                                        \
        hipStream_t st = as_hip(stream);                                                                                           \
        if (channels == 1) hipLaunchKernelGGL(KERNEL<1>, grid, blk, 0, st, a);                                                     \
        else if (channels == 3) hipLaunchKernelGGL(KERNEL<3>, grid, blk, 0, st, a);                                                \
        else hipLaunchKernelGGL(KERNEL<4>, grid, blk, 0, st, a);                                                                   \
        return check_launch(#NAME);                                                                                                \
    }
KH_PYRUP_ENTRY(kh_pyrup_f32, float, pyrup_f32_block_kernel, 1)
static int32_t kh_pyrup_u8_pairs_direct(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t sw, int32_t sh, int32_t channels, int32_t batch,
                                        int64_t ss, int64_t ds) {
    return kh_pyrup_u8_direct(stream, src, dst, sw, sh, channels, batch, ss, ds);
}
static KH_PYRUP_ENTRY(kh_pyrup_u8_pairs, uint8_t, pyrup_u8_pair_kernel, 2)   // every channel count; RGB8 takes the rolling kernel below
int32_t kh_pyrup_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t sw, int32_t sh, int32_t channels, int32_t batch, int64_t ss,
                    int64_t ds) {
    const bool direct = dev_opt(kOptPyrDirect) == 1;
    const bool no_roll = dev_opt(kOptPyrRoll) == 0;
    // one channel: the rolling gray kernel
    const bool gray = channels == 1 && sw >= 8 && (int64_t)sw * sh * 4 <= kI32Max;
    if (direct || no_roll || !(channels == 3 || channels == 4 || gray) || sw < 4 || (int64_t)sw * 8 >= (1 << 24)) return kh_pyrup_u8_pairs(stream, src, dst, sw, sh, channels, batch, ss, ds);
    const int dw = sw * 2, dh = sh * 2;
    if (int32_t rc = check_pyr("kh_pyrup_u8", src, dst, sw, sh, channels, batch, ss, ds, dw, dh)) return rc;
    if (batch == 0) return KH_OK;
    return launch_up2_roll<kUpPyr>(stream, src, dst, sw, sh, channels, batch, ss, ds, "kh_pyrup_u8");
}

// Kernel::new (P/morphology/kernels.rs:113-185): shape 0 box, 1 cross, 2 ellipse; out = width*height bytes
int32_t kh_morph_kernel(int32_t shape, int32_t width, int32_t height, uint8_t* out) {
    KH_REQUIRE(out && width > 0 && height > 0, KH_ERR_INVALID_ARG, "kh_morph_kernel: bad arguments");
    KH_REQUIRE(shape >= 0 && shape <= 2, KH_ERR_INVALID_ARG, "kh_morph_kernel: unknown shape %d", shape);
    KH_REQUIRE(shape == 2 || width == height, KH_ERR_INVALID_ARG, "kh_morph_kernel: box / cross kernels are square");
    for (int i = 0; i < width * height; ++i) out[i] = shape == 0 ? 1 : 0;
    if (shape == 1) {
        const int mid = width / 2;
        for (int j = 0; j < width; ++j) out[mid * width + j] = 1;
        for (int i = 0; i < width; ++i) out[i * width + mid] = 1;
    } else if (shape == 2) {
        const float cx = (float)width / 2.0f, cy = (float)height / 2.0f, rx = cx, ry = cy;
        for (int i = 0; i < height; ++i)
            for (int j = 0; j < width; ++j) {
                const float x = (float)j - cx, y = (float)i - cy;
                if ((x * x) / (rx * rx) + (y * y) / (ry * ry) <= 1.0f) out[i * width + j] = 1;
            }
    }
    return KH_OK;
}

int32_t kh_morphology_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t w, int32_t h, int32_t channels,
                         int32_t op, const uint8_t* mask, int32_t kw, int32_t kh_, int32_t border, const uint8_t* cval,
                         int32_t batch, int64_t ss, int64_t ds) {
    const char* what = "kh_morphology_u8";
    if (int32_t rc = check_pyr(what, src, dst, w, h, channels, batch, ss, ds, w, h)) return rc;
    KH_REQUIRE(op == KH_MORPH_DILATE || op == KH_MORPH_ERODE, KH_ERR_INVALID_ARG, "%s: unknown op %d", what, op);
    KH_REQUIRE(mask && kw > 0 && kh_ > 0, KH_ERR_INVALID_ARG, "%s: empty structuring element", what);
    KH_REQUIRE(kw <= 32 && kh_ <= 32, KH_ERR_UNSUPPORTED, "%s: structuring element %dx%d larger than 32x32", what, kw, kh_);
    KH_REQUIRE(border >= KH_BORDER_CONSTANT && border <= KH_BORDER_WRAP, KH_ERR_INVALID_ARG, "%s: unknown border mode %d", what, border);
    KH_REQUIRE(border != KH_BORDER_CONSTANT || cval, KH_ERR_INVALID_ARG, "%s: constant border needs a value", what);
    KH_REQUIRE(src != dst || batch == 0, KH_ERR_INVALID_ARG, "%s: in-place morphology is not supported", what);
    if (batch == 0) return KH_OK;
    Morph a;
    a.src = src; a.dst = dst; a.w = w; a.h = h; a.kw = kw; a.kh = kh_; a.border = border; a.op = op; a.ss = ss; a.ds = ds;
    for (int ky = 0; ky < 32; ++ky) {
        a.rows[ky] = 0;
        if (ky < kh_)
            for (int kx = 0; kx < kw; ++kx)
                if (mask[ky * kw + kx] == 1) a.rows[ky] |= 1u << kx;
    }
    for (int c = 0; c < 4; ++c) a.cval[c] = (cval && c < channels) ? cval[c] : 0;
    hipStream_t st = as_hip(stream);
    // tiled kernel (LDS-staged window; separable for all-ones masks) unless the mask has no active tap (the per-pixel kernel's
    // "no tap" rule), the window does not fit 150 KiB of LDS, or test option morph_direct = 1
    bool any = false, box = true;
    for (int ky = 0; ky < kh_; ++ky) { any = any || a.rows[ky]; box = box && a.rows[ky] == (kw == 32 ? 0xffffffffu : (1u << kw) - 1u); }
    const bool direct = dev_opt(kOptMorphDirect) == 1;
    const bool no_roll = dev_opt(kOptMorphRoll) == 0;   // dev / test knob: the tile kernel
    // the rolling kernels' shapes: the all-ones box, and kh_morph_kernel's cross / ellipse of 3 / 5 / 7 (unions of two rectangles)
    int shape = box ? kMsBox : -1;
    if (!box && kw == kh_ && (kw == 3 || kw == 5 || kw == 7)) {
        for (int sh = kMsCross; sh <= kMsEllipse && shape < 0; ++sh) {
            uint8_t ref[49];
            kh_morph_kernel(sh, kw, kw, ref);
            bool same = true;
            for (int ky = 0; ky < kw && same; ++ky)
                for (int kx = 0; kx < kw; ++kx) same = same && ((a.rows[ky] >> kx) & 1u) == ref[ky * kw + kx];
            if (same) shape = sh;
        }
    }
    // Square boxes of 9 .. 31 (round 6): max / min over a K-box IS the composition of boxes of 7 (and one of 3 / 5 / 7) — K = 1 + sum (k_i - 1),
    // exact, order-independent arithmetic — and every border mode extends the image evenly / periodically / by a constant, so
    // re-applying it to an intermediate equals the K-box on the padded source.  A chain of rolling-kernel passes through one scratch
    // image and `dst` replaces the LDS-tile kernel: 9 x 9 1.20 -> 0.7 ms, 15 x 15 2.6 -> 1.1, 31 x 31 4.6 -> 2.0 per 32 4K images
    // (profiles/r06zl_morph_chain.txt).  Without scratch (stream capture and no registered workspace) the tile kernel keeps the call.
    const bool gray_roll_ok = channels == 1 && w >= 16 && (int64_t)w * h <= kI32Max;
    const bool gray_dword_ok = w % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (batch <= 1 || ds % 4 == 0);   // buffer stores need dword-aligned rows
    const bool gray_ragged = w % 16 != 0 || !gray_dword_ok;   // (round 6: any width / alignment on the RAGGED instantiation)
    if (any && box && !direct && !no_roll && (channels == 3 || channels == 4 || gray_roll_ok) && kw == kh_ && (kw & 1) && kw >= 9 && kw <= 31 && border != KH_BORDER_WRAP && w >= 4 &&
        (int64_t)w * channels < (1 << 24) && dev_opt(kOptMorphRoll) != 2) {
        int chain[8], nchain = 0, rem = kw;
        while (rem > 7) { chain[nchain++] = 7; rem -= 6; }
        if (rem >= 3) chain[nchain++] = rem;
        const size_t img = ((size_t)w * h * channels + 15) & ~(size_t)15;
        Scratch scratch;
        if (get_scratch(stream, img * (size_t)batch, what, scratch) == KH_OK) {
            uint8_t* tmp = scratch.as<uint8_t>();
            const uint8_t* cur = src;
            long long cur_stride = ss;
            for (int i = 0; i < nchain; ++i) {
                const bool to_dst = ((nchain - 1 - i) & 1) == 0;   // the last pass writes dst; the ones before alternate
                uint8_t* out = to_dst ? dst : tmp;
                const long long out_stride = to_dst ? ds : (long long)img;
                uint8_t box_mask[49];
                for (int k = 0; k < chain[i] * chain[i]; ++k) box_mask[k] = 1;
                const uint8_t cv[4] = {(uint8_t)a.cval[0], (uint8_t)a.cval[1], (uint8_t)a.cval[2], (uint8_t)a.cval[3]};
                if (int32_t rc = kh_morphology_u8(stream, cur, out, w, h, channels, op, box_mask, chain[i], chain[i], border, cv, batch, cur_stride, out_stride)) return rc;
                cur = out; cur_stride = out_stride;
            }
            return KH_OK;
        }
    }
    if (any && shape >= 0 && !direct && !no_roll && gray_roll_ok && kw == kh_ && (kw == 3 || kw == 5 || kw == 7) && border != KH_BORDER_WRAP && dev_opt(kOptMorphRoll) != 2) {
        // one channel, square box / cross / ellipse of 3 / 5 / 7, rows of whole 16-pixel groups: the rolling gray kernel (test option morph_roll = 2: the tile kernel)
        MorphRoll r{src, dst, w, h, 0, border, ss, ds, {a.cval[0], 0, 0}, XcdTiles{}, gray_dword_ok ? plain_row_stores((int64_t)w, dst, ds, batch) : 2};
        const unsigned tiles_x = cdiv(w, kMgTilePx);
        const long long cols_blocks = (long long)tiles_x * batch;
        long long strips = (2048 + cols_blocks - 1) / cols_blocks;
        const long long min_strips = cdiv(h, 360), max_strips = cdiv(h, 32);
        strips = strips < min_strips ? min_strips : (strips > max_strips ? max_strips : strips);
        r.th = (int)cdiv(h, strips);
        r.tiles = xcd_tiles(tiles_x, cdiv(h, r.th), (unsigned)batch, kXcdEighth);
        KH_REQUIRE(r.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
        const dim3 grid = xcd_grid(r.tiles);
        const bool dil = op == KH_MORPH_DILATE;
#define KH_MG_R(KK, SH, RG)                                                                                      \
    do {                                                                                                         \
        if (dil) hipLaunchKernelGGL((morph_u8_gray_roll_kernel<KK, true, SH, RG>), grid, dim3(256), 0, st, r);   \
        else hipLaunchKernelGGL((morph_u8_gray_roll_kernel<KK, false, SH, RG>), grid, dim3(256), 0, st, r);      \
    } while (0)
#define KH_MG_S(KK, SH) do { if (gray_ragged) KH_MG_R(KK, SH, true); else KH_MG_R(KK, SH, false); } while (0)
#define KH_MG(KK) do { if (shape == kMsBox) KH_MG_S(KK, kMsBox); else if (shape == kMsCross) KH_MG_S(KK, kMsCross); else KH_MG_S(KK, kMsEllipse); } while (0)
        if (kw == 3) KH_MG(3);
        else if (kw == 5) KH_MG(5);
        else KH_MG(7);
#undef KH_MG
#undef KH_MG_S
#undef KH_MG_R
        return check_launch(what);
    }
    if (any && shape >= 0 && !direct && !no_roll && (channels == 3 || channels == 4) && kw == kh_ && (kw == 3 || kw == 5 || kw == 7) && border != KH_BORDER_WRAP && w >= 4 &&
        (int64_t)w * channels < (1 << 24) && (shape == kMsBox || dev_opt(kOptMorphRoll) != 2) && (channels == 3 || dev_opt(kOptMorphRoll) != 2)) {
        // RGB8 / RGBA8 (round 6), square box / cross / ellipse of 3 / 5 / 7: the rolling planar kernel
        const bool dword_ok = reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (batch <= 1 || ds % 4 == 0);
        MorphRoll r{src, dst, w, h, 0, border, ss, ds, {a.cval[0], a.cval[1], a.cval[2], a.cval[3]}, XcdTiles{},
                    channels == 4 && !dword_ok ? 2 : plain_row_stores((int64_t)w * channels, dst, ds, batch)};
        const unsigned tiles_x = cdiv(w, kMrTilePx);
        const long long cols_blocks = (long long)tiles_x * batch;
        long long strips = (2048 + cols_blocks - 1) / cols_blocks;   // >= 8 blocks per CU
        const long long min_strips = cdiv(h, 360), max_strips = cdiv(h, 32);
        strips = strips < min_strips ? min_strips : (strips > max_strips ? max_strips : strips);
        r.th = (int)cdiv(h, strips);
        r.tiles = xcd_tiles(tiles_x, cdiv(h, r.th), (unsigned)batch, kXcdEighth);
        KH_REQUIRE(r.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
        const dim3 grid = xcd_grid(r.tiles);
        const bool dil = op == KH_MORPH_DILATE;
#define KH_MR_C(KK, SH, CC)                                                                                     \
    do {                                                                                                        \
        if (dil) hipLaunchKernelGGL((morph_u8_rgb_roll_kernel<KK, true, SH, CC>), grid, dim3(256), 0, st, r);   \
        else hipLaunchKernelGGL((morph_u8_rgb_roll_kernel<KK, false, SH, CC>), grid, dim3(256), 0, st, r);      \
    } while (0)
#define KH_MR_S(KK, SH) do { if (channels == 4) KH_MR_C(KK, SH, 4); else KH_MR_C(KK, SH, 3); } while (0)
#define KH_MR(KK) do { if (shape == kMsBox) KH_MR_S(KK, kMsBox); else if (shape == kMsCross) KH_MR_S(KK, kMsCross); else KH_MR_S(KK, kMsEllipse); } while (0)
        if (kw == 3) KH_MR(3);
        else if (kw == 5) KH_MR(5);
        else KH_MR(7);
#undef KH_MR
#undef KH_MR_S
#undef KH_MR_C
        return check_launch(what);
    }
    const int srows = kMorphTH + kh_ - 1, sp = ((kMorphFW + (kw - 1) * channels + 3) & ~3) + 4;
    const size_t lds = (size_t)srows * sp + (box ? (size_t)srows * kMorphFW * 2 : 0);
    // (row bytes < 2^24: the tile kernel forms its 32-bit offsets with 24-bit multiplies)
    if (any && !direct && lds <= 150 * 1024 && (int64_t)w * h * channels <= kI32Max - 8 && (int64_t)w * channels < (1 << 24)) {
        a.tiles = xcd_tiles(cdiv((int64_t)w * channels, kMorphFW), cdiv(h, kMorphTH), (unsigned)batch, cdiv((int64_t)w * channels, kMorphFW) * 4);
        KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
        const int32_t rc = channels == 1 ? launch_morph_tile<1>(st, a, box, sp, srows, lds)
                         : channels == 3 ? launch_morph_tile<3>(st, a, box, sp, srows, lds) : launch_morph_tile<4>(st, a, box, sp, srows, lds);
        return rc ? rc : check_launch(what);
    }
    a.tiles = xcd_tiles(cdiv(w, kBx), cdiv(h, kBy), (unsigned)batch, cdiv(w, kBx) * 8);
    KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
    const dim3 blk(kBx, kBy), grid = xcd_grid(a.tiles);
    if (channels == 1) hipLaunchKernelGGL(morphology_u8_kernel<1>, grid, blk, 0, st, a);
    else if (channels == 3) hipLaunchKernelGGL(morphology_u8_kernel<3>, grid, blk, 0, st, a);
    else hipLaunchKernelGGL(morphology_u8_kernel<4>, grid, blk, 0, st, a);
    return check_launch(what);
}

}  // extern "C"
