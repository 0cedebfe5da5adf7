// Fused camera preprocess for gfx950: raw frame (RGB/BGR/RGBA/BGRA/Gray/NV12/YUYV) ->
// sampled, normalised, channel-planar f32/f16 tensor in one pass.
//
// Behavioural contract = the reference's NVRTC kernel family
// `resize_normalize_to_chw_{bilinear,nearest,lanczos}[_f16]`
// (crates/kornia-imgproc/src/preprocess.rs:430-647) and its launcher `launch_view`
// (:1324-1375).  The reference compiles with fmad=false; this file is built with
// -ffp-contract=off and IEEE division, and keeps the same expression trees, so results are
// bit-identical for nearest/bilinear (Lanczos calls sinf and is only loosely comparable, as in
// the reference's own cpu_close_to_cuda test, :1685).
//
// Two kernels:
//   * preprocess_generic  — one thread per destination pixel, any geometry/format/sampler, 64x4
//     blocks over the destination, grid.z = frame (the reference launches once per frame,
//     :1277-1280).  Taps are cached gathers; an LDS-staged-window variant was built and measured
//     1.7x SLOWER on 1080p->640x640 (4.1 vs 2.4 ms / 1024 frames) and dropped.
//   * preprocess_nv12_identity — the north-star case (NV12, scale 1, pad 0, same size): every
//     source byte is needed exactly once (1.5 B/px) and 12 B/px are written.  A thread owns 4
//     pixels of one row: one dword luma load, one dword chroma load (2 UV pairs), three
//     16-byte plane stores, so a wave writes 1 KiB contiguous per store instruction.
//     At scale 1 the bilinear weights are exactly 0 (`ax == ay == 0.0f`), hence
//     `t00 + (t10 - t00) * 0 == t00` for the finite 0..255 taps and the result equals the
//     generic kernel bit for bit; tests assert that equality.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <tuple>
#include <vector>

#include "kh_common.h"
#include "kh_table_cache.h"

using namespace kh;

namespace {

struct PreArgs {
    float scale_x, scale_y, pad_x, pad_y;
    int src_w, src_h, src_pitch, src_bpp;
    int dst_w, dst_h;
    float m0, m1, m2, is0, is1, is2;
    float pad_value;
    long long src_frame_stride;  // bytes
    long long dst_frame_stride;  // elements
    float rc_x, rc_y;            // 1 / scale_x, 1 / scale_y
    int fast_div;                // host-verified: the 3-op quotient equals IEEE division on this grid
    const float* lz_wx;          // Lanczos only: 6 axis weights per destination column / row, built on the host (see lanczos_tables)
    const float* lz_wy;
    int quad_wide;               // preprocess_generic_quads, NV12 / YUYV: the taps of every destination quad fit one 16-byte run per plane row (host-checked)
    int f16_plain;               // f16 outputs of the nearest / bilinear samplers: no value of the launch reaches 2^16 (host-proved, f16_in_range): plain conversions
};

// kh_preprocess_to_chw_list: the reference's `run_raw_batch(frames: &[&CudaSlice<u8>], ..)` (P/preprocess.rs:1258-1282) hands over
// separately allocated frame buffers and launches once per frame.  Here the bases of up to kFrameListMax frames travel by value in the
// kernel arguments (2 KiB; a block reads its frame's base with one scalar load) and the batch goes out as ceil(n / 256) launches:
// 4.374 ms against 4.373 ms for the equally spaced form on the north star (profiles/r06a_ubench_nv12_one_store.txt).
// The list kernels are separate instantiations (LIST = true): selecting at run time inside one kernel cost the equally spaced launches
// of the short-lived gather kernels a few per cent (round 6, profiles/r06c / r06i), so LIST = false is the round-5 kernel argument for argument.
constexpr int kFrameListMax = 256;
struct FrameList { const uint8_t* p[kFrameListMax]; };
struct NoFrames { int unused; };
template <bool LIST> struct FrameArg { typedef NoFrames type; };
template <> struct FrameArg<true> { typedef FrameList type; };
template <bool LIST>
__device__ __forceinline__ const uint8_t* frame_base(const typename FrameArg<LIST>::type& fl, const PreArgs& a, const uint8_t* src_base, unsigned frame) {
    if constexpr (LIST) return fl.p[frame];
    else return src_base + (long long)frame * a.src_frame_stride;
}

// BT.601 limited-range Q20 decode, constants of P/color/yuv/kernels.rs:696-702 and the fused
// kernel's bt601_q20_to_rgb (P/preprocess.rs:501-508).
constexpr int kCY = 1220542, kCUB = 2116026, kCUG = -409993, kCVG = -852492, kCVR = 1673527;
constexpr int kHalf20 = 1 << 19;

__device__ __forceinline__ int clamp255(int v) { return min(max(v, 0), 255); }

__device__ __forceinline__ void bt601_q20_to_rgb(int yv, int u, int v, float px[3]) {
    int yy = max(yv - 16, 0) * kCY;
    u -= 128;
    v -= 128;
    px[2] = (float)clamp255((yy + kCUB * u + kHalf20) >> 20);
    px[1] = (float)clamp255((yy + kCUG * u + kCVG * v + kHalf20) >> 20);
    px[0] = (float)clamp255((yy + kCVR * v + kHalf20) >> 20);
}

// WIDE: the frame base / pitch alignment lets a tap's chroma pair (NV12, 2 B), YUYV group (4 B) or
// RGBA pixel (4 B) come in with ONE load instead of 2-3 byte loads (checked on the host).
template <int FMT, bool WIDE>
__device__ __forceinline__ void tap_rgb(const uint8_t* __restrict__ src, int x, int y,
                                         const PreArgs& a, float px[3]) {
    if constexpr (FMT == KH_FMT_RGB || FMT == KH_FMT_BGR) {
        const uint8_t* p = src + (unsigned)(y * a.src_pitch + x * a.src_bpp);  // < 2^31, host-checked
        int c0, c1, c2;
        if (WIDE && a.src_bpp == 4) {
            const uint32_t q = *reinterpret_cast<const uint32_t*>(p);
            c0 = q & 0xFF; c1 = (q >> 8) & 0xFF; c2 = (q >> 16) & 0xFF;
        } else {
            c0 = p[0]; c1 = p[1]; c2 = p[2];
        }
        if constexpr (FMT == KH_FMT_RGB) {
            px[0] = (float)c0; px[1] = (float)c1; px[2] = (float)c2;
        } else {
            px[0] = (float)c2; px[1] = (float)c1; px[2] = (float)c0;
        }
    } else if constexpr (FMT == KH_FMT_GRAY) {
        float v = (float)src[(unsigned)(y * a.src_pitch + x)];
        px[0] = v; px[1] = v; px[2] = v;
    } else if constexpr (FMT == KH_FMT_NV12) {
        int yv = src[(unsigned)(y * a.src_w + x)];
        const uint8_t* uv = src + (unsigned)(a.src_w * a.src_h + (y >> 1) * a.src_w + (x >> 1) * 2);
        if constexpr (WIDE) {
            const uint32_t q = *reinterpret_cast<const uint16_t*>(uv);
            bt601_q20_to_rgb(yv, q & 0xFF, q >> 8, px);
        } else {
            bt601_q20_to_rgb(yv, uv[0], uv[1], px);
        }
    } else {  // YUYV
        const uint8_t* grp = src + (unsigned)(y * a.src_pitch + (x >> 1) * 4);
        if constexpr (WIDE) {
            const uint32_t q = *reinterpret_cast<const uint32_t*>(grp);
            bt601_q20_to_rgb((x & 1) ? (q >> 16) & 0xFF : q & 0xFF, (q >> 8) & 0xFF, q >> 24, px);
        } else {
            int yv = grp[(x & 1) ? 2 : 0];
            bt601_q20_to_rgb(yv, grp[1], grp[3], px);
        }
    }
}

template <int FMT, bool WIDE>
__device__ __forceinline__ void nearest_tap(const uint8_t* __restrict__ src, float sx, float sy,
                                               const PreArgs& a, float px[3]) {
    int xn = min(max((int)roundf(sx), 0), a.src_w - 1);
    int yn = min(max((int)roundf(sy), 0), a.src_h - 1);
    tap_rgb<FMT, WIDE>(src, xn, yn, a, px);
}

// The two horizontal taps of a bilinear sample, (x0, y) and (x1, y) with x1 = x0 + 1 (or x0 at the last
// column), decoded from ONE or TWO loads instead of 2-6: the gather kernels are bound by the number of
// vector-memory instructions per pixel (r01n: removing 15 % of the ALU work changed nothing), and the
// taps are adjacent in memory.  Unaligned 2/4/8-byte loads are fine on gfx950 global memory; every wide
// load stays inside the row (and the chroma row), so nothing is read past the frame.
typedef uint16_t u16u __attribute__((aligned(1)));
typedef uint32_t u32u __attribute__((aligned(1)));
typedef uint64_t u64u __attribute__((aligned(1)));

// Branch-free: the wide load is taken from a base clamped so that it stays inside the row, then shifted;
// a parity- or edge-dependent branch here would diverge in every wave and serialise the loads (measured:
// 2.69 vs 2.31 ms on 1080p->640).  Needs rows of at least 2 pixels (4 for the 4:2:x formats) — narrower
// sources take the per-tap path.
template <int FMT>
__device__ __forceinline__ void fetch_pair(const uint8_t* __restrict__ src, int x0, bool has_next, int y, const PreArgs& a,
                                           float l[3], float r[3]) {
    if constexpr (FMT == KH_FMT_RGB || FMT == KH_FMT_BGR) {
        const int xb = min(x0, a.src_w - 2);                 // pixels xb, xb + 1 are in the row
        const bool second = x0 != xb;                         // x0 is the last column
        const uint8_t* p = src + (unsigned)(y * a.src_pitch + xb * a.src_bpp);
        uint32_t c[6];
        if (a.src_bpp == 4) {                                 // uniform
            const uint64_t v = *reinterpret_cast<const u64u*>(p);
            const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
            c[0] = lo & 0xFF; c[1] = (lo >> 8) & 0xFF; c[2] = (lo >> 16) & 0xFF;
            c[3] = hi & 0xFF; c[4] = (hi >> 8) & 0xFF; c[5] = (hi >> 16) & 0xFF;
        } else {
            const uint32_t lo = *reinterpret_cast<const u32u*>(p);
            const uint32_t hi = *reinterpret_cast<const u16u*>(p + 4);
            c[0] = lo & 0xFF; c[1] = (lo >> 8) & 0xFF; c[2] = (lo >> 16) & 0xFF;
            c[3] = lo >> 24; c[4] = hi & 0xFF; c[5] = hi >> 8;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t lv = second ? c[3 + k] : c[k], rv = has_next ? c[3 + k] : lv;
            const int o = (FMT == KH_FMT_RGB) ? k : 2 - k;
            l[o] = (float)lv;
            r[o] = (float)rv;
        }
    } else if constexpr (FMT == KH_FMT_GRAY) {
        const int xb = min(x0, a.src_w - 2);
        const uint32_t v = (uint32_t)*reinterpret_cast<const u16u*>(src + (unsigned)(y * a.src_pitch + xb)) >> ((x0 - xb) * 8);
        const uint32_t g0 = v & 0xFF, g1 = has_next ? v >> 8 : g0;
        l[0] = l[1] = l[2] = (float)g0;
        r[0] = r[1] = r[2] = (float)g1;
    } else if constexpr (FMT == KH_FMT_NV12) {
        const int xb = min(x0, a.src_w - 2);
        const uint32_t yv = (uint32_t)*reinterpret_cast<const u16u*>(src + (unsigned)(y * a.src_w + xb)) >> ((x0 - xb) * 8);
        const uint32_t y0v = yv & 0xFF, y1v = has_next ? yv >> 8 : y0v;
        const int c0 = x0 >> 1, cb = min(c0, (a.src_w >> 1) - 2);  // chroma pairs cb, cb + 1 are in the row
        const uint32_t q = *reinterpret_cast<const u32u*>(src + (unsigned)(a.src_w * a.src_h + (y >> 1) * a.src_w + cb * 2)) >>
                           ((c0 - cb) * 16);
        const uint32_t uv0 = q & 0xFFFF, uv1 = (has_next && (x0 & 1)) ? q >> 16 : uv0;
        bt601_q20_to_rgb((int)y0v, (int)(uv0 & 0xFF), (int)(uv0 >> 8), l);
        bt601_q20_to_rgb((int)y1v, (int)(uv1 & 0xFF), (int)(uv1 >> 8), r);
    } else {  // YUYV: groups of 4 bytes Y0 U Y1 V per 2 pixels
        const int g0 = x0 >> 1, gb = min(g0, (a.src_w >> 1) - 2);
        const uint64_t v = *reinterpret_cast<const u64u*>(src + (unsigned)(y * a.src_pitch + gb * 4)) >> ((g0 - gb) * 32);
        const uint32_t q0 = (uint32_t)v, q1 = (has_next && (x0 & 1)) ? (uint32_t)(v >> 32) : q0;
        const int x1 = has_next ? x0 + 1 : x0;
        bt601_q20_to_rgb((int)((x0 & 1) ? (q0 >> 16) & 0xFF : q0 & 0xFF), (int)((q0 >> 8) & 0xFF), (int)(q0 >> 24), l);
        bt601_q20_to_rgb((int)((x1 & 1) ? (q1 >> 16) & 0xFF : q1 & 0xFF), (int)((q1 >> 8) & 0xFF), (int)(q1 >> 24), r);
    }
}

template <int FMT, bool WIDE>
__device__ __forceinline__ void bilinear_quad(const uint8_t* __restrict__ src, float sx, float sy,
                                                const PreArgs& a, float px[3]) {
    int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    float ax = sx - (float)x0, ay = sy - (float)y0;
    // sx, sy >= 0 here (plan_pixel rejects negatives), so x0, y0 >= 0 and x1 = min(x0 + 1, src_w - 1)
    const bool has_next = x0 + 1 <= a.src_w - 1;
    const int y1 = min(y0 + 1, a.src_h - 1);
    x0 = max(x0, 0);
    y0 = max(y0, 0);
    float t00[3], t10[3], t01[3], t11[3];
    constexpr int kMinW = (FMT == KH_FMT_NV12 || FMT == KH_FMT_YUYV) ? 4 : 2;
    if (a.src_w >= kMinW) {  // uniform
        fetch_pair<FMT>(src, x0, has_next, y0, a, t00, t10);
        fetch_pair<FMT>(src, x0, has_next, y1, a, t01, t11);
    } else {
        const int x1 = has_next ? x0 + 1 : x0;
        tap_rgb<FMT, WIDE>(src, x0, y0, a, t00);
        tap_rgb<FMT, WIDE>(src, x1, y0, a, t10);
        tap_rgb<FMT, WIDE>(src, x0, y1, a, t01);
        tap_rgb<FMT, WIDE>(src, x1, y1, a, t11);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float top = t00[c] + (t10[c] - t00[c]) * ax;
        float bot = t01[c] + (t11[c] - t01[c]) * ax;
        px[c] = top + (bot - top) * ay;
    }
}

// 1-D Lanczos-3 weight, the reference's expression (P/preprocess.rs:481-487).  It runs on the HOST: the six horizontal weights of a
// destination column depend only on the column, the six vertical ones only on the row, so a launch needs 6 * (dst_w + dst_h) of
// them and they are built once per geometry with the host's libm `sinf` — the function the reference's CPU side (`f32::sin`) and
// the restatement call.  Round 2 evaluated them per pixel with the device `sinf`, whose last-bit differences from libm left the
// fused Lanczos output within 2e-4 of the restatement instead of north_star's 1e-6; with the tables it is bit-identical, and the
// kernel loses 24 sinf evaluations per pixel.
inline float lanczos_w(float d) {
    float ad = fabsf(d);
    if (ad < 1e-6f) return 1.0f;
    if (ad >= 3.0f) return 0.0f;
    float pd = 3.14159265358979f * d;
    return 3.0f * sinf(pd) * sinf(pd / 3.0f) / (pd * pd);
}

// One row of the 6-tap Lanczos window, pixels clamp(x0 - 2 + i, 0, w - 1) for i = 0..5, decoded to float RGB.
// The six taps of a row are adjacent in memory, so they come from ONE or TWO wide loads — NV12: 8 luma bytes + 4 chroma pairs,
// Gray: 8 bytes, YUYV: 4 groups — taken from a window base clamped into the row; a tap picks its byte / pair by a variable shift,
// which also implements the border replication (clamped taps index the edge pixel again).  72 loads per destination pixel in the
// per-tap form (6 x 6 x (luma byte + chroma pair)) become 12.  Interleaved RGB / BGR taps are one unaligned dword each (36 loads
// instead of 108).  Needs rows of at least 8 pixels; narrower sources take the per-tap path.
template <int FMT>
__device__ __forceinline__ void window_row6(const uint8_t* __restrict__ src, int x0, int yc, const PreArgs& a, float t[6][3]) {
    if constexpr (FMT == KH_FMT_NV12 || FMT == KH_FMT_GRAY) {
        const int pitch = FMT == KH_FMT_NV12 ? a.src_w : a.src_pitch;
        const int xb = min(max(x0 - 2, 0), a.src_w - 8);                      // bytes xb .. xb + 7 are in the row
        const uint64_t yw = *reinterpret_cast<const u64u*>(src + (unsigned)(yc * pitch + xb));
        uint64_t cw = 0;
        int cb = 0;
        if constexpr (FMT == KH_FMT_NV12) {
            cb = min(max((x0 - 2) >> 1, 0), (a.src_w >> 1) - 4);              // chroma pairs cb .. cb + 3 are in the row
            cw = *reinterpret_cast<const u64u*>(src + (unsigned)(a.src_w * a.src_h + (yc >> 1) * a.src_w + cb * 2));
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int xc = min(max(x0 - 2 + i, 0), a.src_w - 1);
            const int yv = (int)((yw >> (8 * (xc - xb))) & 0xFF);
            if constexpr (FMT == KH_FMT_NV12) {
                const uint32_t uv = (uint32_t)(cw >> (16 * ((xc >> 1) - cb))) & 0xFFFFu;
                bt601_q20_to_rgb(yv, (int)(uv & 0xFF), (int)(uv >> 8), t[i]);
            } else {
                t[i][0] = t[i][1] = t[i][2] = (float)yv;
            }
        }
    } else if constexpr (FMT == KH_FMT_YUYV) {
        const int gb = min(max((x0 - 2) >> 1, 0), (a.src_w >> 1) - 4);        // groups gb .. gb + 3 (Y0 U Y1 V) are in the row
        const uint8_t* p = src + (unsigned)(yc * a.src_pitch + gb * 4);
        const uint64_t lo = *reinterpret_cast<const u64u*>(p), hi = *reinterpret_cast<const u64u*>(p + 8);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int xc = min(max(x0 - 2 + i, 0), a.src_w - 1);
            const int g = (xc >> 1) - gb;                                     // 0 .. 3
            const uint32_t q = (uint32_t)((g & 2 ? hi : lo) >> (32 * (g & 1)));
            bt601_q20_to_rgb((int)((xc & 1) ? (q >> 16) & 0xFF : q & 0xFF), (int)((q >> 8) & 0xFF), (int)(q >> 24), t[i]);
        }
    } else {  // interleaved RGB / BGR, 3 or 4 bytes per pixel: one unaligned dword per tap, kept inside the surface
        const unsigned last = (unsigned)((a.src_h - 1) * a.src_pitch + a.src_w * a.src_bpp) - 4u;   // last dword that is inside
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int xc = min(max(x0 - 2 + i, 0), a.src_w - 1);
            const unsigned off = (unsigned)(yc * a.src_pitch + xc * a.src_bpp), ob = min(off, last);
            const uint32_t q = *reinterpret_cast<const u32u*>(src + ob) >> (8 * (off - ob));
            const float c0 = (float)(q & 0xFF), c1 = (float)((q >> 8) & 0xFF), c2 = (float)((q >> 16) & 0xFF);
            if constexpr (FMT == KH_FMT_RGB) { t[i][0] = c0; t[i][1] = c1; t[i][2] = c2; }
            else { t[i][0] = c2; t[i][1] = c1; t[i][2] = c0; }
        }
    }
}

// Lanczos-3 over the 6 x 6 window (P/preprocess.rs:566-591).  The twelve axis weights are evaluated once per pixel (the reference
// kernel re-evaluates the horizontal weight inside the row loop: 42 sinf pairs instead of 12 — same values, same products); rows come
// in through window_row6.  The accumulation order — rows outer, taps inner, w = wy * wx, acc += w * t, wsum += w, one division at the
// end — is the reference's, so the result is bit-identical to the per-tap form it replaces.
template <int FMT, bool WIDE>
__device__ __forceinline__ void lanczos_window(const uint8_t* __restrict__ src, float sx, float sy, int ox, int oy,
                                               const PreArgs& a, float px[3]) {
    const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    float wx[6], wy[6];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2* wxp = reinterpret_cast<const f32x2*>(a.lz_wx + 6 * ox);   // lanczos_w(sx - (x0 - 2 + i)), i = 0..5 (24-byte rows)
    const f32x2* wyp = reinterpret_cast<const f32x2*>(a.lz_wy + 6 * oy);   // wave-uniform: one row per wave
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x2 u = wxp[i], v = wyp[i];
        wx[2 * i] = u[0]; wx[2 * i + 1] = u[1];
        wy[2 * i] = v[0]; wy[2 * i + 1] = v[1];
    }
    float acc[3] = {0.0f, 0.0f, 0.0f};
    float wsum = 0.0f;
    const bool wide_rows = a.src_w >= 8;  // uniform; narrower sources take the per-tap loads
#pragma unroll 1
    for (int j = 0; j < 6; ++j) {
        const int yc = min(max(y0 - 2 + j, 0), a.src_h - 1);
        float t[6][3];
        if (wide_rows) {
            window_row6<FMT>(src, x0, yc, a, t);
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) tap_rgb<FMT, WIDE>(src, min(max(x0 - 2 + i, 0), a.src_w - 1), yc, a, t[i]);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float w = wy[j] * wx[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] += w * t[i][c];
            wsum += w;
        }
    }
    px[0] = acc[0] / wsum;
    px[1] = acc[1] / wsum;
    px[2] = acc[2] / wsum;
}

// f32 -> binary16 bits, round-to-nearest-even.  v_cvt_f16_f32 implements exactly the
// reference's manual f2h (P/preprocess.rs:452-477) for every |f| < 65536 (f16 denormals are on
// for gfx9 kernels).  At or above 2^16 the reference does NOT saturate to Inf: its `exp >= 31`
// branch returns sign|0x7C00 plus the quiet bit whenever the f32 mantissa is non-zero, so a
// large finite value such as 1e9 becomes a NaN pattern (0x7E00) and only exact powers of two
// and Inf map to 0x7C00.  Reproduced here bit for bit.
__device__ __forceinline__ unsigned short f2h_bits(float f) {
    unsigned int x = __float_as_uint(f);
    if ((x & 0x7FFFFFFFu) >= 0x47800000u)
        return (unsigned short)(((x >> 16) & 0x8000u) | 0x7C00u | ((x & 0x7FFFFFu) ? 0x0200u : 0u));
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {   // two round-to-nearest-even conversions, one dword (valid below 2^16)
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

template <typename OutT>
__device__ __forceinline__ OutT to_out(float v);
template <>
__device__ __forceinline__ float to_out<float>(float v) { return v; }
template <>
__device__ __forceinline__ unsigned short to_out<unsigned short>(float v) { return f2h_bits(v); }

// The generic kernel is VALU-issue bound (about 250 instructions per pixel for NV12 bilinear, 17
// IEEE divisions of ~10 instructions each per 4-pixel thread), so its divisions use the 3-operation
// quotient q = n*rc; r = fma(-q, d, n); q' = fma(r, rc, q) wherever that is PROVEN equal to IEEE
// division:
//  * x / 255: exhaustively equal for every finite float (tests/test_host_math.py sweeps all 2^32
//    patterns through the host twin with the same IEEE fma); `r == 0 ? q : ...` keeps -0 -> -0.
//  * (o - pad) / scale (plan_pixel, P/preprocess.rs:437-448): the host evaluates both forms for every
//    destination column and row of the launch (dst_w + dst_h values) and sets `fast_div` only if all
//    agree bit for bit; otherwise the kernel divides.
__device__ __forceinline__ float div255_any(float x) {
    const float rc = 1.0f / 255.0f;
    const float q = x * rc;
    const float r = __builtin_fmaf(-q, 255.0f, x);
    return r == 0.0f ? q : __builtin_fmaf(r, rc, q);
}
__host__ __device__ __forceinline__ float quot3(float n, float d, float rc) {
    const float q = n * rc;
    const float r = __builtin_fmaf(-q, d, n);
    return __builtin_fmaf(r, rc, q);
}

// 64x4-thread blocks; a thread owns kGenPx destination pixels of one row, 64 apart, so every
// load / store instruction of a wave still covers 64 consecutive pixels while kGenPx independent
// tap gathers are in flight per lane (the kernel is latency-bound: 8 waves/SIMD x 1 pixel measured
// 2.46 ms on 1080p -> 640x640 x 1024; no per-pixel integer division either).  grid.z = frame.
constexpr int kGenPx = 4;
constexpr int kSampleBilinearOnGrid = 100;   // internal sampler id: bilinear whose taps all sit on whole source pixels (bilinear_taps_on_grid below)
template <int FMT, int SAMPLER, typename OutT, bool WIDE, bool LIST>
__global__ __launch_bounds__(kBlock) void preprocess_generic(const uint8_t* __restrict__ src_base,
                                                             OutT* __restrict__ dst_base,
                                                             PreArgs a, typename FrameArg<LIST>::type fl) {
    const int pixels = a.dst_w * a.dst_h;
    const int ox0 = blockIdx.x * (64 * kGenPx) + threadIdx.x;
    const int oy = blockIdx.y * 4 + threadIdx.y;
    if (ox0 >= a.dst_w || oy >= a.dst_h) return;
    const uint8_t* src = frame_base<LIST>(fl, a, src_base, blockIdx.z);
    OutT* dst = dst_base + (long long)blockIdx.z * a.dst_frame_stride;
    const float ny = (float)oy - a.pad_y;
    const float sy = a.fast_div ? quot3(ny, a.scale_y, a.rc_y) : ny / a.scale_y;

    float px[kGenPx][3];
#pragma unroll
    for (int j = 0; j < kGenPx; ++j) {
        const int ox = ox0 + 64 * j;
        // plan_pixel (P/preprocess.rs:437-448)
        const float nx = (float)ox - a.pad_x;
        const float sx = a.fast_div ? quot3(nx, a.scale_x, a.rc_x) : nx / a.scale_x;
        const bool inside = ox < a.dst_w &&
                            !(sx < 0.0f || sy < 0.0f || sx >= (float)a.src_w || sy >= (float)a.src_h);
        if (inside) {
            if constexpr (SAMPLER == KH_SAMPLE_NEAREST) nearest_tap<FMT, WIDE>(src, sx, sy, a, px[j]);
            else if constexpr (SAMPLER == KH_SAMPLE_BILINEAR) bilinear_quad<FMT, WIDE>(src, sx, sy, a, px[j]);
            else if constexpr (SAMPLER == kSampleBilinearOnGrid) tap_rgb<FMT, WIDE>(src, (int)sx, (int)sy, a, px[j]);   // sx, sy whole and in range (host-checked)
            else lanczos_window<FMT, WIDE>(src, sx, sy, ox, oy, a, px[j]);
        } else {
            px[j][0] = a.pad_value; px[j][1] = a.pad_value; px[j][2] = a.pad_value;
        }
    }
#pragma unroll
    for (int j = 0; j < kGenPx; ++j) {
        const int ox = ox0 + 64 * j;
        if (ox >= a.dst_w) break;
        const int i = oy * a.dst_w + ox;
        dst[i] = to_out<OutT>((div255_any(px[j][0]) - a.m0) * a.is0);
        dst[pixels + i] = to_out<OutT>((div255_any(px[j][1]) - a.m1) * a.is1);
        dst[2 * pixels + i] = to_out<OutT>((div255_any(px[j][2]) - a.m2) * a.is2);
    }
}

// ---- generic kernel, flattened destination quads (round 4) ------------------------------------------------------------------------
// profiles/r04c: the kernel above runs the 1080p -> 640 / 608 letterboxes at 4.2-4.6 TB/s with every wave slot occupied and 75-80 %
// of the wave-cycles parked on memory (SQ_WAIT_ANY): bounded by (resident waves) x (bytes a wave has in flight) / (wave lifetime),
// not by the vector ALUs (35-40 % busy) or by HBM.  Two things waste slots there: the 64 x kGenPx = 256-pixel block rows leave the
// third block of a 640- / 608-pixel row half empty (17-21 % of the lanes exit at once), and a lane keeps 4 pixels x 12 B in flight
// as twelve 4-byte stores.  Here the destination is walked as a flat list of 4-pixel quads (dst_w % 4 == 0): no idle lanes but the
// last block's tail, a lane owns one quad and writes each plane with one 16-byte streaming buffer store (a wave: 1 KiB contiguous per
// instruction), like the identity kernel.  Same per-pixel expressions, bit-identical.  (Two quads per lane: slower, r04d.)
// Byte `i` (0..15) of a 16-byte run held in four dwords.
__device__ __forceinline__ uint32_t byte16(const u32x4_t& v, int i) {
    const uint64_t lo = ((uint64_t)v.y << 32) | v.x, hi = ((uint64_t)v.w << 32) | v.z;
    return (uint32_t)((i & 8 ? hi : lo) >> (8 * (i & 7))) & 0xFFu;
}
// The one-tap NV12 samples of a destination quad from TWO 16-byte loads (luma row, chroma row) instead of eight 1- / 2-byte gathers:
// profiles/r04c has the texture addresser 80-90 % busy in the letterbox kernels (a gather instruction costs it a cycle per few lanes,
// whatever it fetches).  Valid when every quad's columns x[0] <= .. <= x[3] satisfy x[3] - (x[0] & ~1) <= 15 (host-checked for the
// launch: scale >= ~1/4) and src_w >= 16; `xs` are clamped into the row, so the run [xb, xb + 16) with xb = min(x[0] & ~1, src_w - 16)
// lies inside the luma row and the same byte range of the chroma row holds the pairs of pixels xb .. xb + 15.
__device__ __forceinline__ void quad_taps_nv12(const uint8_t* __restrict__ src, const int (&xs)[4], int y, const PreArgs& a, float (&px)[4][3]) {
    const int xb = min(xs[0] & ~1, a.src_w - 16);
    const u32x4_t yv = *reinterpret_cast<const u32x4_unaligned*>(src + (unsigned)(y * a.src_w + xb));
    const u32x4_t cv = *reinterpret_cast<const u32x4_unaligned*>(src + (unsigned)(a.src_w * a.src_h + (y >> 1) * a.src_w + xb));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = xs[j] - xb, ci = i & ~1;
        bt601_q20_to_rgb((int)byte16(yv, i), (int)byte16(cv, ci), (int)byte16(cv, ci + 1), px[j]);
    }
}

// YUYV (Y0 U Y1 V per pixel pair): the quad's one-tap samples from one 32-byte run (two 16-byte loads).  Pixel xb + i: Y = byte 2 i,
// U = byte 4 (i >> 1) + 1, V = byte 4 (i >> 1) + 3.  Same precondition as quad_taps_nv12 (x[3] - (x[0] & ~1) <= 15, src_w >= 16).
__device__ __forceinline__ uint32_t byte32(const u32x4_t& lo, const u32x4_t& hi, int k) {
    const u32x4_t v = (k & 16) ? hi : lo;
    return byte16(v, k & 15);
}
__device__ __forceinline__ void quad_taps_yuyv(const uint8_t* __restrict__ src, const int (&xs)[4], int y, const PreArgs& a, float (&px)[4][3]) {
    const int xb = min(xs[0] & ~1, a.src_w - 16);
    const uint8_t* p = src + (unsigned)(y * a.src_pitch + 2 * xb);
    const u32x4_t lo = *reinterpret_cast<const u32x4_unaligned*>(p), hi = *reinterpret_cast<const u32x4_unaligned*>(p + 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = xs[j] - xb, g = 4 * (i >> 1);
        bt601_q20_to_rgb((int)byte32(lo, hi, 2 * i), (int)byte32(lo, hi, g + 1), (int)byte32(lo, hi, g + 3), px[j]);
    }
}

// The same for the FOUR-tap bilinear sampler: the quad's taps x0[0] .. x0[3] + 1 of rows y0 and y1 = min(y0 + 1, h - 1) come from four
// 16-byte loads (two luma rows, their chroma rows) instead of sixteen gathers; tap (x, y) decodes Y[y][x] with the chroma pair
// (y >> 1, x >> 1), which is what fetch_pair's parity cases amount to, and the blend is bilinear_quad's expression.
__device__ __forceinline__ void quad_taps_nv12_bilinear(const uint8_t* __restrict__ src, const float (&sxs)[4], float sy, const PreArgs& a,
                                                        float (&px)[4][3]) {
    const int y0 = min(max((int)floorf(sy), 0), a.src_h - 1), y1 = min(y0 + 1, a.src_h - 1);
    const float ay = sy - (float)(int)floorf(sy);
    int x0[4], x1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x0[j] = min(max((int)floorf(sxs[j]), 0), a.src_w - 1);
        x1[j] = min(x0[j] + 1, a.src_w - 1);
    }
    const int xb = min(x0[0] & ~1, a.src_w - 16);
    const unsigned cbase = (unsigned)(a.src_w * a.src_h + xb);
    const u32x4_t yv0 = *reinterpret_cast<const u32x4_unaligned*>(src + (unsigned)(y0 * a.src_w + xb));
    const u32x4_t yv1 = *reinterpret_cast<const u32x4_unaligned*>(src + (unsigned)(y1 * a.src_w + xb));
    const u32x4_t cv0 = *reinterpret_cast<const u32x4_unaligned*>(src + cbase + (unsigned)((y0 >> 1) * a.src_w));
    const u32x4_t cv1 = *reinterpret_cast<const u32x4_unaligned*>(src + cbase + (unsigned)((y1 >> 1) * a.src_w));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i0 = x0[j] - xb, i1 = x1[j] - xb, c0 = i0 & ~1, c1 = i1 & ~1;
        const float ax = sxs[j] - (float)(int)floorf(sxs[j]);
        float t00[3], t10[3], t01[3], t11[3];
        bt601_q20_to_rgb((int)byte16(yv0, i0), (int)byte16(cv0, c0), (int)byte16(cv0, c0 + 1), t00);
        bt601_q20_to_rgb((int)byte16(yv0, i1), (int)byte16(cv0, c1), (int)byte16(cv0, c1 + 1), t10);
        bt601_q20_to_rgb((int)byte16(yv1, i0), (int)byte16(cv1, c0), (int)byte16(cv1, c0 + 1), t01);
        bt601_q20_to_rgb((int)byte16(yv1, i1), (int)byte16(cv1, c1), (int)byte16(cv1, c1 + 1), t11);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = t00[c] + (t10[c] - t00[c]) * ax;
            const float bot = t01[c] + (t11[c] - t01[c]) * ax;
            px[j][c] = top + (bot - top) * ay;
        }
    }
}

// YUYV twin (round 6): the sixteen taps of a quad from four 16-byte loads (two per row: pixels xb .. xb + 15 are 32 bytes) instead of
// per-tap gathers; tap (x, y) = Y byte 2 i, chroma bytes 4 (i >> 1) + 1 / + 3 of its own row, the blend bilinear_quad's expression.
__device__ __forceinline__ void quad_taps_yuyv_bilinear(const uint8_t* __restrict__ src, const float (&sxs)[4], float sy, const PreArgs& a,
                                                        float (&px)[4][3]) {
    const int y0 = min(max((int)floorf(sy), 0), a.src_h - 1), y1 = min(y0 + 1, a.src_h - 1);
    const float ay = sy - (float)(int)floorf(sy);
    int x0[4], x1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x0[j] = min(max((int)floorf(sxs[j]), 0), a.src_w - 1);
        x1[j] = min(x0[j] + 1, a.src_w - 1);
    }
    const int xb = min(x0[0] & ~1, a.src_w - 16);
    const uint8_t* p0 = src + (unsigned)(y0 * a.src_pitch + 2 * xb);
    const uint8_t* p1 = src + (unsigned)(y1 * a.src_pitch + 2 * xb);
    const u32x4_t lo0 = *reinterpret_cast<const u32x4_unaligned*>(p0), hi0 = *reinterpret_cast<const u32x4_unaligned*>(p0 + 16);
    const u32x4_t lo1 = *reinterpret_cast<const u32x4_unaligned*>(p1), hi1 = *reinterpret_cast<const u32x4_unaligned*>(p1 + 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i0 = x0[j] - xb, i1 = x1[j] - xb, g0 = 4 * (i0 >> 1), g1 = 4 * (i1 >> 1);
        const float ax = sxs[j] - (float)(int)floorf(sxs[j]);
        float t00[3], t10[3], t01[3], t11[3];
        bt601_q20_to_rgb((int)byte32(lo0, hi0, 2 * i0), (int)byte32(lo0, hi0, g0 + 1), (int)byte32(lo0, hi0, g0 + 3), t00);
        bt601_q20_to_rgb((int)byte32(lo0, hi0, 2 * i1), (int)byte32(lo0, hi0, g1 + 1), (int)byte32(lo0, hi0, g1 + 3), t10);
        bt601_q20_to_rgb((int)byte32(lo1, hi1, 2 * i0), (int)byte32(lo1, hi1, g0 + 1), (int)byte32(lo1, hi1, g0 + 3), t01);
        bt601_q20_to_rgb((int)byte32(lo1, hi1, 2 * i1), (int)byte32(lo1, hi1, g1 + 1), (int)byte32(lo1, hi1, g1 + 3), t11);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = t00[c] + (t10[c] - t00[c]) * ax;
            const float bot = t01[c] + (t11[c] - t01[c]) * ax;
            px[j][c] = top + (bot - top) * ay;
        }
    }
}

// (Round 5: picking the 48 bytes of a four-tap quad through a lane-private LDS slot — ds_read_u8 at run-time addresses instead of
// 64-bit shifts and selects — takes 15 % of the kernel's vector instructions away, 682 M -> 577 M per launch, and not a microsecond:
// 1.295 ms both ways (profiles/r05b_four_tap_lds_picks_ab.txt, r05c_*_counters.csv).  Decoding one tap per pixel instead of four
// (a diagnostic build: half of the arithmetic gone, every load and store kept) runs 1.21 ms: the row is bound by the memory system,
// 6 % above its floor (profiles/r05j_four_tap_decode_ablation.txt).  Not kept.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kQuadBlock = 256;
// A quad's four values of plane `c`: one 16-byte store of f32, or (round 6) one 8-byte store of four binary16 values (f2h_bits: the
// reference's rounding) — the f16 outputs took the per-pixel kernel with 2-byte stores before and ran no faster than f32 for half the bytes.
template <typename OutT>
__device__ __forceinline__ void store_quad(__amdgpu_buffer_rsrc_t rdst, int g, int c, int plane, const float (&v)[4], int f16_plain) {
    if constexpr (sizeof(OutT) == 4) {
        __builtin_amdgcn_raw_buffer_store_b128((u32x4_t{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}), rdst,
                                               16 * g + c * (4 * plane), 0, kAuxStream);
    } else {
        uint32_t lo, hi;
        if (f16_plain) { lo = pack_h2(v[0], v[1]); hi = pack_h2(v[2], v[3]); }   // launch-uniform (f16_in_range on the host)
        else { lo = (uint32_t)f2h_bits(v[0]) | ((uint32_t)f2h_bits(v[1]) << 16); hi = (uint32_t)f2h_bits(v[2]) | ((uint32_t)f2h_bits(v[3]) << 16); }
        __builtin_amdgcn_raw_buffer_store_b64((u32x2_t{lo, hi}), rdst, 8 * g + c * (2 * plane), 0, kAuxStream);
    }
}
template <int FMT, int SAMPLER, bool WIDE, bool LIST, typename OutT>
__global__ __launch_bounds__(kQuadBlock) void preprocess_generic_quads(const uint8_t* __restrict__ src_base, OutT* __restrict__ dst_base,
                                                                       PreArgs a, FastDiv by_wq, typename FrameArg<LIST>::type fl) {
    const int wq = a.dst_w >> 2, groups = wq * a.dst_h, plane = a.dst_w * a.dst_h;   // host-checked: 12 * plane < 2^31
    const int g = blockIdx.x * kQuadBlock + threadIdx.x;
    if (g >= groups) return;
    const uint8_t* src = frame_base<LIST>(fl, a, src_base, blockIdx.y);
    const __amdgpu_buffer_rsrc_t rdst = buffer_rsrc(dst_base + (long long)blockIdx.y * a.dst_frame_stride, (uint32_t)(3 * (int)sizeof(OutT) * plane));
    const int oy = (int)fast_quot((uint32_t)g, by_wq), ox0 = 4 * (g - oy * wq);
    const float ny = (float)oy - a.pad_y;
    const float sy = a.fast_div ? quot3(ny, a.scale_y, a.rc_y) : ny / a.scale_y;
    float o[3][4];
    if constexpr ((FMT == KH_FMT_NV12 && (SAMPLER == kSampleBilinearOnGrid || SAMPLER == KH_SAMPLE_BILINEAR || SAMPLER == KH_SAMPLE_NEAREST)) ||
                  (FMT == KH_FMT_YUYV && (SAMPLER == kSampleBilinearOnGrid || SAMPLER == KH_SAMPLE_NEAREST || SAMPLER == KH_SAMPLE_BILINEAR))) {
        if (a.quad_wide) {   // uniform
            float sxs[4], px[4][3];
            int xs[4];
            bool in[4], any = false;
            const bool row_in = !(sy < 0.0f || sy >= (float)a.src_h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float nx = (float)(ox0 + j) - a.pad_x;
                sxs[j] = a.fast_div ? quot3(nx, a.scale_x, a.rc_x) : nx / a.scale_x;
                in[j] = row_in && !(sxs[j] < 0.0f || sxs[j] >= (float)a.src_w);
                any = any || in[j];
                // nearest (round 6): nearest_tap's column, round half away from zero, instead of the truncated one
                xs[j] = min(max(SAMPLER == KH_SAMPLE_NEAREST ? (int)roundf(sxs[j]) : (int)sxs[j], 0), a.src_w - 1);
            }
            if (any) {
                const int yn = min(max(SAMPLER == KH_SAMPLE_NEAREST ? (int)roundf(sy) : (int)sy, 0), a.src_h - 1);
                if constexpr (FMT == KH_FMT_YUYV && SAMPLER == KH_SAMPLE_BILINEAR) quad_taps_yuyv_bilinear(src, sxs, sy, a, px);
                else if constexpr (FMT == KH_FMT_YUYV) quad_taps_yuyv(src, xs, yn, a, px);
                else if constexpr (SAMPLER == KH_SAMPLE_BILINEAR) quad_taps_nv12_bilinear(src, sxs, sy, a, px);
                else quad_taps_nv12(src, xs, yn, a, px);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int c = 0; c < 3; ++c) px[j][c] = in[j] ? px[j][c] : a.pad_value;
                o[0][j] = (div255_any(px[j][0]) - a.m0) * a.is0;
                o[1][j] = (div255_any(px[j][1]) - a.m1) * a.is1;
                o[2][j] = (div255_any(px[j][2]) - a.m2) * a.is2;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) store_quad<OutT>(rdst, g, c, plane, o[c], a.f16_plain);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // plan_pixel (P/preprocess.rs:437-448)
        const float nx = (float)(ox0 + j) - a.pad_x;
        const float sx = a.fast_div ? quot3(nx, a.scale_x, a.rc_x) : nx / a.scale_x;
        float px[3];
        if (!(sx < 0.0f || sy < 0.0f || sx >= (float)a.src_w || sy >= (float)a.src_h)) {
            if constexpr (SAMPLER == KH_SAMPLE_NEAREST) nearest_tap<FMT, WIDE>(src, sx, sy, a, px);
            else if constexpr (SAMPLER == KH_SAMPLE_BILINEAR) bilinear_quad<FMT, WIDE>(src, sx, sy, a, px);
            else tap_rgb<FMT, WIDE>(src, (int)sx, (int)sy, a, px);   // kSampleBilinearOnGrid: sx, sy whole and in range (host-checked)
        } else {
            px[0] = a.pad_value; px[1] = a.pad_value; px[2] = a.pad_value;
        }
        o[0][j] = (div255_any(px[0]) - a.m0) * a.is0;
        o[1][j] = (div255_any(px[1]) - a.m1) * a.is1;
        o[2][j] = (div255_any(px[2]) - a.m2) * a.is2;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) store_quad<OutT>(rdst, g, c, plane, o[c], a.f16_plain);  // plane offset in the vector offset, never in an SGPR soffset (kh_common.h)
}

// ---- north-star fast path ------------------------------------------------------------------

template <bool NT>
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
}

// x / 255.0f for x an integer in [0, 255], without the ~10-instruction IEEE division sequence:
// q = x * (1/255), one fma residual, one fma correction.  Exhaustively equal to the correctly
// rounded quotient for all 256 inputs (tests/test_host_math.py proves it on the host with the
// same IEEE fma; the GPU parity tests sweep every byte value through this path).
__device__ __forceinline__ float div255_u8(float x) {
    const float rc = 1.0f / 255.0f;
    const float q = x * rc;
    const float r = __builtin_fmaf(-q, 255.0f, x);
    return __builtin_fmaf(r, rc, q);
}

// One thread = 4 pixels of ONE row, all three planes: a wave writes three 1 KiB-contiguous segments (3 store streams; the 4x2
// variant's 6 streams measured ~8 % slower, profiles/r01b_ubench_nv12.txt; two rows per thread with write-through stores 12 %
// slower, profiles/r02f_ubench_nv12.txt).  The chroma dword is read by both rows of a pair; the second read is an L2 /
// Infinity-Cache hit (putting the two readers on one XCD changes nothing: r02g `adj` variants).
//
// Round 2 (profiles/r02a..r02l_ubench_nv12.txt; every variant bit-identical to this kernel, same box, interleaved):
//   * loads and stores go through the buffer path, the stores write-through + non-temporal (kAuxStream, kh_common.h): -4..-6 % time
//     against the non-temporal global stores of round 1;
//   * all twelve results are computed BEFORE the first store (the empty asm below pins that), so the three plane stores issue back to
//     back instead of being interleaved with the decode by the scheduler: another -3.5 %;
//   * the row index comes from a plain integer division and BOTH loads depend on it.  Replacing it with the 2-instruction multiply-shift
//     used elsewhere in this library makes the kernel 3.5 % SLOWER (4.42 vs 4.27 ms), and a sleep of the same length does not give the
//     time back: WHEN and how densely a wave touches memory matters more than its instruction count here.
// Together: 4.27-4.30 ms per 1024 frames against 4.69-4.74 ms for the round-1 kernel on the same boxes (6.7 TB/s, 0.84 of 8 TB/s).
// Measured and NOT used: staging the source through LDS with 16-byte loads (block-wide K rounds -1..-17 %, wave-private -12 %), K
// chunks per thread with hoisted loads (+2 % at best, -12 % typical, -18 % with back-to-back stores), persistent block-stride loops
// (-25..-48 %), one plane per wave through an LDS transpose (-19 %), one store per thread with the planes on different waves
// (-33 %), two rows per thread (-12 %), XCD-per-frame block order (-4 %), pairing chroma-sharing chunks on one XCD (0 %),
// frame-stride / base-offset padding (+-0.5 %), non-temporal LOADS (-16 %), s_setprio before the stores (-2 %), sleeps before the
// loads or the stores (0..-6 %), limiting occupancy through LDS (0..-25 %), other block sizes (256 -13 %, 384 -4 %, 448 -3 %, 576 -7 %,
// 640 -12 %, 1024 -18 %).  Ceilings on the same boxes: these three plane stores with no loads or decode 4.04 ms, a flat fill of the
// same bytes 3.55-3.63 ms.
constexpr int kIdBlock = 512;  // 8 KiB contiguous per plane per block; 256 / 384 / 640 / 768 / 1024 are 2-11 % slower (r02e, r02f)
// (An XCD-per-frame block order — XCD k walks frames k, k + 8, ... — measured 4.84 ms against 4.65 ms for this order in round 2,
// profiles/r02a_ab.log, and is not in the library.)
__device__ __forceinline__ void nv12_identity_body(const uint8_t* __restrict__ src_frame, float* __restrict__ dst_frame, const PreArgs& a) {
    const int wq = a.src_w >> 2;     // 4-pixel groups per row
    const int groups = wq * a.src_h;
    const unsigned chunk = blockIdx.x;
    const int g = chunk * kIdBlock + threadIdx.x;
    if (g >= groups) return;
    const int plane = a.src_w * a.src_h;           // host-checked: 12 * plane < 2^31
    // one V# per frame: source = plane * 3 / 2 bytes, destination = 3 planes of f32
    const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(src_frame, (uint32_t)(plane + plane / 2));
    const __amdgpu_buffer_rsrc_t rdst = buffer_rsrc(dst_frame, (uint32_t)(12 * plane));

    const int r = g / wq;  // deliberately NOT fast_quot(g, by_wq): see the header comment (the division paces the loads)
    const int xq = g - r * wq;
    const uint32_t y4 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, r * a.src_w + 4 * xq, 0, 0);
    const uint32_t uv4 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, plane + (r >> 1) * a.src_w + 4 * xq, 0, 0);

    // Chroma terms shared by each pixel pair; integer adds are exact, so hoisting the rounding
    // constant gives the same value as (yy + c*u + half).
    int tb[2], tg[2], tr[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = (int)((uv4 >> (16 * k)) & 0xFFu) - 128;
        const int v = (int)((uv4 >> (16 * k + 8)) & 0xFFu) - 128;
        tb[k] = kCUB * u + kHalf20;
        tg[k] = kCUG * u + kCVG * v + kHalf20;
        tr[k] = kCVR * v + kHalf20;
    }

    f32x4 o[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = max((int)((y4 >> (8 * j)) & 0xFFu) - 16, 0) * kCY;
        const int k = j >> 1;
        const float rr = (float)clamp255((yy + tr[k]) >> 20);
        const float gg = (float)clamp255((yy + tg[k]) >> 20);
        const float bb = (float)clamp255((yy + tb[k]) >> 20);
        o[0][j] = (div255_u8(rr) - a.m0) * a.is0;
        o[1][j] = (div255_u8(gg) - a.m1) * a.is1;
        o[2][j] = (div255_u8(bb) - a.m2) * a.is2;
    }
    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]));  // all twelve values exist here: the stores below issue back to back
#pragma unroll
    for (int c = 0; c < 3; ++c)  // plane offset in the vector offset, never in an SGPR soffset (kh_common.h)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o[c]), rdst, 16 * g + c * (4 * plane), 0, kAuxStream);
}
__global__ __launch_bounds__(kIdBlock) void preprocess_nv12_identity(
    const uint8_t* __restrict__ src_base, float* __restrict__ dst_base, PreArgs a) {
    const unsigned frame = blockIdx.y;
    nv12_identity_body(src_base + (long long)frame * a.src_frame_stride, dst_base + (long long)frame * a.dst_frame_stride, a);
}
// the same kernel for separately allocated frames (kh_preprocess_to_chw_list): the frame's base is one scalar load from the kernel arguments
__global__ __launch_bounds__(kIdBlock) void preprocess_nv12_identity_list(FrameList fl, float* __restrict__ dst_base, PreArgs a) {
    const unsigned frame = blockIdx.y;
    nv12_identity_body(fl.p[frame], dst_base + (long long)frame * a.dst_frame_stride, a);
}

// ---- the same case into binary16 planes (round 6; run_raw_batch_f16, P/preprocess.rs:1234-1256) -------------------------------------
// The f16 twin took the generic kernel until round 6 and ran SLOWER than the f32 north star (6.11 ms against 4.31 per 1024 frames for
// 55 % of the bytes).  Here a thread owns EIGHT pixels of one row — two dwords of luma, two of chroma — so that each of its three stores
// is still 16 bytes (eight binary16 values; a wave writes 1 KiB contiguous per plane): the f32 kernel's store shape at half the output
// bytes.  Same integer decode, same `(x / 255 - m) * is`, then the reference's f32 -> f16 rounding: 2.75 ms (profiles/r06ze_f16_identity.txt;
// 64-bit source loads instead of pairs of dwords: 2.73, not taken).
// LUT (round 6): the kernel above it is bound by its vector ALUs as much as by memory (250 instructions per thread; the f32 kernel's
// store stream is half as long here).  `(x / 255 - m_c) * is_c` rounded to binary16 depends on the clamped 8-bit channel value only: each
// block builds the 3 x 256 table once in LDS with the very expressions it replaces (two entries per thread of the first six waves), and a
// channel value then costs its integer decode and ONE ds_read_u16 instead of nine float instructions.
template <bool LUT>
__device__ __forceinline__ void nv12_identity_f16_body(const uint8_t* __restrict__ src_frame, unsigned short* __restrict__ dst_frame, const PreArgs& a) {
    __shared__ __attribute__((aligned(4))) unsigned short lut[LUT ? 768 : 2];
    const int wo = a.src_w >> 3;     // 8-pixel groups per row
    const int groups = wo * a.src_h;
    const int g0 = blockIdx.x * kIdBlock + threadIdx.x;
    if (!LUT && g0 >= groups) return;
    const int g = LUT ? min(g0, groups - 1) : g0;   // (LUT: every thread reaches the barrier; the tail's stores are dropped below)
    const int plane = a.src_w * a.src_h;
    const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(src_frame, (uint32_t)(plane + plane / 2));
    const __amdgpu_buffer_rsrc_t rdst = buffer_rsrc(dst_frame, (uint32_t)(6 * plane));
    const int r = g / wo;            // (a plain division, as in the f32 kernel)
    const int xo = g - r * wo;
    uint32_t y4[2], uv4[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        y4[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, r * a.src_w + 8 * xo + 4 * q, 0, 0);
        uv4[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, plane + (r >> 1) * a.src_w + 8 * xo + 4 * q, 0, 0);
    }
    if constexpr (LUT) {   // after the loads were issued: the table does not depend on them
        const int t = threadIdx.x;
        if (t < 384) {
            const int c = t >> 7, x = 2 * (t & 127);
            const float m = c == 0 ? a.m0 : (c == 1 ? a.m1 : a.m2), is = c == 0 ? a.is0 : (c == 1 ? a.is1 : a.is2);
            reinterpret_cast<uint32_t*>(lut)[t] = pack_h2((div255_u8((float)x) - m) * is, (div255_u8((float)(x + 1)) - m) * is);
        }
        __syncthreads();
    }
    u32x4_t o[3];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int tb[2], tg[2], tr[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = (int)((uv4[q] >> (16 * k)) & 0xFFu) - 128;
            const int v = (int)((uv4[q] >> (16 * k + 8)) & 0xFFu) - 128;
            tb[k] = kCUB * u + kHalf20;
            tg[k] = kCUG * u + kCVG * v + kHalf20;
            tr[k] = kCVR * v + kHalf20;
        }
        if constexpr (LUT) {
            unsigned short hv[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int yy = max((int)((y4[q] >> (8 * j)) & 0xFFu) - 16, 0) * kCY;
                const int k = j >> 1;
                hv[0][j] = lut[clamp255((yy + tr[k]) >> 20)];
                hv[1][j] = lut[256 + clamp255((yy + tg[k]) >> 20)];
                hv[2][j] = lut[512 + clamp255((yy + tb[k]) >> 20)];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                o[c][2 * q] = (uint32_t)hv[c][0] | ((uint32_t)hv[c][1] << 16);
                o[c][2 * q + 1] = (uint32_t)hv[c][2] | ((uint32_t)hv[c][3] << 16);
            }
            continue;
        }
        float f[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = max((int)((y4[q] >> (8 * j)) & 0xFFu) - 16, 0) * kCY;
            const int k = j >> 1;
            const float rr = (float)clamp255((yy + tr[k]) >> 20);
            const float gg = (float)clamp255((yy + tg[k]) >> 20);
            const float bb = (float)clamp255((yy + tb[k]) >> 20);
            f[0][j] = (div255_u8(rr) - a.m0) * a.is0;
            f[1][j] = (div255_u8(gg) - a.m1) * a.is1;
            f[2][j] = (div255_u8(bb) - a.m2) * a.is2;
        }
        // f2h_bits without its >= 2^16 branch: the host proved that no value of this launch reaches it (identity_fast_path), and below
        // it the reference's rounding IS the round-to-nearest-even conversion instruction
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c][2 * q] = pack_h2(f[c][0], f[c][1]);
            o[c][2 * q + 1] = pack_h2(f[c][2], f[c][3]);
        }
    }
    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]));  // all twenty-four values exist here: the stores issue back to back
#pragma unroll
    for (int c = 0; c < 3; ++c) __builtin_amdgcn_raw_buffer_store_b128(o[c], rdst, g0 < groups ? 16 * g + c * (2 * plane) : 0x7ffffff0, 0, kAuxStream);   // (past the V#'s range: dropped)
}
template <bool LUT>
__global__ __launch_bounds__(kIdBlock) void preprocess_nv12_identity_f16(
    const uint8_t* __restrict__ src_base, unsigned short* __restrict__ dst_base, PreArgs a) {
    const unsigned frame = blockIdx.y;
    nv12_identity_f16_body<LUT>(src_base + (long long)frame * a.src_frame_stride, dst_base + (long long)frame * a.dst_frame_stride, a);
}
template <bool LUT>
__global__ __launch_bounds__(kIdBlock) void preprocess_nv12_identity_f16_list(FrameList fl, unsigned short* __restrict__ dst_base, PreArgs a) {
    const unsigned frame = blockIdx.y;
    nv12_identity_f16_body<LUT>(fl.p[frame], dst_base + (long long)frame * a.dst_frame_stride, a);
}

// Lanczos axis-weight tables: wx[dst_w][6] then wy[dst_h][6], cached per (device, geometry) like the reference's tap tables
// (P/resize/cuda.rs:151-190): blocking upload before publication, leased until the launch is recorded, a cold table under stream
// capture is a typed error (kh_table_cache.h).  `s` is evaluated exactly as the kernel's plan_pixel does ((o - pad) / scale, IEEE).
using LzKey = std::tuple<int, uint32_t, uint32_t, uint32_t, uint32_t, int, int>;
TableCache<LzKey>& g_lz_tabs = *new TableCache<LzKey>(64);  // never destroyed (the runtime may be gone at static destruction)

int32_t lanczos_tables(PreArgs& a, hipStream_t stream, TableLease& lease) {
    int dev = 0;
    KH_HIP(hipGetDevice(&dev));
    auto bits = [](float f) { return __builtin_bit_cast(uint32_t, f); };
    const LzKey key{dev, bits(a.scale_x), bits(a.pad_x), bits(a.scale_y), bits(a.pad_y), a.dst_w, a.dst_h};
    const int32_t rc = g_lz_tabs.lookup(key, stream, "preprocess (Lanczos weights)", [&](DevTable& t) -> int32_t {
        std::vector<float> w((size_t)6 * (a.dst_w + a.dst_h));
        auto axis = [](float* out, int n, float pad, float scale) {
            for (int o = 0; o < n; ++o) {
                const float s = ((float)o - pad) / scale;
                const int i0 = (int)floorf(s);
                for (int i = 0; i < 6; ++i) out[6 * o + i] = lanczos_w(s - (float)(i0 - 2 + i));
            }
        };
        axis(w.data(), a.dst_w, a.pad_x, a.scale_x);
        axis(w.data() + (size_t)6 * a.dst_w, a.dst_h, a.pad_y, a.scale_y);
        t.bytes = w.size() * sizeof(float);
        t.meta[0] = a.dst_w;
        KH_HIP(hipMalloc(&t.dev, t.bytes));
        const hipError_t err = hipMemcpy(t.dev, w.data(), t.bytes, hipMemcpyHostToDevice);
        if (err != hipSuccess) return fail_hip(err, "hipMemcpy (Lanczos weight table)");  // ~DevTable frees the allocation
        return KH_OK;
    }, lease);
    if (rc != KH_OK) return rc;
    a.lz_wx = static_cast<const float*>(lease->dev);
    a.lz_wy = a.lz_wx + (size_t)6 * a.dst_w;
    return KH_OK;
}

// Does quot3 reproduce IEEE division for every (o - pad) / scale the launch will evaluate?  dst_w +
// dst_h host evaluations (microseconds), memoised on the last geometry.
bool plan_division_is_exact(const PreArgs& a) {
    if (dev_opt(kOptPreIeeeDiv) == 1) return false;  // test option: always divide
    struct Key { float sx, sy, px, py; int w, h; bool ok; };
    static thread_local Key last = {0, 0, 0, 0, 0, 0, false};
    if (last.w == a.dst_w && last.h == a.dst_h && last.sx == a.scale_x && last.sy == a.scale_y && last.px == a.pad_x &&
        last.py == a.pad_y)
        return last.ok;
    auto same = [](float u, float v) { return __builtin_bit_cast(uint32_t, u) == __builtin_bit_cast(uint32_t, v); };
    bool ok = a.scale_x != 0.0f && a.scale_y != 0.0f;
    for (int o = 0; ok && o < a.dst_w; ++o) {
        const float n = (float)o - a.pad_x;
        ok = same(quot3(n, a.scale_x, a.rc_x), n / a.scale_x);
    }
    for (int o = 0; ok && o < a.dst_h; ++o) {
        const float n = (float)o - a.pad_y;
        ok = same(quot3(n, a.scale_y, a.rc_y), n / a.scale_y);
    }
    last = Key{a.scale_x, a.scale_y, a.pad_x, a.pad_y, a.dst_w, a.dst_h, ok};
    return ok;
}

// Do ALL the source coordinates this launch evaluates fall on whole pixels?  (1080p -> a 640-wide letterbox: scale 1/3, sx = 3 ox
// exactly, in f32.)  Then both bilinear weights are 0 for every destination pixel and the reference's blend
// t00 + (t10 - t00) * 0 ... returns the first tap bit for bit: the kernel decodes ONE tap per pixel instead of four — the generic
// bilinear path is bound by its vector ALUs (four exact BT.601 decodes + blend: ~103 instructions per pixel, r02t), not by memory.
// Decided by evaluating the kernel's own coordinate expression for every column and row (dst_w + dst_h host evaluations,
// memoised on the last geometry) — not by looking at the scale.  kh_debug_set_option("pre_grid", 0) (test option) keeps the four-tap kernel.
bool bilinear_taps_on_grid(const PreArgs& a) {
    if (dev_opt(kOptPreGrid) == 0) return false;
    struct Key { float sx, sy, px, py; int w, h, sw, sh; bool ok; };
    static thread_local Key last = {0, 0, 0, 0, 0, 0, 0, 0, false};
    if (last.w == a.dst_w && last.h == a.dst_h && last.sw == a.src_w && last.sh == a.src_h && last.sx == a.scale_x && last.sy == a.scale_y &&
        last.px == a.pad_x && last.py == a.pad_y)
        return last.ok;
    auto axis = [](int n, float pad, float scale, int len) {
        for (int o = 0; o < n; ++o) {
            const float s = ((float)o - pad) / scale;   // plan_pixel; the kernel's quotient equals this one or takes the division itself
            if (s < 0.0f || s >= (float)len) continue;  // padding: no tap
            if (!(s == floorf(s))) return false;        // (also false for NaN)
        }
        return true;
    };
    const bool ok = a.scale_x != 0.0f && a.scale_y != 0.0f && axis(a.dst_w, a.pad_x, a.scale_x, a.src_w) && axis(a.dst_h, a.pad_y, a.scale_y, a.src_h);
    last = Key{a.scale_x, a.scale_y, a.pad_x, a.pad_y, a.dst_w, a.dst_h, a.src_w, a.src_h, ok};
    return ok;
}

// quad_taps_nv12's precondition, decided like the other launch checks by evaluating the kernel's own column expression for every
// destination quad (dst_w / 4 groups of four host evaluations, memoised on the last geometry).
bool quad_taps_fit_16(const PreArgs& a, int extra) {   // extra = 1: the bilinear sampler also reads column x + 1; 2: nearest (rounded columns)
    struct Key { float sx, px; int w, sw, extra; bool ok; };
    static thread_local Key last = {0, 0, 0, 0, 0, false};
    if (last.w == a.dst_w && last.sw == a.src_w && last.sx == a.scale_x && last.px == a.pad_x && last.extra == extra) return last.ok;
    bool ok = a.src_w >= 16 && a.dst_w % 4 == 0 && a.scale_x != 0.0f;
    for (int q = 0; ok && q < a.dst_w / 4; ++q) {
        int x[4];
        for (int j = 0; j < 4; ++j) {
            const float s = ((float)(4 * q + j) - a.pad_x) / a.scale_x;   // plan_pixel; the kernel's quotient equals this or it divides itself
            // the float -> int cast is defined only inside the int range: NaN and anything below it are sorted out BEFORE the cast
            if (!(s == s)) { ok = false; x[j] = 0; }
            else if (extra == 2) { const float r = roundf(s); x[j] = r >= 2147483520.0f ? a.src_w - 1 : r <= -2147483648.0f ? 0 : std::min(std::max((int)r, 0), a.src_w - 1); }
            else x[j] = s >= 2147483520.0f ? a.src_w - 1 : s <= -2147483648.0f ? 0 : std::min(std::max((int)s, 0), a.src_w - 1);
        }
        ok = ok && x[0] <= x[1] && x[1] <= x[2] && x[2] <= x[3] && std::min(x[3] + (extra == 1 ? 1 : 0), a.src_w - 1) - (x[0] & ~1) <= 15;
    }
    last = Key{a.scale_x, a.pad_x, a.dst_w, a.src_w, extra, ok};
    return ok;
}

// The source frames of a call: n frames `stride` bytes apart from `src`, or — list != nullptr — at list[k] (host array of device pointers).
struct Frames {
    const uint8_t* src; int64_t stride; const uint8_t* const* list; int n;
    bool listed() const { return list != nullptr; }
    bool aligned(unsigned align) const {   // every frame base a multiple of `align` bytes
        if (!listed()) return reinterpret_cast<uintptr_t>(src) % align == 0 && (n <= 1 || stride % align == 0);
        for (int k = 0; k < n; ++k) if (reinterpret_cast<uintptr_t>(list[k]) % align) return false;
        return true;
    }
};

// The f16 fast paths convert with the plain round-to-nearest-even instruction.  The reference's f2h differs from it only at |v| >= 2^16
// (NaN patterns instead of Inf, see f2h_bits), and v = (q - m) * is is monotone in q (subtraction and multiplication by a constant round
// monotonically), so its largest magnitude is at the ends of q's range: q = x / 255 of a decoded or bilinearly blended sample lies in
// [0, 1] up to the blend's rounding (evaluated here at -0.01 and 1.01), and the padding value contributes pad_value / 255.  All below
// 2^16 for every channel, or the full f2h_bits runs.  (False for NaN / Inf parameters.  Not for Lanczos, whose samples overshoot.)
bool f16_in_range(const kh_preprocess_params* p) {
    for (int c = 0; c < 3; ++c) {
        const float qs[3] = {-0.01f, 1.01f, p->pad_value / 255.0f};
        for (float q : qs) {
            const float v = (q - p->mean[c]) * p->inv_std[c];
            if (!(fabsf(v) < 65536.0f)) return false;
        }
    }
    return true;
}

bool identity_fast_path(const kh_preprocess_params* p, const Frames& f, const void* dst) {
    const bool f16 = p->out_dtype == KH_OUT_F16;   // eight pixels per thread, eight halves per 16-byte store; plain conversions
    if (f16 && !f16_in_range(p)) return false;
    return !(p->flags & KH_PRE_FORCE_GENERIC) && p->fmt == KH_FMT_NV12 &&
           (p->out_dtype == KH_OUT_F32 || f16) &&
           (p->sampling == KH_SAMPLE_BILINEAR || p->sampling == KH_SAMPLE_NEAREST) &&
           p->scale_x == 1.0f && p->scale_y == 1.0f && p->pad_x == 0.0f && p->pad_y == 0.0f &&
           p->dst_w == p->src_w && p->dst_h == p->src_h && (p->src_w % (f16 ? 8 : 4)) == 0 &&
           (int64_t)p->src_w * p->src_h * 12 <= kI32Max &&  // 32-bit buffer offsets within one frame's three planes
           f.aligned(4) && (f.listed() || (p->src_frame_stride % 4) == 0) &&
           (reinterpret_cast<uintptr_t>(dst) % 16) == 0 && (p->dst_frame_stride % (f16 ? 8 : 4)) == 0;
}

int32_t validate(const kh_preprocess_params* p, const Frames& f, const void* dst) {
    KH_REQUIRE(p, KH_ERR_INVALID_ARG, "preprocess: null params");
    KH_REQUIRE(p->nframes >= 0, KH_ERR_INVALID_ARG, "preprocess: negative frame count");
    KH_REQUIRE(p->src_w > 0 && p->src_h > 0 && p->dst_w > 0 && p->dst_h > 0, KH_ERR_INVALID_ARG,
               "preprocess: zero-sized image (src %dx%d, dst %dx%d)", p->src_w, p->src_h, p->dst_w,
               p->dst_h);
    KH_REQUIRE(p->fmt >= KH_FMT_RGB && p->fmt <= KH_FMT_YUYV, KH_ERR_INVALID_ARG,
               "preprocess: unknown source format %d", p->fmt);
    KH_REQUIRE(p->sampling >= KH_SAMPLE_NEAREST && p->sampling <= KH_SAMPLE_LANCZOS,
               KH_ERR_INVALID_ARG, "preprocess: unknown sampling mode %d", p->sampling);
    KH_REQUIRE(p->out_dtype == KH_OUT_F32 || p->out_dtype == KH_OUT_F16, KH_ERR_INVALID_ARG,
               "preprocess: unknown output dtype %d", p->out_dtype);
    // Subsampling constraints (SourceFormat::dims_ok, P/preprocess.rs:191-197).
    if (p->fmt == KH_FMT_NV12)
        KH_REQUIRE(p->src_w % 2 == 0 && p->src_h % 2 == 0, KH_ERR_INVALID_ARG,
                   "preprocess: NV12 needs even dimensions, got %dx%d", p->src_w, p->src_h);
    if (p->fmt == KH_FMT_YUYV)
        KH_REQUIRE(p->src_w % 2 == 0, KH_ERR_INVALID_ARG,
                   "preprocess: YUYV needs an even width, got %d", p->src_w);
    if (p->fmt == KH_FMT_RGB || p->fmt == KH_FMT_BGR)
        KH_REQUIRE(p->src_bpp == 3 || p->src_bpp == 4, KH_ERR_INVALID_ARG,
                   "preprocess: interleaved formats need bpp 3 or 4, got %d", p->src_bpp);
    const int min_bpp = p->fmt == KH_FMT_YUYV ? 2 : (p->fmt <= KH_FMT_BGR ? p->src_bpp : 1);
    KH_REQUIRE((int64_t)p->src_pitch >= (int64_t)p->src_w * min_bpp, KH_ERR_SLICE_TOO_SMALL,
               "preprocess: pitch %d shorter than a %d-px row", p->src_pitch, p->src_w);
    // 32-bit kernel indexing guard (P/preprocess.rs:1336-1339).
    KH_REQUIRE((int64_t)p->dst_w * p->dst_h <= kI32Max / 4 &&
                   (int64_t)p->src_pitch * p->src_h <= kI32Max &&
                   (p->fmt != KH_FMT_NV12 || (int64_t)p->src_w * p->src_h * 3 / 2 <= kI32Max),
               KH_ERR_TOO_LARGE, "preprocess: dimensions exceed the 32-bit kernel index limit");
    KH_REQUIRE(p->nframes <= 65535, KH_ERR_TOO_LARGE,
               "preprocess: at most 65535 frames per launch, got %d", p->nframes);
    if (p->nframes > 0) {
        KH_REQUIRE(dst, KH_ERR_INVALID_ARG, "preprocess: null device pointer");
        if (f.listed()) {
            for (int k = 0; k < f.n; ++k) KH_REQUIRE(f.list[k], KH_ERR_INVALID_ARG, "preprocess: null frame pointer at list index %d", k);
        } else {
            KH_REQUIRE(f.src, KH_ERR_INVALID_ARG, "preprocess: null device pointer");
        }
    }
    return KH_OK;
}

// `wide_ok`: every frame base of the CALL (not only of this launch's slice) is aligned for the format's one-load taps
template <int FMT, int SAMPLER>
void launch_generic_out(hipStream_t s, dim3 grid, const uint8_t* src, void* dst, const PreArgs& a,
                        int out_dtype, const Frames& f, const FrameList& fl) {
    // one-load taps need the frame base and the row pitch aligned to the tap's width
    bool wide = false;
    if (FMT == KH_FMT_NV12) wide = f.aligned(2) && a.src_w % 2 == 0;
    else if (FMT == KH_FMT_YUYV) wide = f.aligned(4) && a.src_pitch % 4 == 0;
    else if (FMT == KH_FMT_RGB || FMT == KH_FMT_BGR)
        wide = a.src_bpp == 4 && f.aligned(4) && a.src_pitch % 4 == 0;
    // flattened quads with 16-byte streaming stores (preprocess_generic_quads) for the ONE-tap samplers (nearest, on-grid bilinear), f32
    // outputs whose rows are whole quads.  Measured on one box, three interleaved rounds (profiles/r04d_quads_ab.txt): 1080p NV12 -> 640
    // on-grid 1.227 vs 1.329 ms, YUYV 1.206 vs 1.279 ms; the four-tap bilinear kernel does NOT gain from it (608: 1.464 vs 1.443 ms) and
    // two quads per lane lose everywhere, so those keep the per-pixel kernel — unless the source is NV12 and a quad's taps fit one 16-byte
    // run per plane row (quad_taps_nv12*: wide loads instead of gathers, r04q / r04r).  Test option pre_quads: 0 = never, 1 = also for
    // four taps with per-tap loads, 3 = no wide loads.
    if constexpr (SAMPLER != KH_SAMPLE_LANCZOS) {
        const int opt = dev_opt(kOptPreQuads);
        const bool quads_ok = a.dst_w % 4 == 0 && (int64_t)a.dst_w * a.dst_h * 12 <= kI32Max &&
                              reinterpret_cast<uintptr_t>(dst) % 16 == 0 && a.dst_frame_stride % 4 == 0;   // (f16 since round 6: 8-byte stores)
        const bool wide_nv12 = (FMT == KH_FMT_NV12 || (FMT == KH_FMT_YUYV && a.src_pitch >= 2 * a.src_w)) &&
                               opt != 3 && quads_ok && quad_taps_fit_16(a, SAMPLER == KH_SAMPLE_BILINEAR ? 1 : (SAMPLER == KH_SAMPLE_NEAREST ? 2 : 0));
        if (quads_ok && opt != 0 && (SAMPLER != KH_SAMPLE_BILINEAR || opt == 1 || wide_nv12)) {
            const int wq = a.dst_w / 4, groups = wq * a.dst_h;
            const dim3 qgrid(cdiv(groups, kQuadBlock), grid.z);
            const FastDiv by_wq = fast_div((uint32_t)wq);
            PreArgs aq = a;
            aq.quad_wide = wide_nv12 ? 1 : 0;   // test option pre_quads = 3: per-tap loads everywhere
            const NoFrames none{0};
#define KH_QUADS(T) do { \
            if (f.listed()) { \
                if (wide) hipLaunchKernelGGL((preprocess_generic_quads<FMT, SAMPLER, true, true, T>), qgrid, dim3(kQuadBlock), 0, s, src, (T*)dst, aq, by_wq, fl); \
                else hipLaunchKernelGGL((preprocess_generic_quads<FMT, SAMPLER, false, true, T>), qgrid, dim3(kQuadBlock), 0, s, src, (T*)dst, aq, by_wq, fl); \
            } else { \
                if (wide) hipLaunchKernelGGL((preprocess_generic_quads<FMT, SAMPLER, true, false, T>), qgrid, dim3(kQuadBlock), 0, s, src, (T*)dst, aq, by_wq, none); \
                else hipLaunchKernelGGL((preprocess_generic_quads<FMT, SAMPLER, false, false, T>), qgrid, dim3(kQuadBlock), 0, s, src, (T*)dst, aq, by_wq, none); \
            } } while (0)
            if (out_dtype == KH_OUT_F32) KH_QUADS(float); else KH_QUADS(unsigned short);
#undef KH_QUADS
            return;
        }
    }
    const dim3 blk(64, 4);
    const NoFrames none{0};
#define KH_GEN(T, W) do { if (f.listed()) hipLaunchKernelGGL((preprocess_generic<FMT, SAMPLER, T, W, true>), grid, blk, 0, s, src, (T*)dst, a, fl); \
                          else hipLaunchKernelGGL((preprocess_generic<FMT, SAMPLER, T, W, false>), grid, blk, 0, s, src, (T*)dst, a, none); } while (0)
    if (out_dtype == KH_OUT_F32) { if (wide) KH_GEN(float, true); else KH_GEN(float, false); }
    else { if (wide) KH_GEN(unsigned short, true); else KH_GEN(unsigned short, false); }
#undef KH_GEN
}

template <int FMT>
void launch_generic_fmt(hipStream_t s, dim3 grid, const uint8_t* src, void* dst, const PreArgs& a,
                        int sampling, int out_dtype, const Frames& f, const FrameList& fl) {
    switch (sampling) {
        case KH_SAMPLE_NEAREST:
            launch_generic_out<FMT, KH_SAMPLE_NEAREST>(s, grid, src, dst, a, out_dtype, f, fl);
            break;
        case KH_SAMPLE_BILINEAR:
            if (bilinear_taps_on_grid(a)) launch_generic_out<FMT, kSampleBilinearOnGrid>(s, grid, src, dst, a, out_dtype, f, fl);
            else launch_generic_out<FMT, KH_SAMPLE_BILINEAR>(s, grid, src, dst, a, out_dtype, f, fl);
            break;
        default:
            launch_generic_out<FMT, KH_SAMPLE_LANCZOS>(s, grid, src, dst, a, out_dtype, f, fl);
            break;
    }
}

int32_t preprocess_impl(kh_stream_t stream, const Frames& f, void* dst, const kh_preprocess_params* p) {
    int32_t rc = validate(p, f, dst);
    if (rc != KH_OK) return rc;
    if (p->nframes == 0) return KH_OK;

    PreArgs a;
    a.scale_x = p->scale_x; a.scale_y = p->scale_y; a.pad_x = p->pad_x; a.pad_y = p->pad_y;
    a.src_w = p->src_w; a.src_h = p->src_h; a.src_pitch = p->src_pitch; a.src_bpp = p->src_bpp;
    a.dst_w = p->dst_w; a.dst_h = p->dst_h;
    a.m0 = p->mean[0]; a.m1 = p->mean[1]; a.m2 = p->mean[2];
    a.is0 = p->inv_std[0]; a.is1 = p->inv_std[1]; a.is2 = p->inv_std[2];
    a.pad_value = p->pad_value;
    a.src_frame_stride = f.listed() ? 0 : p->src_frame_stride;
    a.dst_frame_stride = p->dst_frame_stride;
    a.rc_x = 1.0f / a.scale_x;
    a.rc_y = 1.0f / a.scale_y;
    a.fast_div = plan_division_is_exact(a) ? 1 : 0;
    a.lz_wx = a.lz_wy = nullptr;
    a.quad_wide = 0;
    a.f16_plain = p->out_dtype == KH_OUT_F16 && p->sampling != KH_SAMPLE_LANCZOS && f16_in_range(p) ? 1 : 0;
    hipStream_t s = as_hip(stream);

    const bool identity = identity_fast_path(p, f, dst);
    TableLease lz;  // keeps the weight table alive until the launch that reads it is enqueued and recorded
    if (!identity && p->sampling == KH_SAMPLE_LANCZOS) {
        rc = lanczos_tables(a, s, lz);
        if (rc != KH_OK) return rc;
    }
    const size_t out_elem = p->out_dtype == KH_OUT_F16 ? 2 : 4;
    static const FrameList kNoFrames{};
    // one launch for an equally spaced batch; one per kFrameListMax frames of a list
    for (int first = 0; first < p->nframes; first += f.listed() ? kFrameListMax : p->nframes) {
        const int n = f.listed() ? std::min(kFrameListMax, p->nframes - first) : p->nframes;
        FrameList fl_chunk;
        if (f.listed()) for (int k = 0; k < n; ++k) fl_chunk.p[k] = f.list[first + k];
        const FrameList& fl = f.listed() ? fl_chunk : kNoFrames;
        void* dst_chunk = static_cast<char*>(dst) + (size_t)first * (size_t)p->dst_frame_stride * out_elem;
        if (identity && p->out_dtype == KH_OUT_F16) {
            const unsigned bpf = cdiv((p->src_w / 8) * p->src_h, kIdBlock);
            const bool lut = dev_opt(kOptPreF16Lut) != 0;   // the table kernel (test option pre_f16_lut = 0: the arithmetic one); 2.73 -> 2.57 ms per 1024 frames, profiles/r06zv_f16_lut.txt
            if (f.listed() && lut) hipLaunchKernelGGL(preprocess_nv12_identity_f16_list<true>, dim3(bpf, (unsigned)n), dim3(kIdBlock), 0, s, fl, (unsigned short*)dst_chunk, a);
            else if (f.listed()) hipLaunchKernelGGL(preprocess_nv12_identity_f16_list<false>, dim3(bpf, (unsigned)n), dim3(kIdBlock), 0, s, fl, (unsigned short*)dst_chunk, a);
            else if (lut) hipLaunchKernelGGL(preprocess_nv12_identity_f16<true>, dim3(bpf, (unsigned)n), dim3(kIdBlock), 0, s, f.src, (unsigned short*)dst_chunk, a);
            else hipLaunchKernelGGL(preprocess_nv12_identity_f16<false>, dim3(bpf, (unsigned)n), dim3(kIdBlock), 0, s, f.src, (unsigned short*)dst_chunk, a);
            rc = check_launch("preprocess_nv12_identity_f16");
        } else if (identity) {
            const int groups = (p->src_w / 4) * p->src_h;
            const unsigned bpf = cdiv(groups, kIdBlock);
            if (f.listed()) hipLaunchKernelGGL(preprocess_nv12_identity_list, dim3(bpf, (unsigned)n), dim3(kIdBlock), 0, s, fl, (float*)dst_chunk, a);
            else hipLaunchKernelGGL(preprocess_nv12_identity, dim3(bpf, (unsigned)n), dim3(kIdBlock), 0, s, f.src, (float*)dst_chunk, a);
            rc = check_launch("preprocess_nv12_identity");
        } else {
            dim3 grid(cdiv(p->dst_w, 64 * kGenPx), cdiv(p->dst_h, 4), (unsigned)n);
            switch (p->fmt) {
                case KH_FMT_RGB: launch_generic_fmt<KH_FMT_RGB>(s, grid, f.src, dst_chunk, a, p->sampling, p->out_dtype, f, fl); break;
                case KH_FMT_BGR: launch_generic_fmt<KH_FMT_BGR>(s, grid, f.src, dst_chunk, a, p->sampling, p->out_dtype, f, fl); break;
                case KH_FMT_GRAY: launch_generic_fmt<KH_FMT_GRAY>(s, grid, f.src, dst_chunk, a, p->sampling, p->out_dtype, f, fl); break;
                case KH_FMT_NV12: launch_generic_fmt<KH_FMT_NV12>(s, grid, f.src, dst_chunk, a, p->sampling, p->out_dtype, f, fl); break;
                default: launch_generic_fmt<KH_FMT_YUYV>(s, grid, f.src, dst_chunk, a, p->sampling, p->out_dtype, f, fl); break;
            }
            rc = check_launch("preprocess_generic");
        }
        if (rc != KH_OK) break;
    }
    if (lz) lz->used_on(s);
    return rc;
}

}  // namespace

extern "C" {

const char* kh_preprocess_variant(const kh_preprocess_params* p) {
    // Alignment of the actual buffers is only known at launch; report for aligned buffers.
    const Frames aligned{reinterpret_cast<const uint8_t*>(16), p ? p->src_frame_stride : 0, nullptr, p ? p->nframes : 0};
    if (validate(p, aligned, reinterpret_cast<void*>(16)) != KH_OK)
        return nullptr;
    if (identity_fast_path(p, aligned, reinterpret_cast<void*>(16))) return p->out_dtype == KH_OUT_F16 ? "nv12_identity_f16" : "nv12_identity";
    if (p->sampling == KH_SAMPLE_BILINEAR) {
        PreArgs a{};
        a.scale_x = p->scale_x; a.scale_y = p->scale_y; a.pad_x = p->pad_x; a.pad_y = p->pad_y;
        a.src_w = p->src_w; a.src_h = p->src_h; a.dst_w = p->dst_w; a.dst_h = p->dst_h;
        if (bilinear_taps_on_grid(a)) return "generic_bilinear_on_grid";
    }
    return "generic";
}

int32_t kh_preprocess_to_chw(kh_stream_t stream, const uint8_t* src, void* dst,
                             const kh_preprocess_params* p) {
    return preprocess_impl(stream, Frames{src, p ? p->src_frame_stride : 0, nullptr, p ? p->nframes : 0}, dst, p);
}

int32_t kh_preprocess_to_chw_list(kh_stream_t stream, const uint8_t* const* frames, void* dst,
                                  const kh_preprocess_params* p) {
    KH_REQUIRE(!p || p->nframes <= 0 || frames, KH_ERR_INVALID_ARG, "preprocess: null frame list");
    static const uint8_t* const kEmpty[1] = {nullptr};
    return preprocess_impl(stream, Frames{nullptr, 0, frames ? frames : kEmpty, p ? p->nframes : 0}, dst, p);
}

}  // extern "C"
