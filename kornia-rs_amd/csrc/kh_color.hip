// Per-pixel colour conversions for gfx950.
//
// Device twins of the reference's colour kernels (crates/kornia-imgproc/src/cuda/color/*.rs),
// arithmetic following the CPU scalar paths in crates/kornia-imgproc/src/color/** so that u8
// results are bit-exact and f32 results are bit-identical expression trees (built with
// -ffp-contract=off).  All of these are HBM-bound maps; the only design question is coalescing
// 3-byte / 12-byte interleaved pixels:
//   * u8:  a thread owns 4 pixels = CIN dwords in, COUT dwords out (global_load_dwordx3 /
//          dwordx4), so a wave reads 768 B and writes 256..1024 B contiguous per instruction;
//          bytes are unpacked with shifts in registers.  The <4-pixel tail is handled by one
//          extra thread byte-wise; buffers that are not 4-byte aligned take a 1-px/thread path.
//   * f32: a thread owns one pixel = CIN floats (dwordx3), 768 B contiguous per wave.
#include "kh_common.h"
#include "kh_libm_glibc.h"

using namespace kh;

namespace {

template <int N> struct Words { uint32_t w[N]; };
template <int N> struct Floats { float v[N]; };

template <int CIN, int COUT, typename Op>
__global__ __launch_bounds__(kBlock) void map_u8_quads(const uint8_t* __restrict__ src,
                                                       uint8_t* __restrict__ dst, long long npx, Op op) {
    const long long q = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long nq = npx >> 2;
    if (q < nq) {
        const Words<CIN> in = *reinterpret_cast<const Words<CIN>*>(src + q * 4 * CIN);
        Words<COUT> out;
#pragma unroll
        for (int k = 0; k < COUT; ++k) out.w[k] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int pi[CIN], po[COUT];
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const int b = j * CIN + c;
                pi[c] = (int)((in.w[b >> 2] >> (8 * (b & 3))) & 0xFFu);
            }
            op(pi, po);
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                const int b = j * COUT + c;
                out.w[b >> 2] |= ((uint32_t)po[c] & 0xFFu) << (8 * (b & 3));
            }
        }
        // block-local streaming window: write-through non-temporal store of COUT dwords (kh_common.h::stream_store)
        const long long q0 = (long long)blockIdx.x * kBlock;
        stream_store<COUT>(stream_window(dst + q0 * 4 * COUT, (nq - q0) * 4 * COUT), (int)threadIdx.x * 4 * COUT, out.w);
    } else if (q == nq) {
        for (long long p = nq * 4; p < npx; ++p) {
            int pi[CIN], po[COUT];
            for (int c = 0; c < CIN; ++c) pi[c] = src[p * CIN + c];
            op(pi, po);
            for (int c = 0; c < COUT; ++c) dst[p * COUT + c] = (uint8_t)po[c];
        }
    }
}

template <int CIN, int COUT, typename Op>
__global__ __launch_bounds__(kBlock) void map_u8_bytes(const uint8_t* __restrict__ src,
                                                       uint8_t* __restrict__ dst, long long npx, Op op) {
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= npx) return;
    int pi[CIN], po[COUT];
#pragma unroll
    for (int c = 0; c < CIN; ++c) pi[c] = src[p * CIN + c];
    op(pi, po);
#pragma unroll
    for (int c = 0; c < COUT; ++c) dst[p * COUT + c] = (uint8_t)po[c];
}

template <int CIN, int COUT, typename Op>
__global__ __launch_bounds__(kBlock) void map_f32(const float* __restrict__ src, float* __restrict__ dst,
                                                  long long npx, Op op) {
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= npx) return;
    const Floats<CIN> in = *reinterpret_cast<const Floats<CIN>*>(src + p * CIN);
    Floats<COUT> out;
    op(in.v, out.v);
    uint32_t w[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) w[c] = __float_as_uint(out.v[c]);
    const long long p0 = (long long)blockIdx.x * kBlock;
    stream_store<COUT>(stream_window(dst + p0 * COUT, (npx - p0) * COUT * 4), (int)threadIdx.x * COUT * 4, w);
}

// (Round 5: a gray kernel whose lanes own FOUR pixels — three 16-byte loads 48 bytes apart per lane, one 16-byte store — measured 5.69 ms against
// 5.57 for the pixel-per-lane form above on 1024 1080p frames, three interleaved rounds: the strided loads cost more than the 1 KiB store
// segments return (profiles/r05t_gray_f32_quads_ab.txt).  normalize_mean_std, whose lanes can own four FLOATS, gained 3 % from that width.
// A second form — every global access 16 bytes and lane-contiguous, the pixels re-assembled through a wave-private LDS slot, for 3 -> 1
// and 3 -> 3 channel maps — also lost: gray 5.73 vs 5.57 ms, YCbCr 4.48 vs 4.15, HSV 4.32 vs 4.31 (profiles/r05u_f32_maps_lds_transpose_ab.txt).)
int32_t check_map(const void* src, const void* dst, int64_t npx, const char* what) {
    KH_REQUIRE(npx >= 0, KH_ERR_INVALID_ARG, "%s: negative pixel count", what);
    if (npx == 0) return KH_OK;
    KH_REQUIRE(src && dst, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    KH_REQUIRE(npx <= (int64_t)kI32Max * 64, KH_ERR_TOO_LARGE, "%s: %lld pixels exceed the launch limit",
               what, (long long)npx);
    return KH_OK;
}

template <int CIN, int COUT, typename Op>
int32_t launch_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int64_t npx, Op op, const char* what) {
    int32_t rc = check_map(src, dst, npx, what);
    if (rc != KH_OK || npx == 0) return rc;
    hipStream_t s = as_hip(stream);
    const bool aligned = (reinterpret_cast<uintptr_t>(src) % 4 == 0) && (reinterpret_cast<uintptr_t>(dst) % 4 == 0);
    if (aligned)
        hipLaunchKernelGGL((map_u8_quads<CIN, COUT, Op>), dim3(cdiv((npx >> 2) + 1, kBlock)), dim3(kBlock), 0, s,
                           src, dst, (long long)npx, op);
    else
        hipLaunchKernelGGL((map_u8_bytes<CIN, COUT, Op>), dim3(cdiv(npx, kBlock)), dim3(kBlock), 0, s, src, dst,
                           (long long)npx, op);
    return check_launch(what);
}

template <int CIN, int COUT, typename Op>
int32_t launch_f32(kh_stream_t stream, const float* src, float* dst, int64_t npx, Op op, const char* what) {
    int32_t rc = check_map(src, dst, npx, what);
    if (rc != KH_OK || npx == 0) return rc;
    hipLaunchKernelGGL((map_f32<CIN, COUT, Op>), dim3(cdiv(npx, kBlock)), dim3(kBlock), 0, as_hip(stream), src,
                       dst, (long long)npx, op);
    return check_launch(what);
}

__device__ __forceinline__ int clamp255(int v) { return min(max(v, 0), 255); }

// ---- gray (P/color/gray/kernels.rs:2-13,229-244,405-412,463-469) -----------------------------
struct GrayFromRgbU8 {
    __device__ void operator()(const int in[3], int out[1]) const {
        out[0] = (int)((4899u * (unsigned)in[0] + 9617u * (unsigned)in[1] + 1868u * (unsigned)in[2] + 8192u) >> 14);
    }
};
struct GrayFromRgbF32 {
    __device__ void operator()(const float in[3], float out[1]) const {
        out[0] = 0.299f * in[0] + 0.587f * in[1] + 0.114f * in[2];
    }
};
struct RgbFromGrayU8 {
    __device__ void operator()(const int in[1], int out[3]) const { out[0] = out[1] = out[2] = in[0]; }
};
struct RgbFromGrayF32 {
    __device__ void operator()(const float in[1], float out[3]) const { out[0] = out[1] = out[2] = in[0]; }
};

// ---- swizzles (P/color/rgb/kernels.rs, rgb/mod.rs:128-227,311-316) -----------------------------
struct BgrFromRgbU8 {
    __device__ void operator()(const int in[3], int out[3]) const { out[0] = in[2]; out[1] = in[1]; out[2] = in[0]; }
};
struct BgrFromRgbF32 {
    __device__ void operator()(const float in[3], float out[3]) const { out[0] = in[2]; out[1] = in[1]; out[2] = in[0]; }
};
struct RgbaFromRgbU8 {
    int swap;
    __device__ void operator()(const int in[3], int out[4]) const {
        out[0] = swap ? in[2] : in[0]; out[1] = in[1]; out[2] = swap ? in[0] : in[2]; out[3] = 255;
    }
};
struct RgbaFromRgbF32 {
    int swap;
    __device__ void operator()(const float in[3], float out[4]) const {
        out[0] = swap ? in[2] : in[0]; out[1] = in[1]; out[2] = swap ? in[0] : in[2]; out[3] = 1.0f;
    }
};
struct RgbFromRgbaU8 {
    int swap, has_bg;
    float bg0, bg1, bg2;
    __device__ void operator()(const int in[4], int out[3]) const {
        const int r = swap ? in[2] : in[0], g = in[1], b = swap ? in[0] : in[2];
        if (has_bg) {
            const float alpha = (float)in[3] / 255.0f;
            out[0] = (int)roundf((float)r * alpha + bg0 * (1.0f - alpha));
            out[1] = (int)roundf((float)g * alpha + bg1 * (1.0f - alpha));
            out[2] = (int)roundf((float)b * alpha + bg2 * (1.0f - alpha));
        } else {
            out[0] = r; out[1] = g; out[2] = b;
        }
    }
};

// ---- Family A: full-range YCbCr / YUV (P/color/yuv/kernels.rs:23-62,82-126,541-551,673-690) ----
struct YccFromRgbU8 {
    int order, c_rv, c_bu;
    __device__ void operator()(const int in[3], int out[3]) const {
        const int r = in[0], g = in[1], b = in[2];
        const int y = (4899 * r + 9617 * g + 1868 * b + 8192) >> 14;
        const int cr = clamp255(((r - y) * c_rv + (128 << 14) + 8192) >> 14);
        const int cb = clamp255(((b - y) * c_bu + (128 << 14) + 8192) >> 14);
        out[0] = clamp255(y);
        out[1] = order == 0 ? cr : cb;
        out[2] = order == 0 ? cb : cr;
    }
};
struct RgbFromYccU8 {
    int order;
    __device__ void operator()(const int in[3], int out[3]) const {
        const int y = in[0];
        const int cr = (order == 0 ? in[1] : in[2]) - 128, cb = (order == 0 ? in[2] : in[1]) - 128;
        int r, g, b;
        if (order == 0) {
            r = y + ((22987 * cr + 8192) >> 14);
            g = y + ((-11698 * cr + -5636 * cb + 8192) >> 14);
            b = y + ((29049 * cb + 8192) >> 14);
        } else {
            r = y + ((18678 * cr + 8192) >> 14);
            g = y + ((-9519 * cr + -6472 * cb + 8192) >> 14);
            b = y + (((16646 * cb) * 2 + 8192) >> 14);
        }
        out[0] = clamp255(r); out[1] = clamp255(g); out[2] = clamp255(b);
    }
};
struct YccFromRgbF32 {
    int order;
    float k_rv, k_bu;
    __device__ void operator()(const float in[3], float out[3]) const {
        const float r = in[0], g = in[1], b = in[2];
        const float y = 0.299f * r + 0.587f * g + 0.114f * b;
        const float cr = (r - y) * k_rv + 0.5f;
        const float cb = (b - y) * k_bu + 0.5f;
        out[0] = y;
        out[1] = order == 0 ? cr : cb;
        out[2] = order == 0 ? cb : cr;
    }
};
struct RgbFromYccF32 {
    int order;
    __device__ void operator()(const float in[3], float out[3]) const {
        const float y = in[0];
        const float cr = order == 0 ? in[1] : in[2], cb = order == 0 ? in[2] : in[1];
        if (order == 0) {
            const float r = y + (cr - 0.5f) / 0.713f;
            const float b = y + (cb - 0.5f) / 0.564f;
            out[0] = r;
            out[1] = (y - 0.299f * r - 0.114f * b) / 0.587f;
            out[2] = b;
        } else {
            out[0] = y + 1.140f * (cr - 0.5f);
            out[1] = y + -0.395f * (cb - 0.5f) + -0.581f * (cr - 0.5f);
            out[2] = y + 2.032f * (cb - 0.5f);
        }
    }
};

// ---- HSV / HLS in the 0..255 domain (P/color/hsv/kernels.rs:150-177,310-334;
//      P/color/hls/kernels.rs:159-187,357-391) -------------------------------------------------
#define KH_INV_255 (1.0f / 255.0f)
#define KH_DEG_TO_BYTE (255.0f / 360.0f)
#define KH_BYTE_TO_DEG (360.0f / 255.0f)

__device__ __forceinline__ float hue_deg(float r, float g, float b, float mx, float delta) {
    float h;
    if (mx == r) h = 60.0f * fmodf((g - b) / delta, 6.0f);
    else if (mx == g) h = 60.0f * (((b - r) / delta) + 2.0f);
    else h = 60.0f * (((r - g) / delta) + 4.0f);
    return h < 0.0f ? h + 360.0f : h;
}
struct HsvFromRgbF32 {
    __device__ void operator()(const float in[3], float out[3]) const {
        const float r = in[0] * KH_INV_255, g = in[1] * KH_INV_255, b = in[2] * KH_INV_255;
        const float mx = fmaxf(fmaxf(r, g), b), mn = fminf(fminf(r, g), b), delta = mx - mn;
        const float h = delta == 0.0f ? 0.0f : hue_deg(r, g, b, mx, delta);
        out[0] = h * KH_DEG_TO_BYTE;
        out[1] = mx == 0.0f ? 0.0f : (delta / mx) * 255.0f;
        out[2] = mx * 255.0f;
    }
};
struct RgbFromHsvF32 {
    __device__ void operator()(const float in[3], float out[3]) const {
        const float s = in[1] * KH_INV_255, v = in[2] * KH_INV_255;
        const float hh = in[0] * (KH_BYTE_TO_DEG / 60.0f);
        const float c = v * s;
        const float hmod2 = hh - 2.0f * floorf(hh * 0.5f);
        const float x = c * (1.0f - fabsf(hmod2 - 1.0f));
        const float m = v - c;
        const int sext = (int)floorf(hh);
        float r1, g1, b1;
        switch (sext) {
            case 0: r1 = c; g1 = x; b1 = 0.0f; break;
            case 1: r1 = x; g1 = c; b1 = 0.0f; break;
            case 2: r1 = 0.0f; g1 = c; b1 = x; break;
            case 3: r1 = 0.0f; g1 = x; b1 = c; break;
            case 4: r1 = x; g1 = 0.0f; b1 = c; break;
            default: r1 = c; g1 = 0.0f; b1 = x; break;
        }
        out[0] = (r1 + m) * 255.0f; out[1] = (g1 + m) * 255.0f; out[2] = (b1 + m) * 255.0f;
    }
};
struct HlsFromRgbF32 {
    __device__ void operator()(const float in[3], float out[3]) const {
        const float r = in[0] * KH_INV_255, g = in[1] * KH_INV_255, b = in[2] * KH_INV_255;
        const float mx = fmaxf(fmaxf(r, g), b), mn = fminf(fminf(r, g), b);
        const float diff = mx - mn, sum = mx + mn, l = sum * 0.5f;
        float h = 0.0f, s = 0.0f;
        if (diff != 0.0f) {
            s = l <= 0.5f ? diff / sum : diff / (2.0f - sum);
            h = hue_deg(r, g, b, mx, diff);
        }
        out[0] = h * KH_DEG_TO_BYTE; out[1] = l * 255.0f; out[2] = s * 255.0f;
    }
};
__device__ __forceinline__ float hue2rgb(float p, float q, float t) {
    if (t < 0.0f) t = t + 1.0f;
    if (t > 1.0f) t = t - 1.0f;
    if (t < 1.0f / 6.0f) return p + (q - p) * 6.0f * t;
    if (t < 0.5f) return q;
    if (t < 2.0f / 3.0f) return p + (q - p) * (2.0f / 3.0f - t) * 6.0f;
    return p;
}
struct RgbFromHlsF32 {
    __device__ void operator()(const float in[3], float out[3]) const {
        const float l = in[1] * KH_INV_255, s = in[2] * KH_INV_255;
        if (s == 0.0f) {
            const float v = l * 255.0f;
            out[0] = v; out[1] = v; out[2] = v;
            return;
        }
        const float h_deg = in[0] * KH_BYTE_TO_DEG;
        const float q = l < 0.5f ? l * (1.0f + s) : l + s - l * s;
        const float p = 2.0f * l - q;
        const float hk = h_deg / 360.0f;
        out[0] = hue2rgb(p, q, hk + 1.0f / 3.0f) * 255.0f;
        out[1] = hue2rgb(p, q, hk) * 255.0f;
        out[2] = hue2rgb(p, q, hk - 1.0f / 3.0f) * 255.0f;
    }
};

// ---- sepia (P/color/sepia.rs:17-22,86-104; matrix.rs:122-136), colormap LUT (colormap.rs:115-122)
struct SepiaU8 {
    __device__ void operator()(const int in[3], int out[3]) const {
        const unsigned r = in[0], g = in[1], b = in[2];
        out[0] = (int)min((101u * r + 197u * g + 48u * b + 128u) >> 8, 255u);
        out[1] = (int)min((89u * r + 176u * g + 43u * b + 128u) >> 8, 255u);
        out[2] = (int)min((70u * r + 137u * g + 34u * b + 128u) >> 8, 255u);
    }
};
struct SepiaF32 {
    __device__ void operator()(const float in[3], float out[3]) const {
        const float c0 = in[0], c1 = in[1], c2 = in[2];
        out[0] = 0.0f + 0.393f * c0 + 0.769f * c1 + 0.189f * c2;
        out[1] = 0.0f + 0.349f * c0 + 0.686f * c1 + 0.168f * c2;
        out[2] = 0.0f + 0.272f * c0 + 0.534f * c1 + 0.131f * c2;
    }
};
struct ColormapU8 {
    const uint8_t* lut;  // device: r[256] g[256] b[256]
    __device__ void operator()(const int in[1], int out[3]) const {
        out[0] = lut[in[0]]; out[1] = lut[256 + in[0]]; out[2] = lut[512 + in[0]];
    }
};

// ---- Family B / C: video formats (P/color/yuv/kernels.rs:696-1216 decode, 1223-1573 encode) ---
constexpr int kCY = 1220542, kCUB = 2116026, kCUG = -409993, kCVG = -852492, kCVR = 1673527, kHalf20 = 1 << 19;

__device__ __forceinline__ void decode3(int y, int tr, int tg, int tb, int rgb[3]) {
    const int yy = max(y - 16, 0) * kCY;
    rgb[0] = clamp255((yy + tr) >> 20);
    rgb[1] = clamp255((yy + tg) >> 20);
    rgb[2] = clamp255((yy + tb) >> 20);
}
__device__ __forceinline__ void chroma_terms(int u, int v, int& tr, int& tg, int& tb) {
    u -= 128; v -= 128;
    tb = kCUB * u + kHalf20;
    tg = kCUG * u + kCVG * v + kHalf20;
    tr = kCVR * v + kHalf20;
}
__device__ __forceinline__ void put12(uint8_t* p, const int px[4][3], bool aligned) {
    if (aligned) {
        Words<3> o = {{0, 0, 0}};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int b = j * 3 + c;
                o.w[b >> 2] |= (uint32_t)px[j][c] << (8 * (b & 3));
            }
        *reinterpret_cast<Words<3>*>(p) = o;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) p[j * 3 + c] = (uint8_t)px[j][c];
    }
}

// planar 4:2:0 -> RGB8.  A thread owns a 2x2 block column pair: up to 4 px x 2 rows when the
// width allows dword access (w % 4 == 0), else a 2x2 block with byte access.
// layout: 0 NV12, 1 NV21, 2 I420, 3 YV12
template <int PXW>
__global__ __launch_bounds__(kBlock) void rgb_from_planar420(const uint8_t* __restrict__ src,
                                                             uint8_t* __restrict__ dst, int w, int h,
                                                             int layout, int dst_aligned) {
    const int gw = w / PXW;
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= gw * (h >> 1)) return;
    const int cy = g / gw, x = (g - cy * gw) * PXW;
    const uint8_t* yp = src;
    const uint8_t* c0 = src + (long long)w * h;
    const int cw = w >> 1;
    const uint8_t* c1 = c0 + (long long)cw * (h >> 1);
    int yt[PXW], yb[PXW], uu[PXW / 2], vv[PXW / 2];
    if constexpr (PXW == 4) {
        const uint32_t t = *reinterpret_cast<const uint32_t*>(yp + (long long)(2 * cy) * w + x);
        const uint32_t b = *reinterpret_cast<const uint32_t*>(yp + (long long)(2 * cy + 1) * w + x);
#pragma unroll
        for (int j = 0; j < 4; ++j) { yt[j] = (t >> (8 * j)) & 0xFF; yb[j] = (b >> (8 * j)) & 0xFF; }
    } else {
#pragma unroll
        for (int j = 0; j < PXW; ++j) {
            yt[j] = yp[(long long)(2 * cy) * w + x + j];
            yb[j] = yp[(long long)(2 * cy + 1) * w + x + j];
        }
    }
#pragma unroll
    for (int k = 0; k < PXW / 2; ++k) {
        const int cx = (x >> 1) + k;
        int a, b;
        if (layout <= 1) { a = c0[(long long)cy * cw * 2 + cx * 2]; b = c0[(long long)cy * cw * 2 + cx * 2 + 1]; }
        else { a = c0[(long long)cy * cw + cx]; b = c1[(long long)cy * cw + cx]; }
        // NV12: (U,V)=(a,b); NV21: (b,a); I420: c0=U,c1=V; YV12: c0=V,c1=U
        const bool swap = (layout == 1) || (layout == 3);
        uu[k] = swap ? b : a;
        vv[k] = swap ? a : b;
    }
    int top[4][3], bot[4][3];
#pragma unroll
    for (int j = 0; j < PXW; ++j) {
        int tr, tg, tb;
        chroma_terms(uu[j >> 1], vv[j >> 1], tr, tg, tb);
        decode3(yt[j], tr, tg, tb, top[j]);
        decode3(yb[j], tr, tg, tb, bot[j]);
    }
    uint8_t* dt = dst + ((long long)(2 * cy) * w + x) * 3;
    uint8_t* db = dst + ((long long)(2 * cy + 1) * w + x) * 3;
    if constexpr (PXW == 4) {
        put12(dt, top, dst_aligned);
        put12(db, bot, dst_aligned);
    } else {
#pragma unroll
        for (int j = 0; j < PXW; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) { dt[j * 3 + c] = (uint8_t)top[j][c]; db[j * 3 + c] = (uint8_t)bot[j][c]; }
    }
}

// packed 4:2:2 -> RGB8: a thread owns one 4-byte group (2 px).  layout: 0 YUYV, 1 UYVY, 2 YVYU
__global__ __launch_bounds__(kBlock) void rgb_from_packed422(const uint8_t* __restrict__ src,
                                                             uint8_t* __restrict__ dst, long long ngroups,
                                                             int layout, int src_aligned) {
    const long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (g >= ngroups) return;
    int q[4];
    if (src_aligned) {
        const uint32_t w4 = *reinterpret_cast<const uint32_t*>(src + g * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = (w4 >> (8 * j)) & 0xFF;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = src[g * 4 + j];
    }
    int y0, u, y1, v;
    if (layout == 0) { y0 = q[0]; u = q[1]; y1 = q[2]; v = q[3]; }
    else if (layout == 1) { u = q[0]; y0 = q[1]; v = q[2]; y1 = q[3]; }
    else { y0 = q[0]; v = q[1]; y1 = q[2]; u = q[3]; }
    int tr, tg, tb, a[3], b[3];
    chroma_terms(u, v, tr, tg, tb);
    decode3(y0, tr, tg, tb, a);
    decode3(y1, tr, tg, tb, b);
    uint8_t* d = dst + g * 6;
    d[0] = (uint8_t)a[0]; d[1] = (uint8_t)a[1]; d[2] = (uint8_t)a[2];
    d[3] = (uint8_t)b[0]; d[4] = (uint8_t)b[1]; d[5] = (uint8_t)b[2];
}

__device__ __forceinline__ int encode_y(int r, int g, int b) {
    return clamp255(((66 * r + 129 * g + 25 * b + 128) >> 8) + 16);
}
__device__ __forceinline__ void encode_uv(int r, int g, int b, int& u, int& v) {
    u = clamp255(((-38 * r + -74 * g + 112 * b + 128) >> 8) + 128);
    v = clamp255(((112 * r + -94 * g + -18 * b + 128) >> 8) + 128);
}

// RGB8 -> NV12: a thread owns one 2x2 block (2 luma pairs + 1 chroma pair)
__global__ __launch_bounds__(kBlock) void nv12_from_rgb(const uint8_t* __restrict__ src,
                                                        uint8_t* __restrict__ dst, int w, int h) {
    const int cw = w >> 1;
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= cw * (h >> 1)) return;
    const int cy = g / cw, cx = g - cy * cw;
    const uint8_t* top = src + ((long long)(2 * cy) * w + 2 * cx) * 3;
    const uint8_t* bot = top + (long long)w * 3;
    int t[6], b[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { t[k] = top[k]; b[k] = bot[k]; }
    uint8_t* yo = dst;
    uint8_t* uvo = dst + (long long)w * h;
    yo[(long long)(2 * cy) * w + 2 * cx] = (uint8_t)encode_y(t[0], t[1], t[2]);
    yo[(long long)(2 * cy) * w + 2 * cx + 1] = (uint8_t)encode_y(t[3], t[4], t[5]);
    yo[(long long)(2 * cy + 1) * w + 2 * cx] = (uint8_t)encode_y(b[0], b[1], b[2]);
    yo[(long long)(2 * cy + 1) * w + 2 * cx + 1] = (uint8_t)encode_y(b[3], b[4], b[5]);
    int u, v;
    encode_uv((t[0] + t[3] + b[0] + b[3] + 2) >> 2, (t[1] + t[4] + b[1] + b[4] + 2) >> 2,
              (t[2] + t[5] + b[2] + b[5] + 2) >> 2, u, v);
    uvo[(long long)cy * w + 2 * cx] = (uint8_t)u;
    uvo[(long long)cy * w + 2 * cx + 1] = (uint8_t)v;
}

// RGB8 -> YUYV: a thread owns one pixel pair
__global__ __launch_bounds__(kBlock) void yuyv_from_rgb(const uint8_t* __restrict__ src,
                                                        uint8_t* __restrict__ dst, long long ngroups) {
    const long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (g >= ngroups) return;
    int q[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) q[k] = src[g * 6 + k];
    int u, v;
    encode_uv((q[0] + q[3] + 1) >> 1, (q[1] + q[4] + 1) >> 1, (q[2] + q[5] + 1) >> 1, u, v);
    uint8_t* d = dst + g * 4;
    d[0] = (uint8_t)encode_y(q[0], q[1], q[2]);
    d[1] = (uint8_t)u;
    d[2] = (uint8_t)encode_y(q[3], q[4], q[5]);
    d[3] = (uint8_t)v;
}

int32_t check_wh(const void* src, const void* dst, int w, int h, bool even_w, bool even_h, const char* what) {
    KH_REQUIRE(w >= 0 && h >= 0, KH_ERR_INVALID_ARG, "%s: negative size %dx%d", what, w, h);
    KH_REQUIRE(!even_w || w % 2 == 0, KH_ERR_INVALID_ARG, "%s: width %d must be even", what, w);
    KH_REQUIRE(!even_h || h % 2 == 0, KH_ERR_INVALID_ARG, "%s: height %d must be even", what, h);
    KH_REQUIRE((int64_t)w * h * 3 <= kI32Max, KH_ERR_TOO_LARGE, "%s: %dx%d exceeds 32-bit indexing", what, w, h);
    if ((int64_t)w * h > 0) KH_REQUIRE(src && dst, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    return KH_OK;
}


// ---- CIE colour spaces (P/color/cie: kernels.rs:21-340, transfer.rs:14-45) -------------------------------
// f32 per-pixel formulas of the reference's scalar path.  The matrix-only conversions are plain mul/add and bit-identical.
// The sRGB transfer and the Lab / Luv stages call powf / cbrtf in the reference — libm functions, i.e. a third-party dependency
// whose last bit differs between implementations (round 2 called the device library's and needed 2e-4 + 2e-5 |v| against the
// restatement).  Since round 3 the device evaluates THE SAME functions as the platform libm the reference's CPU path binds to:
// glibc 2.35's powf and cbrtf, restated in f64 arithmetic with glibc's own tables (csrc/kh_libm_glibc.h, generated and checked
// by scripts/gen_libm_tables.py; tests/test_cie.py compares them with the box's libm on millions of arguments, bit for bit).
// With identical functions and identical plain mul / add / div around them, every CIE conversion is bit-identical to the
// restatement.  Arguments outside the restated powf's domain (zero, subnormal, > 2^30, infinite, NaN) take the device library.
__device__ __forceinline__ float pow_libm(float x, float y) {
    return kh_libm::powf_in_domain(x, y) ? kh_libm::powf_glibc(x, y) : powf(x, y);
}
__device__ __forceinline__ float cbrt_libm(float x) { return kh_libm::cbrtf_glibc(x); }
constexpr float kM_RGB2XYZ[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f, 0.072169f, 0.019334f, 0.119193f, 0.950227f};
constexpr float kM_XYZ2RGB[9] = {3.240479f, -1.537150f, -0.498535f, -0.969256f, 1.875991f, 0.041556f, 0.055648f, -0.204043f, 1.057311f};
constexpr float kXN = 0.950456f, kZN = 1.088754f, kInvXN = 1.0f / kXN, kInvZN = 1.0f / kZN;
constexpr float kLabDelta = 0.008856f, kLabFSlope = 1.0f / 0.12841855f, kLabFOffset = 0.13793103f;
constexpr float kLabFinvThresh = 0.20689655f, kLabFinvSlope = 0.12841855f;
constexpr float kLuvUn = 0.19793943f, kLuvVn = 0.46831096f, kLuvKappa = 903.3f;

__device__ __forceinline__ float srgb_to_linear(float x) {  // transfer.rs:27-35
    x = fmaxf(x, 0.0f);
    return x <= 0.04045f ? x * (1.0f / 12.92f) : pow_libm((x + 0.055f) * (1.0f / 1.055f), 2.4f);
}
__device__ __forceinline__ float linear_to_srgb(float l) {  // transfer.rs:37-45
    l = fmaxf(l, 0.0f);
    return l <= 0.0031308f ? l * 12.92f : 1.055f * pow_libm(l, 1.0f / 2.4f) - 0.055f;
}
__device__ __forceinline__ void matvec32(const float m[9], float a, float b, float c, float o[3]) {
    o[0] = m[0] * a + m[1] * b + m[2] * c;
    o[1] = m[3] * a + m[4] * b + m[5] * c;
    o[2] = m[6] * a + m[7] * b + m[8] * c;
}
__device__ __forceinline__ float lab_f(float t) { return t > kLabDelta ? cbrt_libm(t) : t * kLabFSlope + kLabFOffset; }
__device__ __forceinline__ float lab_finv(float f) { return f > kLabFinvThresh ? f * f * f : kLabFinvSlope * (f - kLabFOffset); }
__device__ __forceinline__ void lin_xyz_from_rgb(const float in[3], float o[3]) {
    matvec32(kM_RGB2XYZ, srgb_to_linear(in[0]), srgb_to_linear(in[1]), srgb_to_linear(in[2]), o);
}
__device__ __forceinline__ void rgb_from_lin_xyz(float x, float y, float z, float out[3]) {
    float l[3];
    matvec32(kM_XYZ2RGB, x, y, z, l);
    out[0] = linear_to_srgb(l[0]); out[1] = linear_to_srgb(l[1]); out[2] = linear_to_srgb(l[2]);
}
struct CieF32 {
    int conv;
    __device__ void operator()(const float in[3], float out[3]) const {
        switch (conv) {  // wave-uniform
            case KH_CIE_LINEAR_RGB_FROM_RGB:
                out[0] = srgb_to_linear(in[0]); out[1] = srgb_to_linear(in[1]); out[2] = srgb_to_linear(in[2]);
                break;
            case KH_CIE_RGB_FROM_LINEAR_RGB:
                out[0] = linear_to_srgb(in[0]); out[1] = linear_to_srgb(in[1]); out[2] = linear_to_srgb(in[2]);
                break;
            case KH_CIE_XYZ_FROM_RGB: matvec32(kM_RGB2XYZ, in[0], in[1], in[2], out); break;  // no gamma, kernels.rs:124-131
            case KH_CIE_RGB_FROM_XYZ: matvec32(kM_XYZ2RGB, in[0], in[1], in[2], out); break;
            case KH_CIE_LAB_FROM_RGB: {  // kernels.rs:277-283
                float q[3];
                lin_xyz_from_rgb(in, q);
                const float fx = lab_f(q[0] * kInvXN), fy = lab_f(q[1]), fz = lab_f(q[2] * kInvZN);
                out[0] = 116.0f * fy - 16.0f; out[1] = 500.0f * (fx - fy); out[2] = 200.0f * (fy - fz);
                break;
            }
            case KH_CIE_RGB_FROM_LAB: {  // :286-294
                const float fy = (in[0] + 16.0f) / 116.0f, fx = fy + in[1] / 500.0f, fz = fy - in[2] / 200.0f;
                rgb_from_lin_xyz(kXN * lab_finv(fx), 1.0f * lab_finv(fy), kZN * lab_finv(fz), out);
                break;
            }
            case KH_CIE_LUV_FROM_RGB: {  // :297-312
                float q[3];
                lin_xyz_from_rgb(in, q);
                const float yr = q[1];
                const float l = yr > kLabDelta ? 116.0f * cbrt_libm(yr) - 16.0f : kLuvKappa * yr;
                const float d = q[0] + 15.0f * q[1] + 3.0f * q[2];
                const float up = d == 0.0f ? 0.0f : 4.0f * q[0] / d, vp = d == 0.0f ? 0.0f : 9.0f * q[1] / d;
                out[0] = l; out[1] = 13.0f * l * (up - kLuvUn); out[2] = 13.0f * l * (vp - kLuvVn);
                break;
            }
            default: {  // KH_CIE_RGB_FROM_LUV, :315-337
                const float l = in[0];
                if (l <= 0.0f) { rgb_from_lin_xyz(0.0f, 0.0f, 0.0f, out); break; }
                float y;
                if (l > 8.0f) { const float t = (l + 16.0f) / 116.0f; y = 1.0f * t * t * t; }
                else y = 1.0f * l / kLuvKappa;
                const float inv13l = 1.0f / (13.0f * l);
                const float up = in[1] * inv13l + kLuvUn, vp = in[2] * inv13l + kLuvVn;
                const float x = y * 9.0f * up / (4.0f * vp);
                const float z = y * (12.0f - 3.0f * up - 20.0f * vp) / (4.0f * vp);
                rgb_from_lin_xyz(x, y, z, out);
            }
        }
    }
};

}  // namespace

extern "C" {

#define KH_MAP_U8(NAME, CIN, COUT, OP) \
    int32_t NAME(kh_stream_t s, const uint8_t* src, uint8_t* dst, int64_t n) { return launch_u8<CIN, COUT>(s, src, dst, n, OP, #NAME); }
#define KH_MAP_F32(NAME, CIN, COUT, OP) \
    int32_t NAME(kh_stream_t s, const float* src, float* dst, int64_t n) { return launch_f32<CIN, COUT>(s, src, dst, n, OP, #NAME); }

int32_t kh_cie_convert_f32(kh_stream_t s, const float* src, float* dst, int64_t n, int32_t conv) {
    KH_REQUIRE(conv >= KH_CIE_LINEAR_RGB_FROM_RGB && conv <= KH_CIE_RGB_FROM_LUV, KH_ERR_INVALID_ARG,
               "kh_cie_convert_f32: unknown conversion %d", conv);
    return launch_f32<3, 3>(s, src, dst, n, CieF32{conv}, "kh_cie_convert_f32");
}

KH_MAP_U8(kh_gray_from_rgb_u8, 3, 1, GrayFromRgbU8{})
KH_MAP_F32(kh_gray_from_rgb_f32, 3, 1, GrayFromRgbF32{})
KH_MAP_U8(kh_rgb_from_gray_u8, 1, 3, RgbFromGrayU8{})
KH_MAP_F32(kh_rgb_from_gray_f32, 1, 3, RgbFromGrayF32{})
KH_MAP_U8(kh_bgr_from_rgb_u8, 3, 3, BgrFromRgbU8{})
KH_MAP_F32(kh_bgr_from_rgb_f32, 3, 3, BgrFromRgbF32{})
KH_MAP_F32(kh_hsv_from_rgb_f32, 3, 3, HsvFromRgbF32{})
KH_MAP_F32(kh_rgb_from_hsv_f32, 3, 3, RgbFromHsvF32{})
KH_MAP_F32(kh_hls_from_rgb_f32, 3, 3, HlsFromRgbF32{})
KH_MAP_F32(kh_rgb_from_hls_f32, 3, 3, RgbFromHlsF32{})
KH_MAP_U8(kh_sepia_from_rgb_u8, 3, 3, SepiaU8{})
KH_MAP_F32(kh_sepia_from_rgb_f32, 3, 3, SepiaF32{})

int32_t kh_rgba_from_rgb_u8(kh_stream_t s, const uint8_t* src, uint8_t* dst, int64_t n, int32_t swap_rb) {
    return launch_u8<3, 4>(s, src, dst, n, RgbaFromRgbU8{swap_rb != 0}, "kh_rgba_from_rgb_u8");
}
int32_t kh_rgba_from_rgb_f32(kh_stream_t s, const float* src, float* dst, int64_t n, int32_t swap_rb) {
    return launch_f32<3, 4>(s, src, dst, n, RgbaFromRgbF32{swap_rb != 0}, "kh_rgba_from_rgb_f32");
}
int32_t kh_rgb_from_rgba_u8(kh_stream_t s, const uint8_t* src, uint8_t* dst, int64_t n, int32_t swap_rb,
                            const uint8_t* background) {
    RgbFromRgbaU8 op{swap_rb != 0, background != nullptr, 0.f, 0.f, 0.f};
    if (background) { op.bg0 = (float)background[0]; op.bg1 = (float)background[1]; op.bg2 = (float)background[2]; }
    return launch_u8<4, 3>(s, src, dst, n, op, "kh_rgb_from_rgba_u8");
}

static int32_t check_order(int32_t order, const char* what) {
    KH_REQUIRE(order == KH_YCC_YCRCB || order == KH_YCC_YUV, KH_ERR_INVALID_ARG, "%s: unknown chroma order %d", what, order);
    return KH_OK;
}
int32_t kh_ycc_from_rgb_u8(kh_stream_t s, const uint8_t* src, uint8_t* dst, int64_t n, int32_t order) {
    if (int32_t rc = check_order(order, "kh_ycc_from_rgb_u8")) return rc;
    return launch_u8<3, 3>(s, src, dst, n, YccFromRgbU8{order, order == 0 ? 11682 : 14369, order == 0 ? 9241 : 8061},
                           "kh_ycc_from_rgb_u8");
}
int32_t kh_rgb_from_ycc_u8(kh_stream_t s, const uint8_t* src, uint8_t* dst, int64_t n, int32_t order) {
    if (int32_t rc = check_order(order, "kh_rgb_from_ycc_u8")) return rc;
    return launch_u8<3, 3>(s, src, dst, n, RgbFromYccU8{order}, "kh_rgb_from_ycc_u8");
}
int32_t kh_ycc_from_rgb_f32(kh_stream_t s, const float* src, float* dst, int64_t n, int32_t order) {
    if (int32_t rc = check_order(order, "kh_ycc_from_rgb_f32")) return rc;
    return launch_f32<3, 3>(s, src, dst, n, YccFromRgbF32{order, order == 0 ? 0.713f : 0.877f, order == 0 ? 0.564f : 0.492f},
                            "kh_ycc_from_rgb_f32");
}
int32_t kh_rgb_from_ycc_f32(kh_stream_t s, const float* src, float* dst, int64_t n, int32_t order) {
    if (int32_t rc = check_order(order, "kh_rgb_from_ycc_f32")) return rc;
    return launch_f32<3, 3>(s, src, dst, n, RgbFromYccF32{order}, "kh_rgb_from_ycc_f32");
}

int32_t kh_apply_colormap_u8(kh_stream_t s, const uint8_t* src, uint8_t* dst, int64_t n, const uint8_t* lut_device) {
    KH_REQUIRE(lut_device || n == 0, KH_ERR_INVALID_ARG, "kh_apply_colormap_u8: null LUT");
    return launch_u8<1, 3>(s, src, dst, n, ColormapU8{lut_device}, "kh_apply_colormap_u8");
}

int32_t kh_rgb_from_planar420_u8(kh_stream_t s, const uint8_t* src, uint8_t* dst, int32_t w, int32_t h, int32_t layout) {
    if (int32_t rc = check_wh(src, dst, w, h, true, true, "kh_rgb_from_planar420_u8")) return rc;
    KH_REQUIRE(layout >= 0 && layout <= 3, KH_ERR_INVALID_ARG, "kh_rgb_from_planar420_u8: unknown layout %d", layout);
    if ((int64_t)w * h == 0) return KH_OK;
    const bool quad = (w % 4 == 0) && (reinterpret_cast<uintptr_t>(src) % 4 == 0);
    const int dst_aligned = reinterpret_cast<uintptr_t>(dst) % 4 == 0;  // (w*3*row + x*3) % 4 == 0 when w,x % 4 == 0
    if (quad)
        hipLaunchKernelGGL(rgb_from_planar420<4>, dim3(cdiv((int64_t)(w / 4) * (h / 2), kBlock)), dim3(kBlock), 0,
                           as_hip(s), src, dst, w, h, layout, dst_aligned);
    else
        hipLaunchKernelGGL(rgb_from_planar420<2>, dim3(cdiv((int64_t)(w / 2) * (h / 2), kBlock)), dim3(kBlock), 0,
                           as_hip(s), src, dst, w, h, layout, 0);
    return check_launch("kh_rgb_from_planar420_u8");
}

int32_t kh_rgb_from_packed422_u8(kh_stream_t s, const uint8_t* src, uint8_t* dst, int32_t w, int32_t h, int32_t layout) {
    if (int32_t rc = check_wh(src, dst, w, h, true, false, "kh_rgb_from_packed422_u8")) return rc;
    KH_REQUIRE(layout >= 0 && layout <= 2, KH_ERR_INVALID_ARG, "kh_rgb_from_packed422_u8: unknown layout %d", layout);
    const long long groups = (long long)(w / 2) * h;
    if (groups == 0) return KH_OK;
    hipLaunchKernelGGL(rgb_from_packed422, dim3(cdiv(groups, kBlock)), dim3(kBlock), 0, as_hip(s), src, dst, groups,
                       layout, (int)(reinterpret_cast<uintptr_t>(src) % 4 == 0));
    return check_launch("kh_rgb_from_packed422_u8");
}

int32_t kh_nv12_from_rgb_u8(kh_stream_t s, const uint8_t* src, uint8_t* dst, int32_t w, int32_t h) {
    if (int32_t rc = check_wh(src, dst, w, h, true, true, "kh_nv12_from_rgb_u8")) return rc;
    const long long blocks2x2 = (long long)(w / 2) * (h / 2);
    if (blocks2x2 == 0) return KH_OK;
    hipLaunchKernelGGL(nv12_from_rgb, dim3(cdiv(blocks2x2, kBlock)), dim3(kBlock), 0, as_hip(s), src, dst, w, h);
    return check_launch("kh_nv12_from_rgb_u8");
}

int32_t kh_yuyv_from_rgb_u8(kh_stream_t s, const uint8_t* src, uint8_t* dst, int32_t w, int32_t h) {
    if (int32_t rc = check_wh(src, dst, w, h, true, false, "kh_yuyv_from_rgb_u8")) return rc;
    const long long groups = (long long)(w / 2) * h;
    if (groups == 0) return KH_OK;
    hipLaunchKernelGGL(yuyv_from_rgb, dim3(cdiv(groups, kBlock)), dim3(kBlock), 0, as_hip(s), src, dst, groups);
    return check_launch("kh_yuyv_from_rgb_u8");
}

}  // extern "C"
