/*
 * CPU oracle — colour conversions.  TEST INFRASTRUCTURE (see ko_oracle.h).
 *
 * Scalar restatements of the reference's per-pixel colour arithmetic.  Where the reference has
 * SIMD and scalar variants that differ in rounding (f32 gray: AVX2/NEON use fma, the scalar
 * tail and the CUDA kernel use plain mul/add — P/color/gray/kernels.rs:297,321,392,408 and
 * P/cuda/color/gray.rs:56) the oracle is the SCALAR expression, which is also what the
 * reference's device kernels compute.
 */
#include <string.h>

#include "ko_oracle.h"

static inline int iclamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* ---- gray ---------------------------------------------------------------------------------- */

/* P/color/gray/kernels.rs:229-244 (Q14 constants :10-13) */
void ko_gray_from_rgb_u8(const uint8_t* src, uint8_t* dst, size_t npixels) {
    for (size_t i = 0; i < npixels; ++i) {
        size_t si = i * 3;
        dst[i] = (uint8_t)((4899u * src[si] + 9617u * src[si + 1] + 1868u * src[si + 2] + 8192u) >> 14);
    }
}

/* P/color/gray/kernels.rs:405-412 (weights :2-4) */
void ko_gray_from_rgb_f32(const float* src, float* dst, size_t npixels) {
    for (size_t i = 0; i < npixels; ++i) {
        size_t b = i * 3;
        dst[i] = 0.299f * src[b] + 0.587f * src[b + 1] + 0.114f * src[b + 2];
    }
}

/* P/color/gray/kernels.rs:463-469 */
void ko_rgb_from_gray_u8(const uint8_t* src, uint8_t* dst, size_t npixels) {
    for (size_t i = 0; i < npixels; ++i) {
        dst[3 * i] = src[i];
        dst[3 * i + 1] = src[i];
        dst[3 * i + 2] = src[i];
    }
}

void ko_rgb_from_gray_f32(const float* src, float* dst, size_t npixels) {
    for (size_t i = 0; i < npixels; ++i) {
        dst[3 * i] = src[i];
        dst[3 * i + 1] = src[i];
        dst[3 * i + 2] = src[i];
    }
}

/* ---- Family B: BT.601 limited-range Q20 decode (P/color/yuv/kernels.rs:696-736) ------------ */

#define CY 1220542
#define CUB 2116026
#define CUG (-409993)
#define CVG (-852492)
#define CVR 1673527
#define ITUR_HALF (1 << 19)

/* decode_px, :707-722; yy_term, :724-727 */
static inline void decode_px(int y, int u, int v, uint8_t* rgb) {
    int yy = (y - 16 > 0 ? y - 16 : 0) * CY;
    u -= 128;
    v -= 128;
    int b = (yy + CUB * u + ITUR_HALF) >> 20;
    int g = (yy + CUG * u + CVG * v + ITUR_HALF) >> 20;
    int r = (yy + CVR * v + ITUR_HALF) >> 20;
    rgb[0] = (uint8_t)iclamp255(r);
    rgb[1] = (uint8_t)iclamp255(g);
    rgb[2] = (uint8_t)iclamp255(b);
}

/* rgb_from_planar420 + chroma_at, :986-1071, :1195-1216 */
void ko_rgb_from_planar420(const uint8_t* buf, uint8_t* dst, int width, int height, int layout) {
    const uint8_t* y = buf;
    const uint8_t* c0 = buf + (size_t)width * height;
    int cw = width / 2, ch = height / 2;
    const uint8_t* c1 = c0 + (size_t)cw * ch; /* second plane for I420 / YV12 */
#pragma omp parallel for schedule(static)
    for (int cy = 0; cy < ch; ++cy) {
        for (int cx = 0; cx < cw; ++cx) {
            int u, v;
            switch (layout) {
                case 0: u = c0[cy * cw * 2 + cx * 2]; v = c0[cy * cw * 2 + cx * 2 + 1]; break;
                case 1: v = c0[cy * cw * 2 + cx * 2]; u = c0[cy * cw * 2 + cx * 2 + 1]; break;
                case 2: u = c0[cy * cw + cx]; v = c1[cy * cw + cx]; break;
                default: v = c0[cy * cw + cx]; u = c1[cy * cw + cx]; break;
            }
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    size_t row = (size_t)(2 * cy + dy), col = (size_t)(2 * cx + dx);
                    decode_px(y[row * width + col], u, v, dst + (row * width + col) * 3);
                }
        }
    }
}

/* rgb_from_packed422_row_scalar + Packed422::offsets, :945-965, :748-756 */
void ko_rgb_from_packed422(const uint8_t* src, uint8_t* dst, int width, int height, int layout) {
    static const int offs[3][4] = {{0, 1, 2, 3}, {1, 0, 3, 2}, {0, 3, 2, 1}}; /* y0,u,y1,v */
    const int* o = offs[layout];
#pragma omp parallel for schedule(static)
    for (int row = 0; row < height; ++row) {
        const uint8_t* s = src + (size_t)row * width * 2;
        uint8_t* d = dst + (size_t)row * width * 3;
        for (int g = 0; g < width / 2; ++g) {
            const uint8_t* q = s + g * 4;
            decode_px(q[o[0]], q[o[1]], q[o[3]], d + g * 6);
            decode_px(q[o[2]], q[o[1]], q[o[3]], d + g * 6 + 3);
        }
    }
}

/* ---- Family C: BT.601 limited-range Q8 encode (P/color/yuv/kernels.rs:1223-1260) ----------- */

static inline uint8_t encode_y(int r, int g, int b) {
    return (uint8_t)iclamp255(((66 * r + 129 * g + 25 * b + 128) >> 8) + 16);
}

static inline void encode_uv(int r, int g, int b, uint8_t* u, uint8_t* v) {
    *u = (uint8_t)iclamp255(((-38 * r + -74 * g + 112 * b + 128) >> 8) + 128);
    *v = (uint8_t)iclamp255(((112 * r + -94 * g + -18 * b + 128) >> 8) + 128);
}

/* nv12_from_rgb, :1480-1515; encode_uv_row_scalar :1563-1573 */
void ko_nv12_from_rgb(const uint8_t* src, uint8_t* dst, int width, int height) {
    uint8_t* yo = dst;
    uint8_t* uvo = dst + (size_t)width * height;
#pragma omp parallel for schedule(static)
    for (int cy = 0; cy < height / 2; ++cy) {
        const uint8_t* top = src + (size_t)(2 * cy) * width * 3;
        const uint8_t* bot = top + (size_t)width * 3;
        for (int x = 0; x < width; ++x) {
            yo[(size_t)(2 * cy) * width + x] = encode_y(top[3 * x], top[3 * x + 1], top[3 * x + 2]);
            yo[(size_t)(2 * cy + 1) * width + x] = encode_y(bot[3 * x], bot[3 * x + 1], bot[3 * x + 2]);
        }
        for (int cx = 0; cx < width / 2; ++cx) {
            int s = cx * 6;
            int r = top[s] + top[s + 3] + bot[s] + bot[s + 3];
            int g = top[s + 1] + top[s + 4] + bot[s + 1] + bot[s + 4];
            int b = top[s + 2] + top[s + 5] + bot[s + 2] + bot[s + 5];
            encode_uv((r + 2) >> 2, (g + 2) >> 2, (b + 2) >> 2, &uvo[(size_t)cy * width + cx * 2],
                      &uvo[(size_t)cy * width + cx * 2 + 1]);
        }
    }
}

/* yuyv_from_rgb_row_scalar, :1300-1320 */
void ko_yuyv_from_rgb(const uint8_t* src, uint8_t* dst, int width, int height) {
#pragma omp parallel for schedule(static)
    for (int row = 0; row < height; ++row) {
        const uint8_t* s = src + (size_t)row * width * 3;
        uint8_t* d = dst + (size_t)row * width * 2;
        for (int g = 0; g < width / 2; ++g) {
            const uint8_t* q = s + g * 6;
            d[g * 4] = encode_y(q[0], q[1], q[2]);
            d[g * 4 + 2] = encode_y(q[3], q[4], q[5]);
            encode_uv((q[0] + q[3] + 1) >> 1, (q[1] + q[4] + 1) >> 1, (q[2] + q[5] + 1) >> 1,
                      &d[g * 4 + 1], &d[g * 4 + 3]);
        }
    }
}
