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

/* ---- Family A: full-range RGB <-> YCbCr / YUV (P/color/yuv/kernels.rs:23-62) ---------------- */
/* order: 0 = YCrCb (stores [Y, Cr, Cb]); 1 = YuvCbCr (stores [Y, U=Cb, V=Cr]) */

/* ycc_from_rgb_u8_px, :82-98 */
void ko_ycc_from_rgb_u8(const uint8_t* src, uint8_t* dst, size_t npixels, int order) {
    const int c_rv = order == 0 ? 11682 : 14369, c_bu = order == 0 ? 9241 : 8061;
    for (size_t i = 0; i < npixels; ++i) {
        int r = src[3 * i], g = src[3 * i + 1], b = src[3 * i + 2];
        int y = (4899 * r + 9617 * g + 1868 * b + 8192) >> 14;
        int cr = ((r - y) * c_rv + (128 << 14) + 8192) >> 14;
        int cb = ((b - y) * c_bu + (128 << 14) + 8192) >> 14;
        dst[3 * i] = (uint8_t)iclamp255(y);
        if (order == 0) { dst[3 * i + 1] = (uint8_t)iclamp255(cr); dst[3 * i + 2] = (uint8_t)iclamp255(cb); }
        else            { dst[3 * i + 1] = (uint8_t)iclamp255(cb); dst[3 * i + 2] = (uint8_t)iclamp255(cr); }
    }
}

/* rgb_from_ycc_u8_px, :100-126 */
void ko_rgb_from_ycc_u8(const uint8_t* src, uint8_t* dst, size_t npixels, int order) {
    for (size_t i = 0; i < npixels; ++i) {
        int y = src[3 * i];
        int cr = (order == 0 ? src[3 * i + 1] : src[3 * i + 2]) - 128;
        int cb = (order == 0 ? src[3 * i + 2] : src[3 * i + 1]) - 128;
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
        dst[3 * i] = (uint8_t)iclamp255(r);
        dst[3 * i + 1] = (uint8_t)iclamp255(g);
        dst[3 * i + 2] = (uint8_t)iclamp255(b);
    }
}

/* ycc_from_rgb_f32_px, :541-551 */
void ko_ycc_from_rgb_f32(const float* src, float* dst, size_t npixels, int order) {
    const float k_rv = order == 0 ? 0.713f : 0.877f, k_bu = order == 0 ? 0.564f : 0.492f;
    for (size_t i = 0; i < npixels; ++i) {
        float r = src[3 * i], g = src[3 * i + 1], b = src[3 * i + 2];
        float y = 0.299f * r + 0.587f * g + 0.114f * b;
        float cr = (r - y) * k_rv + 0.5f;
        float cb = (b - y) * k_bu + 0.5f;
        dst[3 * i] = y;
        if (order == 0) { dst[3 * i + 1] = cr; dst[3 * i + 2] = cb; }
        else            { dst[3 * i + 1] = cb; dst[3 * i + 2] = cr; }
    }
}

/* rgb_from_ycc_f32_px, :673-690 */
void ko_rgb_from_ycc_f32(const float* src, float* dst, size_t npixels, int order) {
    for (size_t i = 0; i < npixels; ++i) {
        float y = src[3 * i];
        float cr = order == 0 ? src[3 * i + 1] : src[3 * i + 2];
        float cb = order == 0 ? src[3 * i + 2] : src[3 * i + 1];
        float r, g, b;
        if (order == 0) {
            r = y + (cr - 0.5f) / 0.713f;
            b = y + (cb - 0.5f) / 0.564f;
            g = (y - 0.299f * r - 0.114f * b) / 0.587f;
        } else {
            r = y + 1.140f * (cr - 0.5f);
            g = y + -0.395f * (cb - 0.5f) + -0.581f * (cr - 0.5f);
            b = y + 2.032f * (cb - 0.5f);
        }
        dst[3 * i] = r; dst[3 * i + 1] = g; dst[3 * i + 2] = b;
    }
}

/* ---- HSV / HLS, 0..255 domain (P/color/hsv/kernels.rs:150-177,310-334; hls/kernels.rs:159-187,357-391) */
#include <math.h>
static const float INV_255 = 1.0f / 255.0f;
static const float DEG_TO_BYTE = 255.0f / 360.0f;
static const float BYTE_TO_DEG = 360.0f / 255.0f;
static inline float fmax3(float a, float b, float c) { float m = a > b ? a : b; return m > c ? m : c; }
static inline float fmin3(float a, float b, float c) { float m = a < b ? a : b; return m < c ? m : c; }

void ko_hsv_from_rgb_f32(const float* src, float* dst, size_t npixels) {
    for (size_t i = 0; i < npixels; ++i) {
        float r = src[3 * i] * INV_255, g = src[3 * i + 1] * INV_255, b = src[3 * i + 2] * INV_255;
        float max = fmax3(r, g, b), min = fmin3(r, g, b), delta = max - min;
        float h;
        if (delta == 0.0f) h = 0.0f;
        else if (max == r) h = 60.0f * fmodf((g - b) / delta, 6.0f);
        else if (max == g) h = 60.0f * (((b - r) / delta) + 2.0f);
        else h = 60.0f * (((r - g) / delta) + 4.0f);
        if (h < 0.0f) h = h + 360.0f;
        dst[3 * i] = h * DEG_TO_BYTE;
        dst[3 * i + 1] = max == 0.0f ? 0.0f : (delta / max) * 255.0f;
        dst[3 * i + 2] = max * 255.0f;
    }
}

void ko_rgb_from_hsv_f32(const float* src, float* dst, size_t npixels) {
    for (size_t i = 0; i < npixels; ++i) {
        float s = src[3 * i + 1] * INV_255, v = src[3 * i + 2] * INV_255;
        float hh = src[3 * i] * (BYTE_TO_DEG / 60.0f);
        float c = v * s;
        float hmod2 = hh - 2.0f * floorf(hh * 0.5f);
        float x = c * (1.0f - fabsf(hmod2 - 1.0f));
        float m = v - c;
        int sext = (int)floorf(hh);
        float r1, g1, b1;
        switch (sext) {
            case 0: r1 = c; g1 = x; b1 = 0.0f; break;
            case 1: r1 = x; g1 = c; b1 = 0.0f; break;
            case 2: r1 = 0.0f; g1 = c; b1 = x; break;
            case 3: r1 = 0.0f; g1 = x; b1 = c; break;
            case 4: r1 = x; g1 = 0.0f; b1 = c; break;
            default: r1 = c; g1 = 0.0f; b1 = x; break;
        }
        dst[3 * i] = (r1 + m) * 255.0f; dst[3 * i + 1] = (g1 + m) * 255.0f; dst[3 * i + 2] = (b1 + m) * 255.0f;
    }
}

void ko_hls_from_rgb_f32(const float* src, float* dst, size_t npixels) {
    for (size_t i = 0; i < npixels; ++i) {
        float r = src[3 * i] * INV_255, g = src[3 * i + 1] * INV_255, b = src[3 * i + 2] * INV_255;
        float max = fmax3(r, g, b), min = fmin3(r, g, b), diff = max - min, sum = max + min;
        float l = sum * 0.5f, h = 0.0f, s = 0.0f;
        if (diff != 0.0f) {
            s = l <= 0.5f ? diff / sum : diff / (2.0f - sum);
            if (max == r) h = 60.0f * fmodf((g - b) / diff, 6.0f);
            else if (max == g) h = 60.0f * (((b - r) / diff) + 2.0f);
            else h = 60.0f * (((r - g) / diff) + 4.0f);
            if (h < 0.0f) h = h + 360.0f;
        }
        dst[3 * i] = h * DEG_TO_BYTE; dst[3 * i + 1] = l * 255.0f; dst[3 * i + 2] = s * 255.0f;
    }
}

static inline float hue2rgb(float p, float q, float t) {
    if (t < 0.0f) t = t + 1.0f;
    if (t > 1.0f) t = t - 1.0f;
    if (t < 1.0f / 6.0f) return p + (q - p) * 6.0f * t;
    if (t < 0.5f) return q;
    if (t < 2.0f / 3.0f) return p + (q - p) * (2.0f / 3.0f - t) * 6.0f;
    return p;
}

void ko_rgb_from_hls_f32(const float* src, float* dst, size_t npixels) {
    for (size_t i = 0; i < npixels; ++i) {
        float l = src[3 * i + 1] * INV_255, s = src[3 * i + 2] * INV_255;
        if (s == 0.0f) {
            float v = l * 255.0f;
            dst[3 * i] = v; dst[3 * i + 1] = v; dst[3 * i + 2] = v;
            continue;
        }
        float h_deg = src[3 * i] * BYTE_TO_DEG;
        float q = l < 0.5f ? l * (1.0f + s) : l + s - l * s;
        float p = 2.0f * l - q;
        float hk = h_deg / 360.0f;
        dst[3 * i] = hue2rgb(p, q, hk + 1.0f / 3.0f) * 255.0f;
        dst[3 * i + 1] = hue2rgb(p, q, hk) * 255.0f;
        dst[3 * i + 2] = hue2rgb(p, q, hk - 1.0f / 3.0f) * 255.0f;
    }
}

/* ---- swizzles (P/color/rgb/mod.rs:128-316, rgb/kernels.rs) ---------------------------------- */
void ko_bgr_from_rgb_u8(const uint8_t* src, uint8_t* dst, size_t n) {
    for (size_t i = 0; i < n; ++i) { dst[3 * i] = src[3 * i + 2]; dst[3 * i + 1] = src[3 * i + 1]; dst[3 * i + 2] = src[3 * i]; }
}
void ko_bgr_from_rgb_f32(const float* src, float* dst, size_t n) {
    for (size_t i = 0; i < n; ++i) { dst[3 * i] = src[3 * i + 2]; dst[3 * i + 1] = src[3 * i + 1]; dst[3 * i + 2] = src[3 * i]; }
}
/* swap_rb = 0: rgba_from_rgb; 1: bgra_from_rgb; alpha = 255 / 1.0 (rgb/kernels.rs:177-186,236-245) */
void ko_rgba_from_rgb_u8(const uint8_t* src, uint8_t* dst, size_t n, int swap_rb) {
    for (size_t i = 0; i < n; ++i) {
        dst[4 * i] = src[3 * i + (swap_rb ? 2 : 0)]; dst[4 * i + 1] = src[3 * i + 1];
        dst[4 * i + 2] = src[3 * i + (swap_rb ? 0 : 2)]; dst[4 * i + 3] = 255;
    }
}
void ko_rgba_from_rgb_f32(const float* src, float* dst, size_t n, int swap_rb) {
    for (size_t i = 0; i < n; ++i) {
        dst[4 * i] = src[3 * i + (swap_rb ? 2 : 0)]; dst[4 * i + 1] = src[3 * i + 1];
        dst[4 * i + 2] = src[3 * i + (swap_rb ? 0 : 2)]; dst[4 * i + 3] = 1.0f;
    }
}
/* rgb_from_rgba / rgb_from_bgra with optional background blend (rgb/mod.rs:128-227, alpha_blend :311-316) */
void ko_rgb_from_rgba_u8(const uint8_t* src, uint8_t* dst, size_t n, int swap_rb, const uint8_t* bg) {
    for (size_t i = 0; i < n; ++i) {
        uint8_t r = src[4 * i + (swap_rb ? 2 : 0)], g = src[4 * i + 1], b = src[4 * i + (swap_rb ? 0 : 2)];
        if (bg) {
            float alpha = (float)src[4 * i + 3] / 255.0f;
            dst[3 * i] = (uint8_t)roundf((float)r * alpha + (float)bg[0] * (1.0f - alpha));
            dst[3 * i + 1] = (uint8_t)roundf((float)g * alpha + (float)bg[1] * (1.0f - alpha));
            dst[3 * i + 2] = (uint8_t)roundf((float)b * alpha + (float)bg[2] * (1.0f - alpha));
        } else {
            dst[3 * i] = r; dst[3 * i + 1] = g; dst[3 * i + 2] = b;
        }
    }
}

/* ---- sepia (P/color/sepia.rs:17-22,86-104; matrix.rs:122-136) and colormap LUT (colormap.rs:115-122) */
void ko_sepia_from_rgb_f32(const float* src, float* dst, size_t n) {
    static const float m[9] = {0.393f, 0.769f, 0.189f, 0.349f, 0.686f, 0.168f, 0.272f, 0.534f, 0.131f};
    for (size_t i = 0; i < n; ++i) {
        float c0 = src[3 * i], c1 = src[3 * i + 1], c2 = src[3 * i + 2];
        dst[3 * i] = 0.0f + m[0] * c0 + m[1] * c1 + m[2] * c2;
        dst[3 * i + 1] = 0.0f + m[3] * c0 + m[4] * c1 + m[5] * c2;
        dst[3 * i + 2] = 0.0f + m[6] * c0 + m[7] * c1 + m[8] * c2;
    }
}
void ko_sepia_from_rgb_u8(const uint8_t* src, uint8_t* dst, size_t n) {
    static const uint32_t q[9] = {101, 197, 48, 89, 176, 43, 70, 137, 34};
    for (size_t i = 0; i < n; ++i) {
        uint32_t r = src[3 * i], g = src[3 * i + 1], b = src[3 * i + 2];
        for (int c = 0; c < 3; ++c) {
            uint32_t v = (q[3 * c] * r + q[3 * c + 1] * g + q[3 * c + 2] * b + 128) >> 8;
            dst[3 * i + c] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}
/* lut = r[256] g[256] b[256] */
void ko_apply_colormap_u8(const uint8_t* src, uint8_t* dst, size_t n, const uint8_t* lut) {
    for (size_t i = 0; i < n; ++i) {
        dst[3 * i] = lut[src[i]]; dst[3 * i + 1] = lut[256 + src[i]]; dst[3 * i + 2] = lut[512 + src[i]];
    }
}

/* ---- convert_yuyv_to_rgb_u8 (P/color/yuv/mod.rs:319-480): YUYV -> RGB8 with a selectable Q10 matrix -------------
 * mode 0 = Bt601Full, 1 = Bt709Full, 2 = Bt601Limited.  The row is walked in whole 6-byte RGB chunks (:374-376): with
 * an odd width the last pixel of every row is left as it was.  Rust `>>` on i32 is arithmetic; so is gcc's.         */
static inline uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

static inline void yuv_px_mode(int mode, uint8_t y, uint8_t u, uint8_t v, uint8_t* rgb) {
    int y_val = (int)y, u_val = (int)u - 128, v_val = (int)v - 128, r, g, b;
    switch (mode) {
        case 0:
            r = y_val + ((1436 * v_val + 512) >> 10);
            g = y_val - ((352 * u_val + 731 * v_val + 512) >> 10);
            b = y_val + ((1815 * u_val + 512) >> 10);
            break;
        case 1:
            r = y_val + ((1612 * v_val + 512) >> 10);
            g = y_val - ((192 * u_val + 479 * v_val + 512) >> 10);
            b = y_val + ((1900 * u_val + 512) >> 10);
            break;
        default:
            y_val = (((int)y - 16) * 1192 + 512) >> 10;
            r = y_val + ((1634 * v_val + 512) >> 10);
            g = y_val - ((401 * u_val + 832 * v_val + 512) >> 10);
            b = y_val + ((2066 * u_val + 512) >> 10);
    }
    rgb[0] = sat8(r); rgb[1] = sat8(g); rgb[2] = sat8(b);
}

int ko_yuyv_to_rgb_mode(const uint8_t* src, uint8_t* dst, int w, int h, int mode) {
    if (mode < 0 || mode > 2 || w < 0 || h < 0) return -1;
    for (int row = 0; row < h; ++row) {
        const uint8_t* s = src + (size_t)row * w * 2;
        uint8_t* d = dst + (size_t)row * w * 3;
        for (int col = 0; col < (w * 3) / 6; ++col) {
            const uint8_t* q = s + 4 * col;
            yuv_px_mode(mode, q[0], q[1], q[3], d + 6 * col);
            yuv_px_mode(mode, q[2], q[1], q[3], d + 6 * col + 3);
        }
    }
    return 0;
}

/* ---- rgb_from_bayer (P/color/bayer/mod.rs:37-70; kernels.rs:30-200): bilinear demosaic, u8 -------------------------
 * pattern 0 RGGB, 1 BGGR, 2 GRBG, 3 GBRG.  demosaic_px over replicate-clamped neighbours for every pixel, then
 * bayer_border_replicate: row 0 <- row 1 and last row <- the row above it (when rows >= 3), then column 0 <- column 1 and
 * last column <- the one before it (when cols >= 3).                                                                  */
enum { CELL_R = 0, CELL_G_ON_R = 1, CELL_G_ON_B = 2, CELL_B = 3 };
static const int BAYER_PHASE[4][2][2] = {
    {{CELL_R, CELL_G_ON_R}, {CELL_G_ON_B, CELL_B}},   /* Rggb */
    {{CELL_B, CELL_G_ON_B}, {CELL_G_ON_R, CELL_R}},   /* Bggr */
    {{CELL_G_ON_R, CELL_R}, {CELL_B, CELL_G_ON_B}},   /* Grbg */
    {{CELL_G_ON_B, CELL_B}, {CELL_R, CELL_G_ON_R}},   /* Gbrg */
};
static inline uint8_t bayer_at(const uint8_t* src, long r, long c, int rows, int cols) {
    long rr = r < 0 ? 0 : (r > rows - 1 ? rows - 1 : r), cc = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c);
    return src[rr * cols + cc];
}
static inline uint8_t avg2u(uint8_t a, uint8_t b) { return (uint8_t)(((unsigned)a + b + 1u) >> 1); }
static inline uint8_t avg4u(uint8_t a, uint8_t b, uint8_t c, uint8_t d) { return (uint8_t)(((unsigned)a + b + c + d + 2u) >> 2); }

int ko_rgb_from_bayer(const uint8_t* src, uint8_t* dst, int cols, int rows, int pattern) {
    if (pattern < 0 || pattern > 3 || rows < 0 || cols < 0) return -1;
    if (rows == 0 || cols == 0) return 0;
    for (long r = 0; r < rows; ++r)
        for (long c = 0; c < cols; ++c) {
            uint8_t center = src[r * cols + c], red, green, blue;
            switch (BAYER_PHASE[pattern][r & 1][c & 1]) {
                case CELL_R:
                    green = avg4u(bayer_at(src, r - 1, c, rows, cols), bayer_at(src, r + 1, c, rows, cols), bayer_at(src, r, c - 1, rows, cols), bayer_at(src, r, c + 1, rows, cols));
                    blue = avg4u(bayer_at(src, r - 1, c - 1, rows, cols), bayer_at(src, r - 1, c + 1, rows, cols), bayer_at(src, r + 1, c - 1, rows, cols), bayer_at(src, r + 1, c + 1, rows, cols));
                    red = center;
                    break;
                case CELL_B:
                    green = avg4u(bayer_at(src, r - 1, c, rows, cols), bayer_at(src, r + 1, c, rows, cols), bayer_at(src, r, c - 1, rows, cols), bayer_at(src, r, c + 1, rows, cols));
                    red = avg4u(bayer_at(src, r - 1, c - 1, rows, cols), bayer_at(src, r - 1, c + 1, rows, cols), bayer_at(src, r + 1, c - 1, rows, cols), bayer_at(src, r + 1, c + 1, rows, cols));
                    blue = center;
                    break;
                case CELL_G_ON_R:
                    red = avg2u(bayer_at(src, r, c - 1, rows, cols), bayer_at(src, r, c + 1, rows, cols));
                    blue = avg2u(bayer_at(src, r - 1, c, rows, cols), bayer_at(src, r + 1, c, rows, cols));
                    green = center;
                    break;
                default:
                    blue = avg2u(bayer_at(src, r, c - 1, rows, cols), bayer_at(src, r, c + 1, rows, cols));
                    red = avg2u(bayer_at(src, r - 1, c, rows, cols), bayer_at(src, r + 1, c, rows, cols));
                    green = center;
            }
            uint8_t* o = dst + (r * cols + c) * 3;
            o[0] = red; o[1] = green; o[2] = blue;
        }
    size_t w = (size_t)cols * 3;
    if (rows >= 3) {
        memcpy(dst, dst + w, w);
        memcpy(dst + (size_t)(rows - 1) * w, dst + (size_t)(rows - 2) * w, w);
    }
    if (cols >= 3)
        for (long r = 0; r < rows; ++r) {
            uint8_t* row = dst + (size_t)r * w;
            memcpy(row, row + 3, 3);
            memcpy(row + (size_t)(cols - 1) * 3, row + (size_t)(cols - 2) * 3, 3);
        }
    return 0;
}
