/*
 * CPU oracle — Gaussian pyramid (pyrdown / pyrup, f32 and u8) and u8 morphology (dilate / erode).
 * TEST INFRASTRUCTURE (see ko_oracle.h).
 *
 * P/pyramid.rs:22-250 (pyrup_f32: [1 6 1]/8 even, [1 1]/2 odd, special border rows/columns),
 * :252-430 (reflect_101, pyrdown_f32: 5x5 outer-product taps, ky-major accumulation), :469-655
 * (pyrdown_u8: [1 4 6 4 1] to u16 then (sum+128)>>8), :656-840 (pyrup_u8: (p+6c+n+4)>>3 / (c+n+1)>>1
 * per axis with a u8 intermediate), P/morphology/ops.rs:22-210 + P/padding.rs:32-80 (max / min over
 * the active taps of the padded image; dilate starts from T::default()).
 */
#include <stdlib.h>
#include <string.h>

#include "ko_oracle.h"

static inline int reflect_101(int p, int len) { /* pyramid.rs:252-270 */
    if (len == 1) return 0;
    if (p < 0) p = -p;
    const int period = 2 * (len - 1);
    p %= period;
    if (p >= len) p = period - p;
    return p;
}

/* ---- f32 ------------------------------------------------------------------------------------------ */
void ko_pyrdown_f32(const float* src, int sw, int sh, float* dst, int C) {
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    const float k1[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    float kw[25];
    for (int y = 0, i = 0; y < 5; ++y)
        for (int x = 0; x < 5; ++x) kw[i++] = k1[y] * k1[x];
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x)
            for (int c = 0; c < C; ++c) {
                float sum = 0.0f;
                for (int ky = 0, i = 0; ky < 5; ++ky) {
                    const int sy = reflect_101(2 * y + ky - 2, sh);
                    for (int kx = 0; kx < 5; ++kx, ++i) {
                        const int sx = reflect_101(2 * x + kx - 2, sw);
                        sum += src[((size_t)sy * sw + sx) * C + c] * kw[i];
                    }
                }
                dst[((size_t)y * dw + x) * C + c] = sum;
            }
}

/* pyrup_horizontal_pass_f32, :22-96: value of the 2*sw-wide intermediate at column X of source row `row` */
static inline float pyrup_h(const float* row, int sw, int C, int X, int c) {
    if (sw == 1) return row[c];
    const int x = X >> 1, odd = X & 1;
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

void ko_pyrup_f32(const float* src, int sw, int sh, float* dst, int C) {
    const int dw = 2 * sw, dh = 2 * sh;
#pragma omp parallel for schedule(static)
    for (int Y = 0; Y < dh; ++Y) {
        const int y = Y >> 1, odd = Y & 1;
        int rt, rc, rb; /* pyrup_vertical_pass_f32, :98-170 */
        if (sh == 1) { rt = rc = rb = 0; }
        else if (y == 0) { rt = 0; rc = 0; rb = 1; }
        else if (y == sh - 1) { rt = sh - 2; rc = sh - 1; rb = sh - 1; }
        else { rt = y - 1; rc = y; rb = y + 1; }
        for (int X = 0; X < dw; ++X)
            for (int c = 0; c < C; ++c) {
                const float top = pyrup_h(src + (size_t)rt * sw * C, sw, C, X, c);
                const float cen = pyrup_h(src + (size_t)rc * sw * C, sw, C, X, c);
                const float bot = pyrup_h(src + (size_t)rb * sw * C, sw, C, X, c);
                float v;
                if (y == 0) v = odd ? (cen + bot) * 0.5f : (6.0f * cen + 2.0f * bot) * 0.125f;
                else if (y == sh - 1) v = odd ? cen : (1.0f * top + 7.0f * cen) * 0.125f;
                else v = odd ? (cen + bot) * 0.5f : (1.0f * top + 6.0f * cen + 1.0f * bot) * 0.125f;
                dst[((size_t)Y * dw + X) * C + c] = v;
            }
    }
}

/* ---- u8 ------------------------------------------------------------------------------------------- */
void ko_pyrdown_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int C) {
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    uint16_t* buf = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)dw * sh * C);
    for (int y = 0; y < sh; ++y)
        for (int x = 0; x < dw; ++x) {
            int idx[5];
            for (int t = 0; t < 5; ++t) idx[t] = reflect_101(2 * x + t - 2, sw) * C;
            for (int k = 0; k < C; ++k) {
                const uint8_t* r = src + (size_t)y * sw * C + k;
                buf[((size_t)y * dw + x) * C + k] = (uint16_t)(r[idx[0]] + 4 * r[idx[1]] + 6 * r[idx[2]] + 4 * r[idx[3]] + r[idx[4]]);
            }
        }
    const size_t stride = (size_t)dw * C;
    for (int y = 0; y < dh; ++y) {
        size_t off[5];
        for (int t = 0; t < 5; ++t) off[t] = (size_t)reflect_101(2 * y + t - 2, sh) * stride;
        for (size_t i = 0; i < stride; ++i) {
            uint32_t sum = (uint32_t)buf[off[0] + i] + 4u * buf[off[1] + i] + 6u * buf[off[2] + i] + 4u * buf[off[3] + i] + buf[off[4] + i];
            uint32_t v = (sum + 128u) >> 8;
            dst[(size_t)y * stride + i] = (uint8_t)(v < 255u ? v : 255u);
        }
    }
    free(buf);
}

void ko_pyrup_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int C) {
    const int dw = 2 * sw;
    const size_t stride = (size_t)dw * C;
    uint8_t* buf = (uint8_t*)malloc(stride * sh);
    for (int y = 0; y < sh; ++y)
        for (int x = 0; x < sw; ++x) {
            const int ip = reflect_101(x - 1, sw) * C, in = reflect_101(x + 1, sw) * C;
            for (int k = 0; k < C; ++k) {
                const uint8_t* r = src + (size_t)y * sw * C + k;
                const unsigned pc = r[x * C], pp = r[ip], pn = r[in];
                buf[(size_t)y * stride + (size_t)(2 * x) * C + k] = (uint8_t)((pp + 6 * pc + pn + 4) >> 3);
                buf[(size_t)y * stride + (size_t)(2 * x + 1) * C + k] = (uint8_t)((pc + pn + 1) >> 1);
            }
        }
    for (int y = 0; y < sh; ++y) {
        const size_t op = (size_t)reflect_101(y - 1, sh) * stride, oc = (size_t)y * stride, on = (size_t)reflect_101(y + 1, sh) * stride;
        for (size_t i = 0; i < stride; ++i) {
            const unsigned pc = buf[oc + i], pp = buf[op + i], pn = buf[on + i];
            dst[(size_t)(2 * y) * stride + i] = (uint8_t)((pp + 6 * pc + pn + 4) >> 3);
            dst[(size_t)(2 * y + 1) * stride + i] = (uint8_t)((pc + pn + 1) >> 1);
        }
    }
    free(buf);
}

/* ---- morphology ----------------------------------------------------------------------------------- */
/* Kernel::new, P/morphology/kernels.rs:113-185; shape 0 box, 1 cross, 2 ellipse */
void ko_morph_kernel(int shape, int width, int height, uint8_t* out) {
    if (shape == 0) { memset(out, 1, (size_t)width * height); return; }
    memset(out, 0, (size_t)width * height);
    if (shape == 1) {
        const int size = width, mid = size / 2;
        for (int j = 0; j < size; ++j) out[mid * size + j] = 1;
        for (int i = 0; i < size; ++i) out[i * size + mid] = 1;
        return;
    }
    const float cx = (float)width / 2.0f, cy = (float)height / 2.0f, rx = cx, ry = cy;
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j) {
            const float x = (float)j - cx, y = (float)i - cy;
            if ((x * x) / (rx * rx) + (y * y) / (ry * ry) <= 1.0f) out[i * width + j] = 1;
        }
}

/* PaddingMode::map_index, P/padding.rs:32-80; mode 0 constant (-1 = outside), 1 replicate,
 * 2 reflect101, 3 reflect, 4 wrap */
static int map_index(int mode, long i, int len) {
    if (i >= 0 && i < len) return (int)i;
    switch (mode) {
        case 1: return i < 0 ? 0 : len - 1;
        case 2:
            if (len == 1) return 0;
            while (i < 0 || i >= len) i = i < 0 ? -i : 2L * len - i - 2;
            return (int)i;
        case 3:
            if (len == 1) return 0;
            while (i < 0 || i >= len) i = i < 0 ? -i - 1 : 2L * len - i - 1;
            return (int)i;
        case 4: return (int)(((i % len) + len) % len);
        default: return -1;
    }
}

/* op 0 dilate (max, from 0), 1 erode (min over the active taps; 0 when there is none) */
void ko_morphology_u8(const uint8_t* src, int w, int h, int C, uint8_t* dst, int op, const uint8_t* mask, int kw, int kh,
                      int border, const uint8_t* cval) {
    const int pad_h = kh / 2, pad_w = kw / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < C; ++c) {
                int acc = op == 0 ? 0 : -1;
                for (int ky = 0; ky < kh; ++ky)
                    for (int kx = 0; kx < kw; ++kx) {
                        if (mask[ky * kw + kx] != 1) continue;
                        const int sy = map_index(border, (long)y + ky - pad_h, h), sx = map_index(border, (long)x + kx - pad_w, w);
                        const int v = (sy < 0 || sx < 0) ? cval[c] : src[((size_t)sy * w + sx) * C + c];
                        if (op == 0) acc = v > acc ? v : acc;
                        else acc = (acc < 0 || v < acc) ? v : acc;
                    }
                dst[((size_t)y * w + x) * C + c] = (uint8_t)(acc < 0 ? 0 : acc);
            }
}
