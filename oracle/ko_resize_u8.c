/*
 * CPU oracle — u8 resize cascade (resize_fast_u8_aa) and the OpenCV-compat resize.
 * TEST INFRASTRUCTURE (see ko_oracle.h).
 *
 * P/resize/mod.rs:254-440 (routing, resize_u8_path), P/resize/pyramid.rs:17-112 +
 * P/resize/kernels.rs:62-74,166-183,272-281 (exact-2x RGB box / 75-25 paths),
 * P/resize/nearest.rs:18-72, P/resize/bilinear.rs:25-104 + P/resize/kernels.rs:1141-1165 (Q14
 * bilinear, f64 coordinates), P/resize/common.rs:11-144 + P/resize/separable.rs +
 * P/resize/kernels.rs:403-425,699-708 (Q14 separable bicubic / lanczos, optional antialias),
 * P/resize/opencv_compat.rs:22-330 (cv2-compatible nearest / linear for u8 and f32).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ko_oracle.h"

static inline long long llclamp(long long v, long long lo, long long hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---- exact-2x RGB fast paths ------------------------------------------------------------------- */
static void pyrdown_2x_rgb(const uint8_t* src, uint8_t* dst, int sw, int sh) {
    const int dw = sw / 2, dh = sh / 2;
    for (int y = 0; y < dh; ++y) {
        const uint8_t *r0 = src + (size_t)(2 * y) * sw * 3, *r1 = r0 + (size_t)sw * 3;
        for (int x = 0; x < dw; ++x)
            for (int ch = 0; ch < 3; ++ch) {
                unsigned sum = r0[(2 * x) * 3 + ch] + r0[(2 * x + 1) * 3 + ch] + r1[(2 * x) * 3 + ch] + r1[(2 * x + 1) * 3 + ch];
                dst[((size_t)y * dw + x) * 3 + ch] = (uint8_t)((sum + 2) >> 2);
            }
    }
}

static void hinterp_row_rgb(const uint8_t* src, uint8_t* dst, int sw) {
    memcpy(dst, src, 3);
    for (int j = 0; j < sw - 1; ++j)
        for (int ch = 0; ch < 3; ++ch) {
            unsigned a = src[j * 3 + ch], b = src[(j + 1) * 3 + ch], avg = (a + b + 1) >> 1;
            dst[(2 * j + 1) * 3 + ch] = (uint8_t)((a + avg + 1) >> 1);
            dst[(2 * j + 2) * 3 + ch] = (uint8_t)((b + avg + 1) >> 1);
        }
    memcpy(dst + (size_t)(2 * sw - 1) * 3, src + (size_t)(sw - 1) * 3, 3);
}
static void blend_75_25(const uint8_t* a, const uint8_t* b, uint8_t* dst, int n) {
    for (int i = 0; i < n; ++i) {
        unsigned av = a[i], bv = b[i], avg = (av + bv + 1) >> 1;
        dst[i] = (uint8_t)((av + avg + 1) >> 1);
    }
}
static void pyrup_2x_rgb(const uint8_t* src, uint8_t* dst, int sw, int sh) {
    const int n = 2 * sw * 3;
    uint8_t *ha = (uint8_t*)malloc((size_t)n), *hb = (uint8_t*)malloc((size_t)n);
    hinterp_row_rgb(src, dst, sw);
    hinterp_row_rgb(src + (size_t)(sh - 1) * sw * 3, dst + (size_t)(2 * sh - 1) * n, sw);
    hinterp_row_rgb(src, ha, sw);
    for (int i = 0; i < sh - 1; ++i) {
        hinterp_row_rgb(src + (size_t)(i + 1) * sw * 3, hb, sw);
        blend_75_25(ha, hb, dst + (size_t)(2 * i + 1) * n, n);
        blend_75_25(hb, ha, dst + (size_t)(2 * i + 2) * n, n);
        uint8_t* t = ha; ha = hb; hb = t;
    }
    free(ha); free(hb);
}

/* ---- nearest (nearest.rs:18-21) and Q14 bilinear taps (bilinear.rs:25-39) ------------------------ */
static inline int nearest_index(int i, double scale, int src_len) {
    double v = floor(((double)i + 0.5) * scale);
    return (int)llclamp((long long)v, 0, src_len - 1);
}
static inline void bilinear_tap(int i, double scale, int src_len, unsigned* ofs, unsigned* fq) {
    double s = ((double)i + 0.5) * scale - 0.5;
    long long i0 = (long long)floor(s);
    double f = s - (double)i0;
    if (i0 < 0) { i0 = 0; f = 0.0; }
    else if (i0 >= (long long)src_len - 1) { i0 = (long long)src_len - 2; f = 1.0; }
    unsigned q = (unsigned)round(f * 16384.0);
    *ofs = (unsigned)i0;
    *fq = q > 16384u ? 16384u : q;
}

static void resize_nearest_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C) {
    const double sx = (double)sw / (double)dw, sy = (double)sh / (double)dh;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        const uint8_t* row = src + (size_t)nearest_index(y, sy, sh) * sw * C;
        for (int x = 0; x < dw; ++x) memcpy(dst + ((size_t)y * dw + x) * C, row + (size_t)nearest_index(x, sx, sw) * C, C);
    }
}

static void resize_bilinear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C) {
    const double scx = (double)sw / (double)dw, scy = (double)sh / (double)dh;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        unsigned yi, fy;
        bilinear_tap(y, scy, sh, &yi, &fy);
        const uint64_t fy1 = 16384u - fy;
        const uint8_t *row0 = src + (size_t)yi * sw * C, *row1 = row0 + (size_t)sw * C;
        for (int x = 0; x < dw; ++x) {
            unsigned xi, fx;
            bilinear_tap(x, scx, sw, &xi, &fx);
            const uint64_t fx1 = 16384u - fx;
            for (int ch = 0; ch < C; ++ch) { /* bilinear_row_u8_scalar, kernels.rs:1141-1165 */
                uint64_t p00 = row0[xi * C + ch], p01 = row0[xi * C + C + ch], p10 = row1[xi * C + ch], p11 = row1[xi * C + C + ch];
                uint64_t top = p00 * fx1 + p01 * fx, bot = p10 * fx1 + p11 * fx;
                dst[((size_t)y * dw + x) * C + ch] = (uint8_t)((top * fy1 + bot * fy + (1ull << 27)) >> 28);
            }
        }
    }
}

/* ---- separable Q14 (common.rs:11-144) ---------------------------------------------------------- */
static double filt_support(int filt) { return filt == 0 ? 2.0 : 3.0; }
static double filt_weight(int filt, double x) { /* 0 = Cubic (a = -0.5), 1 = Lanczos3 */
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

/* precompute_contribs: offsets[dst], weights[dst * ksize] (Q14, sum forced to 16384); returns ksize */
int ko_resize_contribs(int src_size, int dst_size, int filt, int antialias, int32_t* offsets, int32_t* weights, int max_ksize) {
    const double scale = (double)src_size / (double)dst_size;
    const double filt_scale = antialias ? (scale > 1.0 ? scale : 1.0) : 1.0;
    const double support = filt_support(filt) * filt_scale;
    int ksize = (int)ceil(support) * 2;
    if (ksize < 2) ksize = 2;
    if (!offsets) return ksize;
    if (ksize > max_ksize) return -ksize;
    double* raw = (double*)malloc(sizeof(double) * (size_t)ksize);
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
        int32_t* qw = weights + (size_t)i * ksize;
        int qsum = 0;
        const double norm = fabs(sum) > 1e-12 ? 16384.0 / sum : 0.0;
        for (int k = 0; k < ksize; ++k) {
            const int v = (int)round(raw[k] * norm);
            qw[k] = v;
            qsum += v;
        }
        if (qsum != 16384) {
            int max_k = 0, max_abs = 0;
            for (int k = 0; k < ksize; ++k)
                if (abs(qw[k]) > max_abs) { max_abs = abs(qw[k]); max_k = k; }
            qw[max_k] += 16384 - qsum;
        }
    }
    free(raw);
    return ksize;
}

static void resize_separable_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C, int filt, int antialias) {
    const int kx = ko_resize_contribs(sw, dw, filt, antialias, NULL, NULL, 0);
    const int ky = ko_resize_contribs(sh, dh, filt, antialias, NULL, NULL, 0);
    int32_t *xofs = (int32_t*)malloc(sizeof(int32_t) * (size_t)dw), *xw = (int32_t*)malloc(sizeof(int32_t) * (size_t)dw * kx);
    int32_t *yofs = (int32_t*)malloc(sizeof(int32_t) * (size_t)dh), *yw = (int32_t*)malloc(sizeof(int32_t) * (size_t)dh * ky);
    ko_resize_contribs(sw, dw, filt, antialias, xofs, xw, kx);
    ko_resize_contribs(sh, dh, filt, antialias, yofs, yw, ky);
    const size_t hrow = (size_t)dw * C;
    int16_t* hbuf = (int16_t*)malloc(sizeof(int16_t) * hrow * sh);
#pragma omp parallel for schedule(static)
    for (int sy = 0; sy < sh; ++sy) /* horizontal_row_scalar, kernels.rs:403-425 */
        for (int x = 0; x < dw; ++x)
            for (int ch = 0; ch < C; ++ch) {
                int32_t acc = 0;
                for (int t = 0; t < kx; ++t) {
                    const int sx = (int)llclamp((long long)xofs[x] + t, 0, sw - 1); /* build_xsrc_lut */
                    acc += (int32_t)src[((size_t)sy * sw + sx) * C + ch] * (int32_t)(int16_t)xw[(size_t)x * kx + t];
                }
                int32_t v = (acc + 8192) >> 14;
                hbuf[(size_t)sy * hrow + (size_t)x * C + ch] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
            }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) /* vertical_row_scalar, kernels.rs:699-708 */
        for (size_t i = 0; i < hrow; ++i) {
            int32_t acc = 0;
            for (int k = 0; k < ky; ++k) {
                const int sy = (int)llclamp((long long)yofs[y] + k, 0, sh - 1);
                acc += (int32_t)hbuf[(size_t)sy * hrow + i] * (int32_t)(int16_t)yw[(size_t)y * ky + k];
            }
            int32_t v = (acc + 8192) >> 14;
            dst[(size_t)y * hrow + i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    free(hbuf); free(xofs); free(xw); free(yofs); free(yw);
}

/* resize_u8_path + resize_fast_u8_aa, mod.rs:283-400.  mode: 0 nearest, 1 bilinear, 2 bicubic,
 * 3 lanczos.  Returns the path taken (1 pyrdown, 2 pyrup, 3 nearest, 4 bilinear, 5 separable) or
 * 0 for the reference's typed errors (unsupported channel count, bilinear source < 2x2). */
int ko_resize_fast_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C, int mode, int antialias) {
    if (mode == 1 && C == 3 && sw == dw * 2 && sh == dh * 2 && sw >= 2 && sh >= 2) { pyrdown_2x_rgb(src, dst, sw, sh); return 1; }
    if (mode == 1 && C == 3 && dw == sw * 2 && dh == sh * 2 && sw >= 2 && sh >= 2) { pyrup_2x_rgb(src, dst, sw, sh); return 2; }
    if (mode == 0) { resize_nearest_u8(src, sw, sh, dst, dw, dh, C); return 3; }
    if (!(C == 1 || C == 3 || C == 4)) return 0;
    if (mode == 1) {
        if (sw < 2 || sh < 2) return 0;
        resize_bilinear_u8(src, sw, sh, dst, dw, dh, C);
        return 4;
    }
    resize_separable_u8(src, sw, sh, dst, dw, dh, C, mode == 2 ? 0 : 1, antialias);
    return 5;
}

/* ---- OpenCV-compatible resize (opencv_compat.rs) --------------------------------------------------- */
typedef struct { int ofs; int border; float w0, w1; int i0, i1; } lin_tap;
static inline lin_tap linear_tap(int dx, double scale, int src_len) { /* linear_axis, :22-65 */
    float fx = (float)(((double)dx + 0.5) * scale - 0.5);
    long long sx = (long long)floorf(fx);
    fx -= (float)sx;
    lin_tap t;
    t.border = 0;
    if (sx < 0) { sx = 0; fx = 0.0f; }
    if (sx >= (long long)src_len - 1) { sx = (long long)src_len - 1; fx = 0.0f; t.border = 1; }
    t.ofs = (int)sx;
    t.w0 = 1.0f - fx;
    t.w1 = fx;
    t.i0 = (int)rintf((1.0f - fx) * 2048.0f); /* round_ties_even */
    t.i1 = (int)rintf(fx * 2048.0f);
    return t;
}
static inline int cv_nearest_index(int i, double iscale, int src_len) { /* nearest_axis, :67-74 */
    long long v = (long long)floor((double)i * iscale);
    return (int)(v < src_len - 1 ? v : src_len - 1);
}

/* mode: 0 nearest, 1 linear; elem = bytes per channel sample (1 or 4).  Returns 0 for other modes. */
static int cv_resize_nearest(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C, int elem) {
    const double isx = 1.0 / ((double)dw / (double)sw), isy = 1.0 / ((double)dh / (double)sh);
    const size_t px = (size_t)C * elem;
    for (int y = 0; y < dh; ++y) {
        const uint8_t* row = src + (size_t)cv_nearest_index(y, isy, sh) * sw * px;
        for (int x = 0; x < dw; ++x) memcpy(dst + ((size_t)y * dw + x) * px, row + (size_t)cv_nearest_index(x, isx, sw) * px, px);
    }
    return 1;
}

int ko_resize_opencv_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C, int mode) {
    if (mode == 0) return cv_resize_nearest(src, sw, sh, dst, dw, dh, C, 1);
    if (mode != 1) return 0;
    const double scx = 1.0 / ((double)dw / (double)sw), scy = 1.0 / ((double)dh / (double)sh);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        const lin_tap ty = linear_tap(y, scy, sh);
        const int sy1 = ty.ofs + 1 < sh - 1 ? ty.ofs + 1 : sh - 1;
        const uint8_t *r0 = src + (size_t)ty.ofs * sw * C, *r1 = src + (size_t)sy1 * sw * C;
        for (int x = 0; x < dw; ++x) {
            const lin_tap tx = linear_tap(x, scx, sw);
            for (int k = 0; k < C; ++k) { /* resize_linear_u8, :139-195 */
                const int sx = tx.ofs * C + k;
                const int32_t s0 = tx.border ? ((int32_t)r0[sx] << 11) : (int32_t)r0[sx] * tx.i0 + (int32_t)r0[sx + C] * tx.i1;
                const int32_t s1 = tx.border ? ((int32_t)r1[sx] << 11) : (int32_t)r1[sx] * tx.i0 + (int32_t)r1[sx + C] * tx.i1;
                dst[((size_t)y * dw + x) * C + k] = (uint8_t)((((ty.i0 * (s0 >> 4)) >> 16) + ((ty.i1 * (s1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    return 1;
}

int ko_resize_opencv_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int C, int mode) {
    if (mode == 0) return cv_resize_nearest((const uint8_t*)src, sw, sh, (uint8_t*)dst, dw, dh, C, 4);
    if (mode != 1) return 0;
    const double scx = 1.0 / ((double)dw / (double)sw), scy = 1.0 / ((double)dh / (double)sh);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        const lin_tap ty = linear_tap(y, scy, sh);
        const int sy1 = ty.ofs + 1 < sh - 1 ? ty.ofs + 1 : sh - 1;
        const float *r0 = src + (size_t)ty.ofs * sw * C, *r1 = src + (size_t)sy1 * sw * C;
        for (int x = 0; x < dw; ++x) {
            const lin_tap tx = linear_tap(x, scx, sw);
            for (int k = 0; k < C; ++k) { /* resize_linear_f32, :197-250 */
                const int sx = tx.ofs * C + k;
                const float s0 = tx.border ? r0[sx] : r0[sx] * tx.w0 + r0[sx + C] * tx.w1;
                const float s1 = tx.border ? r1[sx] : r1[sx] * tx.w0 + r1[sx + C] * tx.w1;
                dst[((size_t)y * dw + x) * C + k] = s0 * ty.w0 + s1 * ty.w1;
            }
        }
    }
    return 1;
}

/* ---- fused resize + normalize + HWC->CHW for RGB8 (P/resize/fused.rs) --------------------------------
 * out = sample * scale[c] + bias[c] with NormalizeParams::from_mean_std (fused.rs:29-38):
 * scale = 1/(std*255), bias = -mean/std.  The oracle is the SCALAR expression of each path (the
 * reference's NEON/AVX2 rows use FMA and differ by <= 1 ulp, cf. SURVEY.md 8c caveat). */
void ko_normalize_params(const float mean[3], const float std[3], float scale[3], float bias[3]) {
    for (int c = 0; c < 3; ++c) {
        scale[c] = 1.0f / (std[c] * 255.0f);
        bias[c] = -mean[c] / std[c];
    }
}

/* mode: 0 nearest (fused.rs:885-936), 1 bilinear incl. the exact-2x box dispatch (:147-232, :528-558,
 * :273-320), 2 bicubic / 3 lanczos separable (:938-1040).  Returns the path: 1 box2x, 2 bilinear,
 * 3 nearest, 4 separable. */
int ko_resize_normalize_to_chw(const uint8_t* src, int sw, int sh, float* dst, int dw, int dh, const float scale[3],
                               const float bias[3], int mode, int antialias) {
    const size_t plane = (size_t)dw * dh;
    if (mode == 1 && sw == 2 * dw && sh == 2 * dh) { /* fused_row_scalar */
        const float s[3] = {scale[0] * 0.25f, scale[1] * 0.25f, scale[2] * 0.25f};
#pragma omp parallel for schedule(static)
        for (int y = 0; y < dh; ++y) {
            const uint8_t *r0 = src + (size_t)(2 * y) * sw * 3, *r1 = r0 + (size_t)sw * 3;
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < 3; ++c) {
                    const uint32_t sum = (uint32_t)r0[6 * x + c] + r0[6 * x + 3 + c] + r1[6 * x + c] + r1[6 * x + 3 + c];
                    dst[c * plane + (size_t)y * dw + x] = (float)sum * s[c] + bias[c];
                }
        }
        return 1;
    }
    if (mode == 1) {
        const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < dh; ++y) {
            float fy = ((float)y + 0.5f) * scale_y - 0.5f;
            fy = fy > 0.0f ? fy : 0.0f; /* .max(0.0) */
            int y0 = (int)fy; if (y0 > sh - 1) y0 = sh - 1;
            const int y1 = y0 + 1 < sh - 1 ? y0 + 1 : sh - 1;
            const float wy = fy - (float)y0;
            const uint8_t *row0 = src + (size_t)y0 * sw * 3, *row1 = src + (size_t)y1 * sw * 3;
            for (int x = 0; x < dw; ++x) {
                float fx = ((float)x + 0.5f) * scale_x - 0.5f;
                fx = fx > 0.0f ? fx : 0.0f;
                int x0 = (int)fx; if (x0 > sw - 1) x0 = sw - 1;
                const int x1 = x0 + 1 < sw - 1 ? x0 + 1 : sw - 1;
                const float w = fx - (float)x0;
                for (int c = 0; c < 3; ++c) { /* blerp, :285-289 */
                    const float a = row0[x0 * 3 + c], b = row0[x1 * 3 + c], cc = row1[x0 * 3 + c], d = row1[x1 * 3 + c];
                    const float top = a + w * (b - a), bot = cc + w * (d - cc);
                    const float v = top + wy * (bot - top);
                    dst[c * plane + (size_t)y * dw + x] = v * scale[c] + bias[c];
                }
            }
        }
        return 2;
    }
    if (mode == 0) {
        const double sx = (double)sw / (double)dw, sy = (double)sh / (double)dh;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < dh; ++y) {
            const uint8_t* srow = src + (size_t)nearest_index(y, sy, sh) * sw * 3;
            for (int x = 0; x < dw; ++x) {
                const int o = nearest_index(x, sx, sw) * 3;
                for (int c = 0; c < 3; ++c) dst[c * plane + (size_t)y * dw + x] = (float)srow[o + c] * scale[c] + bias[c];
            }
        }
        return 3;
    }
    /* separable: Q14 horizontal pass to i16 as in resize_separable_u8, vertical pass accumulates i32 and
     * emits acc * (scale / 2^14) + bias without rounding back to u8 */
    const int filt = mode == 2 ? 0 : 1;
    const int kx = ko_resize_contribs(sw, dw, filt, antialias, NULL, NULL, 0);
    const int ky = ko_resize_contribs(sh, dh, filt, antialias, NULL, NULL, 0);
    int32_t *xofs = (int32_t*)malloc(sizeof(int32_t) * (size_t)dw), *xw = (int32_t*)malloc(sizeof(int32_t) * (size_t)dw * kx);
    int32_t *yofs = (int32_t*)malloc(sizeof(int32_t) * (size_t)dh), *yw = (int32_t*)malloc(sizeof(int32_t) * (size_t)dh * ky);
    ko_resize_contribs(sw, dw, filt, antialias, xofs, xw, kx);
    ko_resize_contribs(sh, dh, filt, antialias, yofs, yw, ky);
    const size_t hrow = (size_t)dw * 3;
    int16_t* hbuf = (int16_t*)malloc(sizeof(int16_t) * hrow * sh);
#pragma omp parallel for schedule(static)
    for (int sy = 0; sy < sh; ++sy)
        for (int x = 0; x < dw; ++x)
            for (int ch = 0; ch < 3; ++ch) {
                int32_t acc = 0;
                for (int t = 0; t < kx; ++t) {
                    const int sx = (int)llclamp((long long)xofs[x] + t, 0, sw - 1);
                    acc += (int32_t)src[((size_t)sy * sw + sx) * 3 + ch] * (int32_t)(int16_t)xw[(size_t)x * kx + t];
                }
                int32_t v = (acc + 8192) >> 14;
                hbuf[(size_t)sy * hrow + (size_t)x * 3 + ch] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
            }
    const float inv_q = 1.0f / 16384.0f;
    const float s[3] = {scale[0] * inv_q, scale[1] * inv_q, scale[2] * inv_q};
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x)
            for (int c = 0; c < 3; ++c) {
                int32_t acc = 0;
                for (int k = 0; k < ky; ++k) {
                    const int sy = (int)llclamp((long long)yofs[y] + k, 0, sh - 1);
                    acc += (int32_t)hbuf[(size_t)sy * hrow + (size_t)x * 3 + c] * yw[(size_t)y * ky + k]; /* i32 weight here */
                }
                dst[c * plane + (size_t)y * dw + x] = (float)acc * s[c] + bias[c];
            }
    free(hbuf); free(xofs); free(xw); free(yofs); free(yw);
    return 4;
}

/* ---- fused per-pixel pipelines (P/cuda/fusion.rs) ---------------------------------------------------
 * source ReadU8RgbBilinear (:545-585) -> maps Normalize (kind 16, :610-616) / RgbToGray (kind 17,
 * :636-640) -> sink WriteChwF32 (kind 32) / WriteC1F32 (kind 33).  rdw/rdh = the source stage's own
 * dst size (the half-pixel coefficients are built from it, :538-543).  f32, uncontracted. */
void ko_fused_pipeline(const uint8_t* src, int sw, int sh, int rdw, int rdh, int dw, int dh, const int* map_kinds,
                       const float* map_params /* [nmaps][6] */, int nmaps, int sink, float* dst) {
    const float axv = (float)sw / (float)rdw, ayv = (float)sh / (float)rdh;
    const float ax = axv, bx = 0.5f * axv - 0.5f, ay = ayv, by = 0.5f * ayv - 0.5f;
    const size_t plane = (size_t)dw * dh;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            float sxf = ax * (float)x + bx; sxf = sxf > 0.0f ? sxf : 0.0f;
            float syf = ay * (float)y + by; syf = syf > 0.0f ? syf : 0.0f;
            unsigned sx0 = (unsigned)sxf; if (sx0 > (unsigned)sw - 1u) sx0 = (unsigned)sw - 1u;
            unsigned sy0 = (unsigned)syf; if (sy0 > (unsigned)sh - 1u) sy0 = (unsigned)sh - 1u;
            const unsigned sx1 = sx0 + 1u < (unsigned)sw - 1u ? sx0 + 1u : (unsigned)sw - 1u;
            const unsigned sy1 = sy0 + 1u < (unsigned)sh - 1u ? sy0 + 1u : (unsigned)sh - 1u;
            const float wx = sxf - (float)sx0, wy = syf - (float)sy0;
            const uint8_t *r0 = src + (size_t)sy0 * sw * 3, *r1 = src + (size_t)sy1 * sw * 3;
            const float w00 = (1.0f - wy) * (1.0f - wx), w01 = (1.0f - wy) * wx, w10 = wy * (1.0f - wx), w11 = wy * wx;
            float v[3];
            for (int c = 0; c < 3; ++c)
                v[c] = w00 * (float)r0[sx0 * 3 + c] + w01 * (float)r0[sx1 * 3 + c] + w10 * (float)r1[sx0 * 3 + c] + w11 * (float)r1[sx1 * 3 + c];
            for (int i = 0; i < nmaps; ++i) {
                const float* f = map_params + (size_t)i * 6;
                if (map_kinds[i] == 16) {
                    v[0] = v[0] * f[0] + f[3]; v[1] = v[1] * f[1] + f[4]; v[2] = v[2] * f[2] + f[5];
                } else {
                    const float g = 0.299f * v[0] + 0.587f * v[1] + 0.114f * v[2];
                    v[0] = g; v[1] = g; v[2] = g;
                }
            }
            const size_t di = (size_t)y * dw + x;
            dst[di] = v[0];
            if (sink == 32) { dst[di + plane] = v[1]; dst[di + 2 * plane] = v[2]; }
        }
}
