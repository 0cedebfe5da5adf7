/*
 * CPU oracle — u8 fixed-point twins: Q8 separable blur, 3x3 binomial, Q10 bilinear remap and
 * affine warp.  TEST INFRASTRUCTURE (see ko_oracle.h).
 *
 * P/filter/ops.rs:595-760 (parameter resolution, quantize_kernel_256, path selection, the per-pass
 * `(acc + 128) >> 8` rounding with replicate borders — stated most directly by the device twins
 * P/cuda/filter.rs:116-215), P/warp/common.rs:16-165 (Q10 bilinear sample),
 * P/interpolation/remap.rs:157-300 (remap_u8), P/warp/affine.rs:373-445 + P/warp/span.rs:36-85 +
 * P/warp/kernels.rs:386-415 (warp_affine_u8: per-row valid span, Q16 stepping with wrapping adds).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ko_oracle.h"

static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* quantize_kernel_256, ops.rs:748-760: `(k*256+0.5) as u8` (saturating), centre absorbs the error */
void ko_quantize_kernel_256(const float* k, int n, uint8_t* out) {
    int sum = 0;
    for (int i = 0; i < n; ++i) {
        float v = k[i] * 256.0f + 0.5f;
        int q = v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (int)v); /* Rust `as u8` saturates */
        out[i] = (uint8_t)q;
        sum += q;
    }
    if (sum != 256) out[n / 2] = (uint8_t)iclamp((int)out[n / 2] + (256 - sum), 0, 255);
}

/* one Q8 pass, replicate borders: P/cuda/filter.rs:116-165 == hpass_u8_row / striped V pass */
static void pass_q8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C, const uint8_t* k, int n, int horizontal) {
    const int half = n / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            for (int ch = 0; ch < C; ++ch) {
                unsigned acc = 0;
                for (int t = 0; t < n; ++t) {
                    int xx = horizontal ? iclamp(x + t - half, 0, cols - 1) : x;
                    int yy = horizontal ? y : iclamp(y + t - half, 0, rows - 1);
                    acc += (unsigned)src[((size_t)yy * cols + xx) * C + ch] * k[t];
                }
                dst[((size_t)y * cols + x) * C + ch] = (uint8_t)((acc + 128u) >> 8);
            }
}

void ko_separable_blur_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C, const uint8_t* kx, int nx,
                          const uint8_t* ky, int ny) {
    uint8_t* tmp = (uint8_t*)malloc((size_t)cols * rows * C);
    pass_q8(src, tmp, cols, rows, C, kx, nx, 1);
    pass_q8(tmp, dst, cols, rows, C, ky, ny, 0);
    free(tmp);
}

/* [1,2,1]/4 as nested rounding halving adds, H then V (P/cuda/filter.rs:170-215) */
static inline unsigned rhadd(unsigned a, unsigned b) { return (a + b + 1u) >> 1; }
void ko_binomial3_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C) {
    uint8_t* tmp = (uint8_t*)malloc((size_t)cols * rows * C);
    for (int pass = 0; pass < 2; ++pass) {
        const uint8_t* s = pass ? tmp : src;
        uint8_t* d = pass ? dst : tmp;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) {
                int xm = pass ? x : iclamp(x - 1, 0, cols - 1), xp = pass ? x : iclamp(x + 1, 0, cols - 1);
                int ym = pass ? iclamp(y - 1, 0, rows - 1) : y, yp = pass ? iclamp(y + 1, 0, rows - 1) : y;
                for (int ch = 0; ch < C; ++ch) {
                    unsigned a = s[((size_t)ym * cols + xm) * C + ch], b = s[((size_t)y * cols + x) * C + ch],
                             e = s[((size_t)yp * cols + xp) * C + ch];
                    d[((size_t)y * cols + x) * C + ch] = (uint8_t)rhadd(rhadd(a, b), rhadd(b, e));
                }
            }
    }
    free(tmp);
}

/* gaussian_blur_u8, ops.rs:639-745: returns 0 on invalid parameters; path 1 = binomial, 2 = general */
int ko_gaussian_blur_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C, int kx, int ky, float sx, float sy) {
    int k[2] = {kx, ky};
    float s[2] = {sx, sy};
    if (!ko_gaussian_resolve(k, s)) return 0;
    if (k[0] == 3 && k[1] == 3 && s[0] >= 0.6f && s[0] <= 1.2f && s[1] >= 0.6f && s[1] <= 1.2f) { /* blur_u8_path, :21-27 */
        ko_binomial3_u8(src, dst, cols, rows, C);
        return 1;
    }
    float fx[64], fy[64];
    uint8_t qx[64], qy[64];
    if (k[0] > 63 || k[1] > 63) return 0;
    ko_gaussian_kernel_1d(k[0], s[0], fx);
    ko_gaussian_kernel_1d(k[1], s[1], fy);
    ko_quantize_kernel_256(fx, k[0], qx);
    ko_quantize_kernel_256(fy, k[1], qy);
    ko_separable_blur_u8(src, dst, cols, rows, C, qx, k[0], qy, k[1]);
    return 2;
}

/* box_blur_u8, ops.rs:59-100: odd positive sizes only */
int ko_box_blur_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C, int kx, int ky) {
    if (kx <= 0 || ky <= 0 || kx % 2 == 0 || ky % 2 == 0 || kx > 63 || ky > 63) return 0;
    float fx[64], fy[64];
    uint8_t qx[64], qy[64];
    ko_box_blur_kernel_1d(kx, fx);
    ko_box_blur_kernel_1d(ky, fy);
    ko_quantize_kernel_256(fx, kx, qx);
    ko_quantize_kernel_256(fy, ky, qy);
    ko_separable_blur_u8(src, dst, cols, rows, C, qx, kx, qy, ky);
    return 1;
}

/* bilinear_sample_u8_valid, common.rs:79-165 (scalar tail): xi, yi in range, fx/fy Q10 */
static inline void sample_q10(const uint8_t* src, int sw, int sh, int C, int xi, int yi, unsigned fx, unsigned fy, uint8_t* o) {
    const unsigned fx1 = 1024u - fx, fy1 = 1024u - fy;
    const int xi1 = xi + 1 < sw ? xi + 1 : xi, yi1 = yi + 1 < sh ? yi + 1 : yi;
    const size_t r0 = (size_t)yi * sw * C, r1 = (size_t)yi1 * sw * C, x0 = (size_t)xi * C, x1 = (size_t)xi1 * C;
    for (int ch = 0; ch < C; ++ch) {
        unsigned p00 = src[r0 + x0 + ch], p01 = src[r0 + x1 + ch], p10 = src[r1 + x0 + ch], p11 = src[r1 + x1 + ch];
        unsigned top = p00 * fx1 + p01 * fx, bot = p10 * fx1 + p11 * fx;
        o[ch] = (uint8_t)((top * fy1 + bot * fy + (1u << 19)) >> 20);
    }
}

/* remap_u8, remap.rs:157-300; mode 0 nearest, 1 bilinear */
void ko_remap_u8(const uint8_t* src, int sw, int sh, const float* map_x, const float* map_y, uint8_t* dst, int dw, int dh, int C, int mode) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            const size_t i = (size_t)y * dw + x;
            const float xf = map_x[i], yf = map_y[i];
            uint8_t* o = dst + i * C;
            if (mode == 1) {
                if (!isfinite(xf) || !isfinite(yf)) { memset(o, 0, C); continue; }
                const float fxf = floorf(xf), fyf = floorf(yf);
                /* `as i32` saturates; only the range test matters */
                const int xi = fxf >= 2147483648.0f ? 2147483647 : (fxf <= -2147483648.0f ? -2147483647 - 1 : (int)fxf);
                const int yi = fyf >= 2147483648.0f ? 2147483647 : (fyf <= -2147483648.0f ? -2147483647 - 1 : (int)fyf);
                if (xi < 0 || xi >= sw || yi < 0 || yi >= sh) { memset(o, 0, C); continue; }
                sample_q10(src, sw, sh, C, xi, yi, (unsigned)((xf - (float)xi) * 1024.0f), (unsigned)((yf - (float)yi) * 1024.0f), o);
            } else {
                if (!(xf >= 0.0f && xf < (float)sw && yf >= 0.0f && yf < (float)sh)) { memset(o, 0, C); continue; }
                const int xi = iclamp((int)roundf(xf), 0, sw - 1), yi = iclamp((int)roundf(yf), 0, sh - 1);
                memcpy(o, src + ((size_t)yi * sw + xi) * C, C);
            }
        }
}

/* Rust `f32 as i64`: saturating, NaN -> 0 */
static inline long long f2ll(float v) {
    if (v != v) return 0;
    if (v >= 9223372036854775808.0f) return 9223372036854775807LL;
    if (v <= -9223372036854775808.0f) return -9223372036854775807LL - 1;
    return (long long)v;
}
static inline long long inc_sat(long long v) { return v == 9223372036854775807LL ? v : v + 1; }

/* Rust `f32 as i32`: saturating, NaN -> 0 */
static inline int f2i(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return -2147483647 - 1;
    return (int)v;
}

/* constrain_span, span.rs:36-59 */
static void constrain_span(float a, float b, int ge, float eps, long long* lo, long long* hi) {
    if (fabsf(a) < eps || a == 0.0f) {
        int feasible = ge ? (b >= 0.0f) : (b < 0.0f);
        if (!feasible) *hi = *lo;
        return;
    }
    float k = -b / a;
    if (ge && a > 0.0f) { long long v = f2ll(ceilf(k)); if (v > *lo) *lo = v; }
    else if (ge) { long long v = inc_sat(f2ll(floorf(k))); if (v < *hi) *hi = v; }
    else if (a > 0.0f) { long long v = f2ll(ceilf(k)); if (v < *hi) *hi = v; }
    else { long long v = inc_sat(f2ll(floorf(k))); if (v > *lo) *lo = v; }
}

/* warp_affine_u8, affine.rs:373-445 (m = forward 2x3) */
void ko_warp_affine_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C, const float m[6]) {
    float mi[6];
    ko_invert_affine_transform(m, mi);
    const float dsx = mi[0], dsy = mi[3];
    const int dsx_q = f2i(dsx * 65536.0f), dsy_q = f2i(dsy * 65536.0f);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        uint8_t* row = dst + (size_t)y * dw * C;
        const float yf = (float)y, sx0 = mi[1] * yf + mi[2], sy0 = mi[4] * yf + mi[5];
        long long lo = 0, hi = dw;
        int empty = 0;
        const float axes[2][3] = {{dsx, sx0, (float)sw}, {dsy, sy0, (float)sh}};
        for (int a = 0; a < 2 && !empty; ++a) { /* affine_valid_span, span.rs:61-85 */
            constrain_span(axes[a][0], axes[a][1], 1, 1e-12f, &lo, &hi);
            constrain_span(axes[a][0], axes[a][1] - axes[a][2], 0, 1e-12f, &lo, &hi);
            if (lo >= hi) empty = 1;
        }
        long long lo_c = lo < 0 ? 0 : (lo > dw ? dw : lo), hi_c = hi < 0 ? 0 : (hi > dw ? dw : hi);
        if (empty || lo_c >= hi_c) { lo_c = 0; hi_c = 0; }
        memset(row, 0, (size_t)lo_c * C);
        memset(row + (size_t)hi_c * C, 0, (size_t)(dw - hi_c) * C);
        if (lo_c >= hi_c) continue;
        int sx_q = f2i((sx0 + dsx * (float)lo_c) * 65536.0f), sy_q = f2i((sy0 + dsy * (float)lo_c) * 65536.0f);
        for (long long x = lo_c; x < hi_c; ++x) { /* process_affine_span_scalar, kernels.rs:386-415 */
            /* the span guarantees in-range indices in exact arithmetic; clamp so Q16 drift can never
             * read out of bounds (a no-op whenever the reference itself is memory-safe) */
            const int xi = iclamp(sx_q >> 16, 0, sw - 1), yi = iclamp(sy_q >> 16, 0, sh - 1);
            sample_q10(src, sw, sh, C, xi, yi, ((unsigned)(sx_q & 0xFFFF)) >> 6, ((unsigned)(sy_q & 0xFFFF)) >> 6, row + (size_t)x * C);
            sx_q = (int)((unsigned)sx_q + (unsigned)dsx_q);
            sy_q = (int)((unsigned)sy_q + (unsigned)dsy_q);
        }
    }
}

/* bilinear_sample_u8, common.rs:16-70: bounds-checked, zeros when non-finite or outside */
static inline void sample_q10_checked(const uint8_t* src, int sw, int sh, int C, float xf, float yf, uint8_t* o) {
    if (!isfinite(xf) || !isfinite(yf)) { memset(o, 0, C); return; }
    const float fxf = floorf(xf), fyf = floorf(yf);
    const int xi = fxf >= 2147483648.0f ? 2147483647 : (fxf <= -2147483648.0f ? -2147483647 - 1 : (int)fxf);
    const int yi = fyf >= 2147483648.0f ? 2147483647 : (fyf <= -2147483648.0f ? -2147483647 - 1 : (int)fyf);
    if (xi < 0 || xi >= sw || yi < 0 || yi >= sh) { memset(o, 0, C); return; }
    sample_q10(src, sw, sh, C, xi, yi, (unsigned)((xf - (float)xi) * 1024.0f), (unsigned)((yf - (float)yi) * 1024.0f), o);
}

/* warp_perspective_u8, perspective.rs:179-322 (m = forward 3x3).  Returns 0 if m is singular.
 * Rows whose denominator keeps one sign get the analytic column span (zeros outside); inside the
 * span the reference's edge pixels use the bounds-checked sampler and the interior the unchecked
 * one — the two agree whenever the interior coordinate is in bounds, i.e. whenever the reference
 * itself is memory-safe, so the checked sampler is used throughout. */
int ko_warp_perspective_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C, const float m[9]) {
    float inv[9];
    if (!ko_invert_homography(m, inv)) return 0;
    const float swf = (float)sw, shf = (float)sh;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        uint8_t* row = dst + (size_t)y * dw * C;
        const float yf = (float)y;
        float nx0 = inv[1] * yf + inv[2], ny0 = inv[4] * yf + inv[5], nd0 = inv[7] * yf + inv[8];
        float dnx = inv[0], dny = inv[3], dnd = inv[6];
        const float nd_end = nd0 + dnd * ((float)dw - 1.0f);
        const int pos = nd0 > 1e-6f && nd_end > 1e-6f, neg = nd0 < -1e-6f && nd_end < -1e-6f;
        long long lo = 0, hi = dw;
        if (pos || neg) {
            if (neg) { nx0 = -nx0; ny0 = -ny0; nd0 = -nd0; dnx = -dnx; dny = -dny; dnd = -dnd; }
            constrain_span(dnx, nx0, 1, 0.0f, &lo, &hi);
            constrain_span(dnx - swf * dnd, nx0 - swf * nd0, 0, 0.0f, &lo, &hi);
            constrain_span(dny, ny0, 1, 0.0f, &lo, &hi);
            constrain_span(dny - shf * dnd, ny0 - shf * nd0, 0, 0.0f, &lo, &hi);
            lo = lo < 0 ? 0 : (lo > dw ? dw : lo);
            hi = hi < 0 ? 0 : (hi > dw ? dw : hi);
            if (lo >= hi) { lo = 0; hi = 0; }
        }
        memset(row, 0, (size_t)lo * C);
        memset(row + (size_t)hi * C, 0, (size_t)(dw - hi) * C);
        for (long long x = lo; x < hi; ++x) { /* perspective_coord_at, kernels.rs:107-122 */
            const float xf_ = (float)x;
            const float nx = nx0 + dnx * xf_, ny = ny0 + dny * xf_, nd = nd0 + dnd * xf_;
            const float inv_nd = 1.0f / nd;
            sample_q10_checked(src, sw, sh, C, nx * inv_nd, ny * inv_nd, row + (size_t)x * C);
        }
    }
    return 1;
}
