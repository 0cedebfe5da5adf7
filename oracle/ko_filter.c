/*
 * CPU oracle — separable f32 filters.  TEST INFRASTRUCTURE (see ko_oracle.h).
 *
 * P/filter/kernels.rs:10-100 (tap builders), P/filter/separable_filter.rs:87-164 (H pass into an f32
 * temp, then V pass; `acc += v * k` in ascending tap order; taps that fall outside the image are
 * skipped — zero border, no renormalisation), P/filter/ops.rs:39-221 (box / gaussian / sobel /
 * scharr front-ends, SciPy kernel-size / sigma conventions).
 * Pinned on the exact 25-float vectors of P/filter/ops.rs:2185-2262 by tests/test_oracle_filter.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ko_oracle.h"

/* kernels.rs:10-13 */
void ko_box_blur_kernel_1d(int n, float* out) {
    for (int i = 0; i < n; ++i) out[i] = 1.0f / (float)n;
}

/* kernels.rs:25-43 (f32 exp = libm expf; sum in index order; divide) */
void ko_gaussian_kernel_1d(int n, float sigma, float* out) {
    float mean = (float)(n - 1) / 2.0f, sigma_sq = sigma * sigma;
    for (int i = 0; i < n; ++i) {
        float x = (float)i - mean;
        out[i] = expf(-(x * x) / (2.0f * sigma_sq));
    }
    float norm = 0.0f;
    for (int i = 0; i < n; ++i) norm += out[i];
    for (int i = 0; i < n; ++i) out[i] /= norm;
}

/* kernels.rs:55-100; returns 0 on unsupported size.  kind: 0 sobel (3|5), 1 scharr (3) */
int ko_gradient_kernels_1d(int kind, int n, float* kx, float* ky) {
    if (kind == 0 && n == 3) { const float a[3] = {-1, 0, 1}, b[3] = {1, 2, 1}; memcpy(kx, a, 12); memcpy(ky, b, 12); return 1; }
    if (kind == 0 && n == 5) { const float a[5] = {-1, -2, 0, 2, 1}, b[5] = {1, 4, 6, 4, 1}; memcpy(kx, a, 20); memcpy(ky, b, 20); return 1; }
    if (kind == 1 && n == 3) { const float a[3] = {-1, 0, 1}, b[3] = {3, 10, 3}; memcpy(kx, a, 12); memcpy(ky, b, 12); return 1; }
    return 0;
}

/* gaussian_blur parameter resolution, ops.rs:122-155.  In/out: k[2], s[2]; returns 0 if invalid. */
int ko_gaussian_resolve(int k[2], float s[2]) {
    int kx = k[0], ky = k[1];
    float sx = s[0], sy = s[1];
    if (sy <= 0.0f) sy = sx;
    if (kx == 0 && sx > 0.0f) kx = (int)(2.0f * roundf(4.0f * sx) + 1.0f) | 1;
    if (ky == 0 && sy > 0.0f) ky = (int)(2.0f * roundf(4.0f * sy) + 1.0f) | 1;
    if (!(kx > 0 && kx % 2 == 1 && ky > 0 && ky % 2 == 1)) return 0;
    sx = sx > 0.0f ? sx : 0.0f;
    sy = sy > 0.0f ? sy : 0.0f;
    if (sx == 0.0f) sx = ((float)kx - 1.0f) / 8.0f;
    if (sy == 0.0f) sy = ((float)ky - 1.0f) / 8.0f;
    k[0] = kx; k[1] = ky; s[0] = sx; s[1] = sy;
    return 1;
}

/* SeparableFilter::apply, separable_filter.rs:87-164.  threads: 0 = the reference's single thread
 * (parallel rows only when ko_set_threads > 1 was requested: the "beyond reference" baseline). */
void ko_separable_filter_f32(const float* src, float* dst, int cols, int rows, int C, const float* kx, int nx,
                             const float* ky, int ny) {
    const int hx = nx / 2, hy = ny / 2;
    float* temp = (float*)malloc((size_t)rows * cols * C * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            for (int ch = 0; ch < C; ++ch) {
                float acc = 0.0f;
                for (int i = 0; i < nx; ++i) {
                    int x = c + i - hx;
                    if (x >= 0 && x < cols) acc += src[((size_t)r * cols + x) * C + ch] * kx[i];
                }
                temp[((size_t)r * cols + c) * C + ch] = acc;
            }
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            for (int ch = 0; ch < C; ++ch) {
                float acc = 0.0f;
                for (int i = 0; i < ny; ++i) {
                    int y = r + i - hy;
                    if (y >= 0 && y < rows) acc += temp[((size_t)y * cols + c) * C + ch] * ky[i];
                }
                dst[((size_t)r * cols + c) * C + ch] = acc;
            }
    free(temp);
}

/* sobel / scharr, ops.rs:174-247: gx = sep(kx, ky), gy = sep(ky, kx), dst = sqrt(gx^2 + gy^2) */
void ko_gradient_magnitude_f32(const float* src, float* dst, int cols, int rows, int C, const float* kx, const float* ky, int n) {
    size_t len = (size_t)rows * cols * C;
    float* gx = (float*)malloc(len * sizeof(float));
    float* gy = (float*)malloc(len * sizeof(float));
    ko_separable_filter_f32(src, gx, cols, rows, C, kx, n, ky, n);
    ko_separable_filter_f32(src, gy, cols, rows, C, ky, n, kx, n);
    for (size_t i = 0; i < len; ++i) dst[i] = sqrtf(gx[i] * gx[i] + gy[i] * gy[i]);
    free(gx);
    free(gy);
}
