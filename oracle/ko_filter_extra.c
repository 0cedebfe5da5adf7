/*
 * ko_filter_extra.c — TEST INFRASTRUCTURE (see ko_oracle.h): CPU restatement of the remaining operators of the
 * reference's filter module: the 3x3 spatial gradients, the repeated-box "fast" blur, the median blur and the
 * cv2-compatible bilateral filter.  Paths relative to /root/reference, P/ = crates/kornia-imgproc/src/.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ko_oracle.h"

/* ---- spatial gradients ------------------------------------------------------------------------------------------------
 * spatial_gradient_float (P/filter/ops.rs:287-353), its _parallel_row / _parallel variants (:355-509, same arithmetic)
 * and scharr_spatial_gradient_float (:511-590): a 3x3 cross-correlation with the normalised kernels of
 * P/filter/kernels.rs:107-140, replicate border written as row = min(r + dy, rows).max(1) - 1, and the nine
 * products added in (dy, dx) row-major order onto 0.0 — zero taps included (val * 0.0 is still added).            */
static const float k_sobel_x[3][3] = {{-0.125f, 0.0f, 0.125f}, {-0.25f, 0.0f, 0.25f}, {-0.125f, 0.0f, 0.125f}};
static const float k_sobel_y[3][3] = {{-0.125f, -0.25f, -0.125f}, {0.0f, 0.0f, 0.0f}, {0.125f, 0.25f, 0.125f}};
static const float k_scharr_x[3][3] = {{-0.09375f, 0.0f, 0.09375f}, {-0.3125f, 0.0f, 0.3125f}, {-0.09375f, 0.0f, 0.09375f}};
static const float k_scharr_y[3][3] = {{-0.09375f, -0.3125f, -0.09375f}, {0.0f, 0.0f, 0.0f}, {0.09375f, 0.3125f, 0.09375f}};

void ko_spatial_gradient_f32(const float* src, float* gx, float* gy, int cols, int rows, int C, int kind) {
    const float(*kx)[3] = kind == 1 ? k_scharr_x : k_sobel_x;
    const float(*ky)[3] = kind == 1 ? k_scharr_y : k_sobel_y;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            for (int ch = 0; ch < C; ++ch) {
                float sum_x = 0.0f, sum_y = 0.0f;
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx = 0; dx < 3; ++dx) {
                        int row = r + dy < rows ? r + dy : rows;
                        row = (row > 1 ? row : 1) - 1;
                        int col = c + dx < cols ? c + dx : cols;
                        col = (col > 1 ? col : 1) - 1;
                        const float val = src[((size_t)row * cols + col) * C + ch];
                        sum_x += val * kx[dy][dx];
                        sum_y += val * ky[dy][dx];
                    }
                gx[((size_t)r * cols + c) * C + ch] = sum_x;
                gy[((size_t)r * cols + c) * C + ch] = sum_y;
            }
}

/* ---- box_blur_fast ------------------------------------------------------------------------------------------------------
 * box_blur_fast_kernels_1d (P/filter/kernels.rs:151-170): Kovesi's box sizes for a target sigma, all in f32;
 * `ideal_m.round() as u8` is Rust's saturating cast of the half-away-from-zero rounding.                           */
void ko_box_blur_fast_kernels_1d(float sigma, int kernels, int* out) {
    const float n = (float)kernels;
    const float ideal_size = sqrtf(12.0f * sigma * sigma / n + 1.0f);
    float size_l = floorf(ideal_size);
    size_l -= fmodf(size_l, 2.0f) == 0.0f ? 1.0f : 0.0f;
    const float size_u = size_l + 2.0f;
    const float ideal_m = (12.0f * sigma * sigma - n * size_l * size_l - 4.0f * n * size_l - 3.0f * n) / (-4.0f * size_l - 4.0f);
    const float rm = roundf(ideal_m);
    const int m = rm != rm ? 0 : rm <= 0.0f ? 0 : rm >= 255.0f ? 255 : (int)rm;
    for (int i = 0; i < kernels; ++i) {
        const float s = i < m ? size_l : size_u;
        out[i] = s != s ? 0 : s <= 0.0f ? 0 : (int)s; /* `as usize` saturates at 0 */
    }
}

/* fast_horizontal_filter (P/filter/separable_filter.rs:202-257): running row sum with replicated ends, written
 * TRANSPOSED (dst is rows-wide, cols-tall).  Returns -1 where the reference would index out of bounds (half >= cols). */
int ko_fast_horizontal_filter(const float* src, float* dst, int cols, int rows, int C, int half) {
    if (half < 0 || half >= cols) return -1;
    const float first_scale = (float)(half + 1), norm = (float)(half * 2 + 1);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int ch = 0; ch < C; ++ch) {
            const float* row = src + (size_t)r * cols * C + ch;
            const float leftmost = row[0], rightmost = row[(size_t)(cols - 1) * C];
            float acc = 0.0f;
            for (int c = 0; c < cols; ++c) {
                if (c == 0) {
                    acc = row[0] * first_scale;
                    for (int p = 0; p < half; ++p) acc += row[(size_t)(p + 1) * C];
                } else {
                    acc -= c >= half + 1 ? row[(size_t)(c - half - 1) * C] : leftmost;
                    acc += c + half < cols ? row[(size_t)(c + half) * C] : rightmost;
                }
                dst[((size_t)c * rows + r) * C + ch] = acc / norm;
            }
        }
    return 0;
}

/* box_blur_fast (P/filter/ops.rs:252-285): three rounds of (horizontal box into a transposed image, horizontal box
 * of that back into dst); the sizes from box_blur_fast_kernels_1d are used as HALF widths, as the reference does. */
int ko_box_blur_fast_f32(const float* src, float* dst, int cols, int rows, int C, float sigma_x, float sigma_y) {
    int hx[3], hy[3];
    ko_box_blur_fast_kernels_1d(sigma_x, 3, hx);
    ko_box_blur_fast_kernels_1d(sigma_y, 3, hy);
    float* transposed = (float*)malloc(sizeof(float) * (size_t)cols * rows * C);
    if (!transposed) return -2;
    const float* in = src;
    int rc = 0;
    for (int i = 0; i < 3 && rc == 0; ++i) {
        rc = ko_fast_horizontal_filter(in, transposed, cols, rows, C, hx[i]);
        if (rc == 0) rc = ko_fast_horizontal_filter(transposed, dst, rows, cols, C, hy[i]);
        in = dst;
    }
    free(transposed);
    return rc;
}

/* ---- median_blur (P/filter/median.rs:174-250) -----------------------------------------------------------------------
 * The exact median of the replicate-bordered k x k window per channel, k in {3, 5}.  The reference selects it with
 * sorting networks and asserts they equal the naive median (median.rs:942); the order statistic is restated here by
 * counting, which is the definition.                                                                                */
int ko_median_blur_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C, int ksize) {
    if (ksize != 3 && ksize != 5) return -1;
    const int r = ksize / 2, rank = ksize * ksize / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < C; ++c) {
                int hist[256] = {0};
                for (int dy = -r; dy <= r; ++dy) {
                    const int sy = y + dy < 0 ? 0 : y + dy > rows - 1 ? rows - 1 : y + dy;
                    for (int dx = -r; dx <= r; ++dx) {
                        const int sx = x + dx < 0 ? 0 : x + dx > cols - 1 ? cols - 1 : x + dx;
                        ++hist[src[((size_t)sy * cols + sx) * C + c]];
                    }
                }
                int v = 0, below = hist[0];
                while (below <= rank) below += hist[++v];
                dst[((size_t)y * cols + x) * C + c] = (uint8_t)v;
            }
    return 0;
}

/* ---- bilateral_filter (P/filter/bilateral.rs) -----------------------------------------------------------------------
 * v_exp_f32 (:44-78): OpenCV's vectorised exp polynomial, every step a fused multiply-add as written there.       */
float ko_v_exp_f32(float x) {
    const float LO = -88.37626f, HI = 89.0f, LOG2EF = 1.44269504088896340736f;
    const float C1 = -6.9335938E-1f, C2 = 2.1219444E-4f;
    const float P0 = 1.9875692E-4f, P1 = 1.3981999E-3f, P2 = 8.333452E-3f, P3 = 4.1665796E-2f, P4 = 1.6666665E-1f, P5 = 5.0000002E-1f;
    x = x < LO ? LO : x > HI ? HI : x;
    const float t = fmaf(x, LOG2EF, 0.5f);
    const float mm = floorf(t);
    const int32_t mi = (int32_t)mm;
    const uint32_t sbits = (uint32_t)(mi + 0x7f) << 23;
    float scale;
    memcpy(&scale, &sbits, 4);
    x = fmaf(mm, C1, x);
    x = fmaf(mm, C2, x);
    const float xx = x * x;
    float y = fmaf(x, P0, P1);
    y = fmaf(y, x, P2);
    y = fmaf(y, x, P3);
    y = fmaf(y, x, P4);
    y = fmaf(y, x, P5);
    y = fmaf(y, xx, x);
    y = y + 1.0f;
    return y * scale;
}

static int bilateral_radius(int d, double sigma_space) {
    int radius;
    if (d <= 0) {
        const double r = nearbyint(sigma_space * 1.5); /* round_ties_even, then a saturating `as i32` */
        radius = r != r ? 0 : r >= 2147483647.0 ? 2147483647 : r <= -2147483648.0 ? (-2147483647 - 1) : (int)r;
    } else {
        radius = d / 2;
    }
    return radius < 1 ? 1 : radius;
}

/* build_tables (:110-170).  Returns the tap count; with capacity < count only `radius` is valid (size query). */
int ko_bilateral_tables(int d, double sigma_color, double sigma_space, int capacity, int* radius_out, int* tap_dy, int* tap_dx,
                        float* space_weight, float* color_weight, int* simd_order) {
    const float gauss_color_coeff = (float)(-0.5 / (sigma_color * sigma_color));
    const float gauss_space_coeff = (float)(-0.5 / (sigma_space * sigma_space));
    const int radius = bilateral_radius(d, sigma_space);
    if (radius_out) *radius_out = radius;
    int n = 0;
    for (int dy = -radius; dy <= radius; ++dy)
        for (int dx = -radius; dx <= radius; ++dx) {
            const double r = sqrt((double)(dy * dy + dx * dx));
            if (r > (double)radius) continue;
            if (n < capacity) {
                space_weight[n] = (float)exp((r * r) * (double)gauss_space_coeff);
                tap_dy[n] = dy;
                tap_dx[n] = dx;
            }
            ++n;
        }
    if (n > capacity) return n;
    /* cv2 fills entries 0 .. 256 - nlanes with its SIMD polynomial and the last nlanes = 4 with scalar expf */
    int i = 0;
    for (; i < 256 - 4; i += 4)
        for (int k = 0; k < 4; ++k) {
            const float fi = (float)(i + k);
            color_weight[i + k] = ko_v_exp_f32(fi * fi * gauss_color_coeff);
        }
    for (int j = i; j < 256; ++j) color_weight[j] = expf((float)(j * j) * gauss_color_coeff);
    static const int order13[13] = {0, 12, 1, 2, 3, 9, 10, 11, 4, 5, 6, 7, 8};
    for (int k = 0; k < n; ++k) simd_order[k] = n == 13 ? order13[k] : k;
    return n;
}

static long long reflect_101_iter(long long p, long long len) { /* P/clahe.rs:36-48 */
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

/* bilateral_filter + scalar_pixel (:172-300): degenerate sigmas copy through; otherwise per pixel
 * w = space[k] * color[|val - val0|], wsum += w, sum = fma(val, w, sum) in cv2's position-dependent tap order,
 * output round-half-even(sum / wsum).                                                                               */
int ko_bilateral_filter_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int d, double sigma_color, double sigma_space) {
    if (sigma_color <= 1e-6 || sigma_space <= 1e-6) {
        memcpy(dst, src, (size_t)cols * rows);
        return 0;
    }
    int radius = 0;
    int n = ko_bilateral_tables(d, sigma_color, sigma_space, 0, &radius, NULL, NULL, NULL, NULL, NULL);
    int* tdy = (int*)malloc(sizeof(int) * 3 * (size_t)n);
    float* sw = (float*)malloc(sizeof(float) * (size_t)n);
    float cw[256];
    if (!tdy || !sw) { free(tdy); free(sw); return -2; }
    int *tdx = tdy + n, *order = tdy + 2 * n;
    ko_bilateral_tables(d, sigma_color, sigma_space, n, &radius, tdy, tdx, sw, cw, order);
    const int simd_end = cols >= 16 ? ((cols - 16) / 16) * 16 + 16 : 0; /* simd_region_end, :99-106 */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const int val0 = src[(size_t)y * cols + x];
            float wsum = 0.0f, sum = 0.0f;
            const int in_simd = x < simd_end;
            for (int kk = 0; kk < n; ++kk) {
                const int k = in_simd ? order[kk] : kk;
                const long long sy = reflect_101_iter((long long)y + tdy[k], rows), sx = reflect_101_iter((long long)x + tdx[k], cols);
                const int val = src[(size_t)sy * cols + sx];
                const float wgt = sw[k] * cw[abs(val - val0)];
                wsum += wgt;
                sum = fmaf((float)val, wgt, sum);
            }
            const float q = nearbyintf(sum / wsum);
            dst[(size_t)y * cols + x] = q != q ? 0 : q <= 0.0f ? 0 : q >= 255.0f ? 255 : (uint8_t)q;
        }
    free(tdy);
    free(sw);
    return 0;
}
