/*
 * CPU oracle — f32 geometric ops: resize, warp_affine, warp_perspective, remap, undistort maps.
 * TEST INFRASTRUCTURE (see ko_oracle.h).
 *
 * Follows the reference's CPU arithmetic expression for expression.  Where the CPU path leaves
 * `dst` untouched (warp_perspective out-of-range pixels, P/warp/perspective.rs:147) or reads out
 * of bounds (remap has no guard, P/interpolation/remap.rs:92-105) the oracle writes 0 — the rule
 * of the reference's device kernels (P/cuda/warp_perspective.rs:80-84, P/cuda/remap.rs:80-85) and
 * equal to the CPU result on the zero-initialised `dst` every reference test and example uses.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ko_oracle.h"

/* ---- per-pixel samplers (P/interpolation/{bilinear,nearest,bicubic}.rs) ---------------------- */

/* bilinear_interpolation, bilinear.rs:16-66: trunc, edge taps replicate val00 */
static inline float sample_bilinear(const float* img, int rows, int cols, int C, float u, float v, int c) {
    int iu = (int)u, iv = (int)v; /* u.trunc() as usize; callers guarantee u,v >= 0 */
    float frac_u = u - truncf(u), frac_v = v - truncf(v);
    float val00 = img[((size_t)iv * cols + iu) * C + c];
    float val01 = (iu + 1 < cols) ? img[((size_t)iv * cols + iu + 1) * C + c] : val00;
    float val10 = (iv + 1 < rows) ? img[((size_t)(iv + 1) * cols + iu) * C + c] : val00;
    float val11 = (iu + 1 < cols && iv + 1 < rows) ? img[((size_t)(iv + 1) * cols + iu + 1) * C + c] : val00;
    float frac_uu = 1.0f - frac_u, frac_vv = 1.0f - frac_v;
    float w00 = frac_vv * frac_uu, w10 = frac_vv * frac_u, w01 = frac_v * frac_uu, w11 = frac_v * frac_u;
    return w00 * val00 + w10 * val01 + w01 * val10 + w11 * val11;
}

/* nearest_neighbor_interpolation, nearest.rs:15-30: round half away, saturating cast, clamp */
static inline float sample_nearest(const float* img, int rows, int cols, int C, float u, float v, int c) {
    float ru = roundf(u), rv = roundf(v);
    long iu = ru > 0.0f ? (long)ru : 0, iv = rv > 0.0f ? (long)rv : 0; /* `as usize` saturates at 0 */
    if (iu > cols - 1) iu = cols - 1;
    if (iv > rows - 1) iv = rows - 1;
    return img[((size_t)iv * cols + iu) * C + c];
}

/* keys_weights, bicubic.rs:15-28 (explicit mul_add = fmaf) */
static inline void keys_weights(float frac, float w[4]) {
    float t;
    t = 1.0f + frac; w[0] = fmaf(fmaf(fmaf(-0.5f, t, 2.5f), t, -4.0f), t, 2.0f);
    t = frac;        w[1] = fmaf(fmaf(1.5f, t, -2.5f) * t, t, 1.0f);
    t = 1.0f - frac; w[2] = fmaf(fmaf(1.5f, t, -2.5f) * t, t, 1.0f);
    t = 2.0f - frac; w[3] = fmaf(fmaf(fmaf(-0.5f, t, 2.5f), t, -4.0f), t, 2.0f);
}

/* bicubic_sample, bicubic.rs:33-62 */
static inline float sample_bicubic(const float* img, int rows, int cols, int C, float sx, float sy, int c) {
    float x0f = floorf(sx), y0f = floorf(sy);
    float wx[4], wy[4];
    keys_weights(sx - x0f, wx);
    keys_weights(sy - y0f, wy);
    long x0 = (long)x0f, y0 = (long)y0f;
    float acc = 0.0f;
    for (int dy = 0; dy < 4; ++dy) {
        long yi = y0 + dy - 1; if (yi < 0) yi = 0; if (yi > rows - 1) yi = rows - 1;
        for (int dx = 0; dx < 4; ++dx) {
            long xi = x0 + dx - 1; if (xi < 0) xi = 0; if (xi > cols - 1) xi = cols - 1;
            float w = wx[dx] * wy[dy];
            acc = fmaf(w, img[((size_t)yi * cols + xi) * C + c], acc);
        }
    }
    return acc;
}

/* ---- Lanczos-3 (P/interpolation/lanczos.rs) ---------------------------------------------------- */
#define KO_PI 3.14159265358979323846f /* std::f32::consts::PI */

/* sin_pi, lanczos.rs:19-37: integer reduction + odd polynomial in plain mul/add */
float ko_sin_pi(float x) {
    float k = roundf(x);
    float r = x - k;
    float z = KO_PI * r;
    float z2 = z * z;
    float p = -2.5052108e-8f;
    p = p * z2 + 2.7557319e-6f;
    p = p * z2 + -1.984127e-4f;
    p = p * z2 + 8.333334e-3f;
    p = p * z2 + -1.6666667e-1f;
    float s = z + z * z2 * p;
    return ((int)k & 1) ? -s : s;
}

/* lanczos3, lanczos.rs:40-51 */
float ko_lanczos3(float x) {
    if (fabsf(x) < 1e-5f) return 1.0f;
    if (fabsf(x) >= 3.0f) return 0.0f;
    float pix = KO_PI * x;
    float pix3 = pix * 0.33333334f;
    return ko_sin_pi(x) * ko_sin_pi(x * (1.0f / 3.0f)) / (pix * pix3);
}

/* lanczos_axis, lanczos.rs:59-101: tap base + six normalised weights per destination index */
static void lanczos_axis_ab(int src_len, int dst_len, float a, float b, int32_t* x0s, float* weights);
void ko_lanczos_axis(int src_len, int dst_len, int32_t* x0s, float* weights) {
    const float a = (float)src_len / (float)dst_len, b = 0.5f * a - 0.5f;
    lanczos_axis_ab(src_len, dst_len, a, b, x0s, weights);
}
static void lanczos_axis_ab(int src_len, int dst_len, float a, float b, int32_t* x0s, float* weights) {
    const float max = (float)(src_len - 1);
    for (int i = 0; i < dst_len; ++i) {
        float s = a * (float)i + b;
        s = s < 0.0f ? 0.0f : (s > max ? max : s);
        float x0 = floorf(s), frac = s - x0;
        x0s[i] = (int32_t)x0;
        float w[6] = {ko_lanczos3(frac + 2.0f), ko_lanczos3(frac + 1.0f), ko_lanczos3(frac),
                      ko_lanczos3(frac - 1.0f), ko_lanczos3(frac - 2.0f), ko_lanczos3(frac - 3.0f)};
        float sum = w[0] + w[1] + w[2] + w[3] + w[4] + w[5];
        float inv = 1.0f / sum;
        for (int t = 0; t < 6; ++t) weights[i * 6 + t] = w[t] * inv;
    }
}

/* lanczos3_weights, lanczos.rs:107-141: four sin_pi evaluations */
static inline float lz_den(float x) {
    float pix = KO_PI * x;
    float pix3 = pix * 0.33333334f;
    return pix * pix3;
}
void ko_lanczos3_weights(float frac, float w[6]) {
    float s = ko_sin_pi(frac);
    float t0 = ko_sin_pi(frac * (1.0f / 3.0f));
    float t1 = ko_sin_pi((frac - 1.0f) * (1.0f / 3.0f));
    float t2 = ko_sin_pi((frac - 2.0f) * (1.0f / 3.0f));
    float st0 = s * t0, st1 = s * t1, st2 = s * t2;
    w[0] = -st1 / lz_den(frac + 2.0f);
    w[1] = st2 / lz_den(frac + 1.0f);
    w[2] = st0 / lz_den(frac);
    w[3] = -st1 / lz_den(frac - 1.0f);
    w[4] = st2 / lz_den(frac - 2.0f);
    w[5] = st0 / lz_den(frac - 3.0f);
    if (frac < 1e-5f) w[2] = 1.0f;
    if (fabsf(frac - 1.0f) < 1e-5f) w[3] = 1.0f;
}

/* lanczos_sample, lanczos.rs:143-187: per-axis normalisation, two-level fma accumulation */
static inline float sample_lanczos(const float* img, int rows, int cols, int C, float sx, float sy, int c) {
    float x0f = floorf(sx), y0f = floorf(sy);
    float wx[6], wy[6];
    ko_lanczos3_weights(sx - x0f, wx);
    ko_lanczos3_weights(sy - y0f, wy);
    long x0 = (long)x0f, y0 = (long)y0f;
    float sum_wx = wx[0] + wx[1] + wx[2] + wx[3] + wx[4] + wx[5];
    float sum_wy = wy[0] + wy[1] + wy[2] + wy[3] + wy[4] + wy[5];
    float inv_x = 1.0f / sum_wx, inv_y = 1.0f / sum_wy;
    for (int t = 0; t < 6; ++t) { wx[t] *= inv_x; wy[t] *= inv_y; }
    float acc = 0.0f;
    for (int dy = 0; dy < 6; ++dy) {
        long yi = y0 + dy - 2; if (yi < 0) yi = 0; if (yi > rows - 1) yi = rows - 1;
        float rx = 0.0f;
        for (int dx = 0; dx < 6; ++dx) {
            long xi = x0 + dx - 2; if (xi < 0) xi = 0; if (xi > cols - 1) xi = cols - 1;
            rx = fmaf(wx[dx], img[((size_t)yi * cols + xi) * C + c], rx);
        }
        acc = fmaf(wy[dy], rx, acc);
    }
    return acc;
}

static inline float sample(int mode, const float* img, int rows, int cols, int C, float u, float v, int c) {
    switch (mode) {
        case 0: return sample_nearest(img, rows, cols, C, u, v, c);
        case 1: return sample_bilinear(img, rows, cols, C, u, v, c);
        case 2: return sample_bicubic(img, rows, cols, C, u, v, c);
        default: return sample_lanczos(img, rows, cols, C, u, v, c);
    }
}

/* resize_lanczos_separable, lanczos.rs:189-245: H pass into a dst_w x src_h f32 intermediate, then V */
static void resize_lanczos_separable_ab(const float* s, int sw, int sh, float* d, int dw, int dh, int C, const float cx[2], const float cy[2]);
static void resize_lanczos_separable(const float* s, int sw, int sh, float* d, int dw, int dh, int C) {
    const float ax = (float)sw / (float)dw, ay = (float)sh / (float)dh;
    const float cx[2] = {ax, 0.5f * ax - 0.5f}, cy[2] = {ay, 0.5f * ay - 0.5f};
    resize_lanczos_separable_ab(s, sw, sh, d, dw, dh, C, cx, cy);
}
static void resize_lanczos_separable_ab(const float* s, int sw, int sh, float* d, int dw, int dh, int C, const float cx[2], const float cy[2]) {
    int32_t* x0s = (int32_t*)malloc(sizeof(int32_t) * (size_t)(dw + dh));
    int32_t* y0s = x0s + dw;
    float* wx = (float*)malloc(sizeof(float) * 6 * (size_t)(dw + dh));
    float* wy = wx + 6 * (size_t)dw;
    lanczos_axis_ab(sw, dw, cx[0], cx[1], x0s, wx);
    lanczos_axis_ab(sh, dh, cy[0], cy[1], y0s, wy);
    float* inter = (float*)malloc(sizeof(float) * (size_t)dw * sh * C);
#pragma omp parallel for schedule(static)
    for (int sy = 0; sy < sh; ++sy)
        for (int dx = 0; dx < dw; ++dx)
            for (int k = 0; k < C; ++k) {
                float acc = 0.0f;
                for (int t = 0; t < 6; ++t) {
                    long xi = (long)x0s[dx] + t - 2; if (xi < 0) xi = 0; if (xi > sw - 1) xi = sw - 1;
                    acc = fmaf(wx[dx * 6 + t], s[((size_t)sy * sw + xi) * C + k], acc);
                }
                inter[((size_t)sy * dw + dx) * C + k] = acc;
            }
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy)
        for (int dx = 0; dx < dw; ++dx)
            for (int k = 0; k < C; ++k) {
                float acc = 0.0f;
                for (int t = 0; t < 6; ++t) {
                    long yi = (long)y0s[dy] + t - 2; if (yi < 0) yi = 0; if (yi > sh - 1) yi = sh - 1;
                    acc = fmaf(wy[dy * 6 + t], inter[((size_t)yi * dw + dx) * C + k], acc);
                }
                d[((size_t)dy * dw + dx) * C + k] = acc;
            }
    free(inter); free(wx); free(x0s);
}

/* ---- resize (P/resize/mod.rs:114-238) -------------------------------------------------------- */
static inline float fclamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* mode: 0 nearest, 1 bilinear, 2 bicubic, 3 lanczos */
void ko_resize_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int C, int mode) {
    if (sw == dw && sh == dh) { /* same-size short circuit, :134-137 */
        memcpy(dst, src, (size_t)sw * sh * C * sizeof(float));
        return;
    }
    if (mode == 3) { /* :139-146 */
        resize_lanczos_separable(src, sw, sh, dst, dw, dh, C);
        return;
    }
    const float ax = (float)sw / (float)dw, bx = 0.5f * ax - 0.5f;
    const float ay = (float)sh / (float)dh, by = 0.5f * ay - 0.5f;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        float sy = fclamp(ay * (float)y + by, 0.0f, (float)(sh - 1));
        for (int x = 0; x < dw; ++x) {
            float sx = fclamp(ax * (float)x + bx, 0.0f, (float)(sw - 1));
            for (int c = 0; c < C; ++c) dst[((size_t)y * dw + x) * C + c] = sample(mode, src, sh, sw, C, sx, sy, c);
        }
    }
}

/* ---- the resize LAUNCHERS' PixelMapping (P/cuda/resize.rs:433-473) and the fused resize + normalise kernel
 * (resize_bilinear_normalize_3c, :184-236; launcher :580-650).  mapping 0 = HalfPixel, 1 = AlignCorners.  The
 * launchers have no same-size short circuit; the sample is the bilinear sampler of `resize`, the epilogue the
 * kernel's `(ch - mean) * inv_std` with inv_std = 1.0f / std computed by the launcher (:621-623).            */
int ko_pixel_mapping_coeffs(int mapping, int src_len, int dst_len, float out[2]) {
    if (src_len <= 0 || dst_len <= 0) return -1;
    if (mapping == 0) {
        float a = (float)src_len / (float)dst_len;
        out[0] = a; out[1] = 0.5f * a - 0.5f;
    } else if (mapping == 1) {
        if (dst_len > 1) { out[0] = (float)(src_len - 1) / (float)(dst_len - 1); out[1] = 0.0f; }
        else { out[0] = 0.0f; out[1] = 0.0f; }
    } else return -1;
    return 0;
}

int ko_resize_mapped_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int C, int mode, int mapping) {
    float cx[2], cy[2];
    if (ko_pixel_mapping_coeffs(mapping, sw, dw, cx) || ko_pixel_mapping_coeffs(mapping, sh, dh, cy)) return -1;
    if (mode == 3) { resize_lanczos_separable_ab(src, sw, sh, dst, dw, dh, C, cx, cy); return 0; }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        float sy = fclamp(cy[0] * (float)y + cy[1], 0.0f, (float)(sh - 1));
        for (int x = 0; x < dw; ++x) {
            float sx = fclamp(cx[0] * (float)x + cx[1], 0.0f, (float)(sw - 1));
            for (int c = 0; c < C; ++c) dst[((size_t)y * dw + x) * C + c] = sample(mode, src, sh, sw, C, sx, sy, c);
        }
    }
    return 0;
}

int ko_resize_bilinear_normalize_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, const float mean[3],
                                     const float std_dev[3], int mapping) {
    float cx[2], cy[2];
    if (ko_pixel_mapping_coeffs(mapping, sw, dw, cx) || ko_pixel_mapping_coeffs(mapping, sh, dh, cy)) return -1;
    if (std_dev[0] == 0.0f || std_dev[1] == 0.0f || std_dev[2] == 0.0f) return -2;
    const float inv_std[3] = {1.0f / std_dev[0], 1.0f / std_dev[1], 1.0f / std_dev[2]};
    for (int y = 0; y < dh; ++y) {
        float sy = fclamp(cy[0] * (float)y + cy[1], 0.0f, (float)(sh - 1));
        for (int x = 0; x < dw; ++x) {
            float sx = fclamp(cx[0] * (float)x + cx[1], 0.0f, (float)(sw - 1));
            for (int c = 0; c < 3; ++c) {
                float ch = sample(1, src, sh, sw, 3, sx, sy, c);
                dst[((size_t)y * dw + x) * 3 + c] = (ch - mean[c]) * inv_std[c];
            }
        }
    }
    return 0;
}

/* ---- affine (P/warp/affine.rs:18-38 invert; :123-372 warp) ----------------------------------- */
void ko_invert_affine_transform(const float m[6], float out[6]) {
    float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5];
    float determinant = a * e - b * d;
    float inv = determinant != 0.0f ? 1.0f / determinant : 0.0f;
    float na = e * inv, nb = -b * inv, nd = -d * inv, ne = a * inv;
    out[0] = na; out[1] = nb; out[2] = -(na * c + nb * f);
    out[3] = nd; out[4] = ne; out[5] = -(nd * c + ne * f);
}

/* m = FORWARD 2x3 (src->dst), as the public API takes it. */
void ko_warp_affine_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int C, const float m[6], int mode) {
    float mi[6];
    ko_invert_affine_transform(m, mi);
    const float dsx = mi[0], dsy = mi[3], swf = (float)sw, shf = (float)sh;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        const float yf = (float)y;
        const float sx0 = mi[1] * yf + mi[2], sy0 = mi[4] * yf + mi[5];
        for (int x = 0; x < dw; ++x) {
            float* o = dst + ((size_t)y * dw + x) * C;
            const float sx = dsx * (float)x + sx0, sy = dsy * (float)x + sy0;
            /* in_bounds, :201-215 (degenerate-axis rule) */
            int x_ok = fabsf(dsx) < 1e-6f ? (sx0 >= 0.0f && sx0 < swf) : (sx >= 0.0f && sx < swf);
            int y_ok = fabsf(dsy) < 1e-6f ? (sy0 >= 0.0f && sy0 < shf) : (sy >= 0.0f && sy < shf);
            if (!(x_ok && y_ok)) { for (int c = 0; c < C; ++c) o[c] = 0.0f; continue; }
            if (mode == 0) { /* :270-276 */
                size_t xi = (size_t)fclamp(roundf(sx), 0.0f, swf - 1.0f), yi = (size_t)fclamp(roundf(sy), 0.0f, shf - 1.0f);
                for (int c = 0; c < C; ++c) o[c] = src[(yi * sw + xi) * C + c];
            } else if (mode == 1) { /* inlined sampler, :281-318 */
                float sxc = fclamp(sx, 0.0f, swf - 1.0f), syc = fclamp(sy, 0.0f, shf - 1.0f);
                size_t x0 = (size_t)sxc, y0 = (size_t)syc;
                size_t x1 = x0 + 1 < (size_t)sw - 1 ? x0 + 1 : (size_t)sw - 1;
                size_t y1 = y0 + 1 < (size_t)sh - 1 ? y0 + 1 : (size_t)sh - 1;
                float fx = sxc - (float)x0, fy = syc - (float)y0;
                float w00 = (1.0f - fy) * (1.0f - fx), w10 = (1.0f - fy) * fx, w01 = fy * (1.0f - fx), w11 = fy * fx;
                size_t b00 = (y0 * sw + x0) * C, b10 = (y0 * sw + x1) * C, b01 = (y1 * sw + x0) * C, b11 = (y1 * sw + x1) * C;
                for (int c = 0; c < C; ++c) o[c] = w00 * src[b00 + c] + w10 * src[b10 + c] + w01 * src[b01 + c] + w11 * src[b11 + c];
            } else { /* per-pixel samplers, :322-362 */
                for (int c = 0; c < C; ++c) o[c] = sample(mode, src, sh, sw, C, sx, sy, c);
            }
        }
    }
}

/* ---- perspective (P/warp/perspective.rs:12-72 invert, :115-166 warp) -------------------------- */
int ko_invert_homography(const float m[9], float inv[9]) {
    float det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    float h8_sq = m[8] * m[8];
    float det_norm = h8_sq > 1.1920929e-7f ? det / (h8_sq * fabsf(m[8])) : det;
    if (fabsf(det_norm) < 1e-10f) return 0;
    float adj[9] = {
        m[4] * m[8] - m[5] * m[7], m[2] * m[7] - m[1] * m[8], m[1] * m[5] - m[2] * m[4],
        m[5] * m[6] - m[3] * m[8], m[0] * m[8] - m[2] * m[6], m[2] * m[3] - m[0] * m[5],
        m[3] * m[7] - m[4] * m[6], m[1] * m[6] - m[0] * m[7], m[0] * m[4] - m[1] * m[3],
    };
    float inv_det = 1.0f / det;
    for (int i = 0; i < 9; ++i) inv[i] = adj[i] * inv_det;
    return 1;
}

/* m = FORWARD 3x3.  Returns 0 if singular (CannotComputeDeterminant). */
int ko_warp_perspective_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int C, const float m[9], int mode) {
    float im[9];
    if (!ko_invert_homography(m, im)) return 0;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        for (int x = 0; x < dw; ++x) {
            float* o = dst + ((size_t)y * dw + x) * C;
            const float xf = (float)x, yf = (float)y;
            /* transform_point, :67-72 */
            const float w = im[6] * xf + im[7] * yf + im[8];
            const float u = (im[0] * xf + im[1] * yf + im[2]) / w;
            const float v = (im[3] * xf + im[4] * yf + im[5]) / w;
            if (u >= 0.0f && u < (float)sw && v >= 0.0f && v < (float)sh) {
                for (int c = 0; c < C; ++c) o[c] = sample(mode, src, sh, sw, C, u, v, c);
            } else {
                for (int c = 0; c < C; ++c) o[c] = 0.0f;
            }
        }
    }
    return 1;
}

/* ---- remap (P/interpolation/remap.rs:43-107) -------------------------------------------------- */
void ko_remap_f32(const float* src, int sw, int sh, const float* map_x, const float* map_y, float* dst, int dw, int dh, int C, int mode) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        for (int x = 0; x < dw; ++x) {
            const size_t i = (size_t)y * dw + x;
            const float u = map_x[i], v = map_y[i];
            float* o = dst + i * C;
            if (u >= 0.0f && u < (float)sw && v >= 0.0f && v < (float)sh) {
                for (int c = 0; c < C; ++c) o[c] = sample(mode, src, sh, sw, C, u, v, c);
            } else {
                for (int c = 0; c < C; ++c) o[c] = 0.0f;
            }
        }
    }
}

/* ---- undistort maps (P/calibration/distortion.rs:68-152) ------------------------------------- */
/* intr = {fx, fy, cx, cy}; dist = {k1,k2,k3,k4,k5,k6,p1,p2}; all f64, cast to f32 at the end */
void ko_correction_map_polynomial(const double intr[4], const double dist[8], int w, int h, float* map_x, float* map_y) {
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const double k1 = dist[0], k2 = dist[1], k3 = dist[2], k4 = dist[3], k5 = dist[4], k6 = dist[5], p1 = dist[6], p2 = dist[7];
#pragma omp parallel for schedule(static)
    for (int yy = 0; yy < h; ++yy) {
        for (int xx = 0; xx < w; ++xx) {
            double x = ((double)xx - cx) / fx, y = ((double)yy - cy) / fy;
            double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
            double kr = (1.0 + k1 * r2 + k2 * r4 + k3 * r6) / (1.0 + k4 * r2 + k5 * r4 + k6 * r6);
            double x_2 = 2.0 * x, y_2 = 2.0 * y, xy_2 = x_2 * y;
            double xd = x * kr + xy_2 * p1 + p2 * (r2 + x_2 * x);
            double yd = y * kr + p1 * (r2 + y_2 * y) + xy_2 * p2;
            map_x[(size_t)yy * w + xx] = (float)(fx * xd + cx);
            map_y[(size_t)yy * w + xx] = (float)(fy * yd + cy);
        }
    }
}
