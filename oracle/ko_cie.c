/*
 * CPU oracle — CIE colour spaces (sRGB transfer, XYZ, L*a*b*, L*u*v*).  TEST INFRASTRUCTURE.
 *
 * P/color/cie/kernels.rs:21-340 and transfer.rs:14-45: `ko_cie_f32` restates the f32 scalar per-pixel path
 * (x86 fallback / SIMD tail), `ko_cie_f64` the f64 "exact formula" functions the reference's own tests
 * compare against (mod.rs:150-250: linear / xyz 5e-4, lab [1e-2, 2e-2, 2e-2], luv [1e-2, 5e-2, 5e-2]).
 * conv: 0 linear_rgb_from_rgb, 1 rgb_from_linear_rgb, 2 xyz_from_rgb, 3 rgb_from_xyz, 4 lab_from_rgb,
 * 5 rgb_from_lab, 6 luv_from_rgb, 7 rgb_from_luv.
 */
#include <math.h>
#include <stddef.h>

#include "ko_oracle.h"

static const float M_RGB2XYZ[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f, 0.072169f, 0.019334f, 0.119193f, 0.950227f};
static const float M_XYZ2RGB[9] = {3.240479f, -1.537150f, -0.498535f, -0.969256f, 1.875991f, 0.041556f, 0.055648f, -0.204043f, 1.057311f};
#define XN 0.950456f
#define YN 1.0f
#define ZN 1.088754f
#define INV_XN (1.0f / XN)
#define INV_ZN (1.0f / ZN)
#define LAB_DELTA 0.008856f
#define LAB_F_SLOPE (1.0f / 0.12841855f)
#define LAB_F_OFFSET 0.13793103f
#define LAB_FINV_THRESH 0.20689655f
#define LAB_FINV_SLOPE 0.12841855f
#define LUV_UN 0.19793943f
#define LUV_VN 0.46831096f
#define LUV_KAPPA 903.3f

/* ---- f32 scalar path ---------------------------------------------------------------------------- */
static float srgb_to_linear(float x) { x = x > 0.0f ? x : 0.0f; return x <= 0.04045f ? x * (1.0f / 12.92f) : powf((x + 0.055f) * (1.0f / 1.055f), 2.4f); }
static float linear_to_srgb(float l) { l = l > 0.0f ? l : 0.0f; return l <= 0.0031308f ? l * 12.92f : 1.055f * powf(l, 1.0f / 2.4f) - 0.055f; }
static void matvec32(const float* m, float a, float b, float c, float* o) {
    o[0] = m[0] * a + m[1] * b + m[2] * c; o[1] = m[3] * a + m[4] * b + m[5] * c; o[2] = m[6] * a + m[7] * b + m[8] * c;
}
static float lab_f32(float t) { return t > LAB_DELTA ? cbrtf(t) : t * LAB_F_SLOPE + LAB_F_OFFSET; }
static float lab_finv32(float f) { return f > LAB_FINV_THRESH ? f * f * f : LAB_FINV_SLOPE * (f - LAB_F_OFFSET); }
static void lin_xyz32(const float* in, float* o) { matvec32(M_RGB2XYZ, srgb_to_linear(in[0]), srgb_to_linear(in[1]), srgb_to_linear(in[2]), o); }
static void rgb_from_lin_xyz32(float x, float y, float z, float* out) {
    float l[3];
    matvec32(M_XYZ2RGB, x, y, z, l);
    out[0] = linear_to_srgb(l[0]); out[1] = linear_to_srgb(l[1]); out[2] = linear_to_srgb(l[2]);
}

void ko_cie_f32(const float* src, float* dst, size_t npixels, int conv) {
    for (size_t i = 0; i < npixels; ++i) {
        const float* in = src + 3 * i;
        float* out = dst + 3 * i;
        float q[3];
        switch (conv) {
            case 0: for (int c = 0; c < 3; ++c) out[c] = srgb_to_linear(in[c]); break;
            case 1: for (int c = 0; c < 3; ++c) out[c] = linear_to_srgb(in[c]); break;
            case 2: matvec32(M_RGB2XYZ, in[0], in[1], in[2], out); break;
            case 3: matvec32(M_XYZ2RGB, in[0], in[1], in[2], out); break;
            case 4: {
                lin_xyz32(in, q);
                float fx = lab_f32(q[0] * INV_XN), fy = lab_f32(q[1]), fz = lab_f32(q[2] * INV_ZN);
                out[0] = 116.0f * fy - 16.0f; out[1] = 500.0f * (fx - fy); out[2] = 200.0f * (fy - fz);
                break;
            }
            case 5: {
                float fy = (in[0] + 16.0f) / 116.0f, fx = fy + in[1] / 500.0f, fz = fy - in[2] / 200.0f;
                rgb_from_lin_xyz32(XN * lab_finv32(fx), YN * lab_finv32(fy), ZN * lab_finv32(fz), out);
                break;
            }
            case 6: {
                lin_xyz32(in, q);
                float yr = q[1];
                float l = yr > LAB_DELTA ? 116.0f * cbrtf(yr) - 16.0f : LUV_KAPPA * yr;
                float d = q[0] + 15.0f * q[1] + 3.0f * q[2];
                float up = d == 0.0f ? 0.0f : 4.0f * q[0] / d, vp = d == 0.0f ? 0.0f : 9.0f * q[1] / d;
                out[0] = l; out[1] = 13.0f * l * (up - LUV_UN); out[2] = 13.0f * l * (vp - LUV_VN);
                break;
            }
            default: {
                float l = in[0];
                if (l <= 0.0f) { rgb_from_lin_xyz32(0.0f, 0.0f, 0.0f, out); break; }
                float y;
                if (l > 8.0f) { float t = (l + 16.0f) / 116.0f; y = YN * t * t * t; } else y = YN * l / LUV_KAPPA;
                float inv13l = 1.0f / (13.0f * l);
                float up = in[1] * inv13l + LUV_UN, vp = in[2] * inv13l + LUV_VN;
                float x = y * 9.0f * up / (4.0f * vp);
                float z = y * (12.0f - 3.0f * up - 20.0f * vp) / (4.0f * vp);
                rgb_from_lin_xyz32(x, y, z, out);
            }
        }
    }
}

/* ---- f64 "exact formula" oracle (kernels.rs:64-215) ------------------------------------------------ */
static double lin64(double x) { x = x > 0.0 ? x : 0.0; return x <= (double)0.04045f ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4); }
static double srgb64(double l) { l = l > 0.0 ? l : 0.0; return l <= (double)0.0031308f ? 12.92 * l : 1.055 * pow(l, 1.0 / 2.4) - 0.055; }
static void matvec64(const float* m, double a, double b, double c, double* o) {
    o[0] = (double)m[0] * a + (double)m[1] * b + (double)m[2] * c;
    o[1] = (double)m[3] * a + (double)m[4] * b + (double)m[5] * c;
    o[2] = (double)m[6] * a + (double)m[7] * b + (double)m[8] * c;
}
static double lab_f64(double t) { return t > (double)LAB_DELTA ? cbrt(t) : t * (double)LAB_F_SLOPE + (double)LAB_F_OFFSET; }
static double lab_finv64(double f) { return f > (double)LAB_FINV_THRESH ? f * f * f : (double)LAB_FINV_SLOPE * (f - (double)LAB_F_OFFSET); }
static void lin_xyz64(const double* in, double* o) { matvec64(M_RGB2XYZ, lin64(in[0]), lin64(in[1]), lin64(in[2]), o); }
static void rgb_from_lin_xyz64(double x, double y, double z, double* out) {
    double l[3];
    matvec64(M_XYZ2RGB, x, y, z, l);
    out[0] = srgb64(l[0]); out[1] = srgb64(l[1]); out[2] = srgb64(l[2]);
}

void ko_cie_f64(const double* src, double* dst, size_t npixels, int conv) {
    for (size_t i = 0; i < npixels; ++i) {
        const double* in = src + 3 * i;
        double* out = dst + 3 * i;
        double q[3];
        switch (conv) {
            case 0: for (int c = 0; c < 3; ++c) out[c] = lin64(in[c]); break;
            case 1: for (int c = 0; c < 3; ++c) out[c] = srgb64(in[c]); break;
            case 2: matvec64(M_RGB2XYZ, in[0], in[1], in[2], out); break;
            case 3: matvec64(M_XYZ2RGB, in[0], in[1], in[2], out); break;
            case 4: {
                lin_xyz64(in, q);
                double fx = lab_f64(q[0] / (double)XN), fy = lab_f64(q[1] / (double)YN), fz = lab_f64(q[2] / (double)ZN);
                out[0] = 116.0 * fy - 16.0; out[1] = 500.0 * (fx - fy); out[2] = 200.0 * (fy - fz);
                break;
            }
            case 5: {
                double fy = (in[0] + 16.0) / 116.0, fx = fy + in[1] / 500.0, fz = fy - in[2] / 200.0;
                rgb_from_lin_xyz64((double)XN * lab_finv64(fx), (double)YN * lab_finv64(fy), (double)ZN * lab_finv64(fz), out);
                break;
            }
            case 6: {
                lin_xyz64(in, q);
                double yr = q[1] / (double)YN;
                double l = yr > (double)LAB_DELTA ? 116.0 * cbrt(yr) - 16.0 : (double)LUV_KAPPA * yr;
                double d = q[0] + 15.0 * q[1] + 3.0 * q[2];
                double up = d == 0.0 ? 0.0 : 4.0 * q[0] / d, vp = d == 0.0 ? 0.0 : 9.0 * q[1] / d;
                out[0] = l; out[1] = 13.0 * l * (up - (double)LUV_UN); out[2] = 13.0 * l * (vp - (double)LUV_VN);
                break;
            }
            default: {
                double l = in[0];
                if (l <= 0.0) { rgb_from_lin_xyz64(0.0, 0.0, 0.0, out); break; }
                double y = l > 8.0 ? (double)YN * pow((l + 16.0) / 116.0, 3.0) : (double)YN * l / (double)LUV_KAPPA;
                double up = in[1] / (13.0 * l) + (double)LUV_UN, vp = in[2] / (13.0 * l) + (double)LUV_VN;
                double x = y * 9.0 * up / (4.0 * vp);
                double z = y * (12.0 - 3.0 * up - 20.0 * vp) / (4.0 * vp);
                rgb_from_lin_xyz64(x, y, z, out);
            }
        }
    }
}
