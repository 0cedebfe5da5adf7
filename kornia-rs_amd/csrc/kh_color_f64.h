// f64 colour conversions — the per-pixel arithmetic of the reference's f64 public API, shared by the device kernel
// (kh_color_f64.hip) and a host harness (tests/cpp/color_f64_host.cpp) that checks the same source on a CPU.
// Plain IEEE double arithmetic, no contraction (both compiles use -ffp-contract=off):
//   gray / rgb_from_gray   P/color/gray/mod.rs:41-49 (0.299 r + 0.587 g + 0.114 b, left to right)
//   hsv / hls              P/color/hsv/mod.rs:64-113, P/color/hls/mod.rs:64-130 ([0,255] domain, `%` = fmod)
//   YCbCr / YUV            P/color/yuv/mod.rs:95-145 (one constant set for both chroma orders, true divisions)
//   sRGB transfer, XYZ, L*a*b*, L*u*v*   P/color/cie/kernels.rs:64-215 (the `*_scalar64_px` functions: matrices and
//                          thresholds are the f32 constants widened to double, the transfer uses double literals)
#pragma once

#include <math.h>

#ifdef __HIPCC__
#define KH_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define KH_HD inline
#endif

namespace kh_f64 {

// conversion codes of kh_color_convert_f64; 0..7 follow KH_CIE_*
enum {
    kLinearFromRgb = 0, kRgbFromLinear = 1, kXyzFromRgb = 2, kRgbFromXyz = 3, kLabFromRgb = 4, kRgbFromLab = 5,
    kLuvFromRgb = 6, kRgbFromLuv = 7, kGrayFromRgb = 8, kRgbFromGray = 9, kHsvFromRgb = 10, kRgbFromHsv = 11,
    kHlsFromRgb = 12, kRgbFromHls = 13, kYcbcrFromRgb = 14, kRgbFromYcbcr = 15, kYuvFromRgb = 16, kRgbFromYuv = 17,
    kCount = 18
};
KH_HD int channels_in(int conv) { return conv == kRgbFromGray ? 1 : 3; }
KH_HD int channels_out(int conv) { return conv == kGrayFromRgb ? 1 : 3; }

// ---- hue helpers -------------------------------------------------------------------------------------------
KH_HD double hue_degrees(double r, double g, double b, double maxv, double delta) {
    double h;
    if (maxv == r) h = 60.0 * fmod((g - b) / delta, 6.0);
    else if (maxv == g) h = 60.0 * (((b - r) / delta) + 2.0);
    else h = 60.0 * (((r - g) / delta) + 4.0);
    return h < 0.0 ? h + 360.0 : h;
}
KH_HD void hsv_from_rgb(const double* in, double* out) {
    const double r = in[0] / 255.0, g = in[1] / 255.0, b = in[2] / 255.0;
    const double maxv = fmax(fmax(r, g), b), minv = fmin(fmin(r, g), b), delta = maxv - minv;
    const double h = delta == 0.0 ? 0.0 : hue_degrees(r, g, b, maxv, delta);
    const double s = maxv == 0.0 ? 0.0 : (delta / maxv) * 255.0;
    out[0] = (h / 360.0) * 255.0; out[1] = s; out[2] = maxv * 255.0;
}
KH_HD void rgb_from_hsv(const double* in, double* out) {
    const double s = in[1] / 255.0, v = in[2] / 255.0;
    const double hh = (in[0] / 255.0) * 6.0;
    const double c = v * s;
    const double hmod2 = hh - 2.0 * floor(hh * 0.5);
    const double x = c * (1.0 - fabs(hmod2 - 1.0));
    const double m = v - c;
    // `hh.floor() as i32` saturates and maps NaN to 0; everything outside 0..4 takes the last arm
    const int sector = hh != hh ? 0 : ((hh >= 0.0 && hh < 5.0) ? (int)floor(hh) : 5);
    double r1, g1, b1;
    switch (sector) {
        case 0: r1 = c; g1 = x; b1 = 0.0; break;
        case 1: r1 = x; g1 = c; b1 = 0.0; break;
        case 2: r1 = 0.0; g1 = c; b1 = x; break;
        case 3: r1 = 0.0; g1 = x; b1 = c; break;
        case 4: r1 = x; g1 = 0.0; b1 = c; break;
        default: r1 = c; g1 = 0.0; b1 = x; break;
    }
    out[0] = (r1 + m) * 255.0; out[1] = (g1 + m) * 255.0; out[2] = (b1 + m) * 255.0;
}
KH_HD void hls_from_rgb(const double* in, double* out) {
    const double r = in[0] / 255.0, g = in[1] / 255.0, b = in[2] / 255.0;
    const double maxv = fmax(fmax(r, g), b), minv = fmin(fmin(r, g), b);
    const double diff = maxv - minv, sum = maxv + minv, l = sum * 0.5;
    double h = 0.0, s = 0.0;
    if (diff != 0.0) {
        s = l <= 0.5 ? diff / sum : diff / (2.0 - sum);
        h = hue_degrees(r, g, b, maxv, diff);
    }
    out[0] = (h / 360.0) * 255.0; out[1] = l * 255.0; out[2] = s * 255.0;
}
KH_HD double hue2rgb(double p, double q, double t) {
    if (t < 0.0) t = t + 1.0;
    if (t > 1.0) t = t - 1.0;
    if (t < 1.0 / 6.0) return p + (q - p) * 6.0 * t;
    if (t < 0.5) return q;
    if (t < 2.0 / 3.0) return p + (q - p) * (2.0 / 3.0 - t) * 6.0;
    return p;
}
KH_HD void rgb_from_hls(const double* in, double* out) {
    const double l = in[1] / 255.0, s = in[2] / 255.0;
    if (s == 0.0) { out[0] = out[1] = out[2] = l * 255.0; return; }
    const double h_deg = (in[0] / 255.0) * 360.0;
    const double q = l < 0.5 ? l * (1.0 + s) : l + s - l * s;
    const double p = 2.0 * l - q;
    const double hk = h_deg / 360.0;
    out[0] = hue2rgb(p, q, hk + 1.0 / 3.0) * 255.0;
    out[1] = hue2rgb(p, q, hk) * 255.0;
    out[2] = hue2rgb(p, q, hk - 1.0 / 3.0) * 255.0;
}

// ---- Y'CbCr family ------------------------------------------------------------------------------------------
KH_HD void ycc_from_rgb(const double* in, double* out, bool cb_first) {
    const double r = in[0], g = in[1], b = in[2];
    const double y = 0.299 * r + 0.587 * g + 0.114 * b;
    const double cr = (r - y) * 0.713 + 0.5, cb = (b - y) * 0.564 + 0.5;
    out[0] = y; out[1] = cb_first ? cb : cr; out[2] = cb_first ? cr : cb;
}
KH_HD void rgb_from_ycc(const double* in, double* out, bool cb_first) {
    const double y = in[0], cr = cb_first ? in[2] : in[1], cb = cb_first ? in[1] : in[2];
    const double r = y + (cr - 0.5) / 0.713;
    const double b = y + (cb - 0.5) / 0.564;
    out[0] = r; out[1] = (y - 0.299 * r - 0.114 * b) / 0.587; out[2] = b;
}

// ---- CIE ----------------------------------------------------------------------------------------------------
KH_HD double srgb_to_linear(double x) {
    x = x > 0.0 ? x : 0.0;
    return x <= (double)0.04045f ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4);
}
KH_HD double linear_to_srgb(double l) {
    l = l > 0.0 ? l : 0.0;
    return l <= (double)0.0031308f ? 12.92 * l : 1.055 * pow(l, 1.0 / 2.4) - 0.055;
}
KH_HD void xyz_from_linear(double r, double g, double b, double* o) {
    o[0] = (double)0.412453f * r + (double)0.357580f * g + (double)0.180423f * b;
    o[1] = (double)0.212671f * r + (double)0.715160f * g + (double)0.072169f * b;
    o[2] = (double)0.019334f * r + (double)0.119193f * g + (double)0.950227f * b;
}
KH_HD void linear_from_xyz(double x, double y, double z, double* o) {
    o[0] = (double)3.240479f * x + (double)-1.537150f * y + (double)-0.498535f * z;
    o[1] = (double)-0.969256f * x + (double)1.875991f * y + (double)0.041556f * z;
    o[2] = (double)0.055648f * x + (double)-0.204043f * y + (double)1.057311f * z;
}
KH_HD double lab_f(double t) {
    return t > (double)0.008856f ? cbrt(t) : t * (double)(1.0f / 0.12841855f) + (double)0.13793103f;
}
KH_HD double lab_finv(double f) {
    return f > (double)0.20689655f ? f * f * f : (double)0.12841855f * (f - (double)0.13793103f);
}
KH_HD void srgb_from_xyz(double x, double y, double z, double* out) {
    double l[3];
    linear_from_xyz(x, y, z, l);
    out[0] = linear_to_srgb(l[0]); out[1] = linear_to_srgb(l[1]); out[2] = linear_to_srgb(l[2]);
}
constexpr double kXn = (double)0.950456f, kYn = (double)1.0f, kZn = (double)1.088754f;
constexpr double kUn = (double)0.19793943f, kVn = (double)0.46831096f, kKappa = (double)903.3f;

KH_HD void cie(int conv, const double* in, double* out) {
    double q[3];
    switch (conv) {
        case kLinearFromRgb: out[0] = srgb_to_linear(in[0]); out[1] = srgb_to_linear(in[1]); out[2] = srgb_to_linear(in[2]); break;
        case kRgbFromLinear: out[0] = linear_to_srgb(in[0]); out[1] = linear_to_srgb(in[1]); out[2] = linear_to_srgb(in[2]); break;
        case kXyzFromRgb: xyz_from_linear(in[0], in[1], in[2], out); break;   // no transfer (kernels.rs:124-131)
        case kRgbFromXyz: linear_from_xyz(in[0], in[1], in[2], out); break;
        case kLabFromRgb: {
            xyz_from_linear(srgb_to_linear(in[0]), srgb_to_linear(in[1]), srgb_to_linear(in[2]), q);
            const double fx = lab_f(q[0] / kXn), fy = lab_f(q[1] / kYn), fz = lab_f(q[2] / kZn);
            out[0] = 116.0 * fy - 16.0; out[1] = 500.0 * (fx - fy); out[2] = 200.0 * (fy - fz);
            break;
        }
        case kRgbFromLab: {
            const double fy = (in[0] + 16.0) / 116.0, fx = fy + in[1] / 500.0, fz = fy - in[2] / 200.0;
            srgb_from_xyz(kXn * lab_finv(fx), kYn * lab_finv(fy), kZn * lab_finv(fz), out);
            break;
        }
        case kLuvFromRgb: {
            xyz_from_linear(srgb_to_linear(in[0]), srgb_to_linear(in[1]), srgb_to_linear(in[2]), q);
            const double yr = q[1] / kYn;
            const double l = yr > (double)0.008856f ? 116.0 * cbrt(yr) - 16.0 : kKappa * yr;
            const double d = q[0] + 15.0 * q[1] + 3.0 * q[2];
            const double up = d == 0.0 ? 0.0 : 4.0 * q[0] / d, vp = d == 0.0 ? 0.0 : 9.0 * q[1] / d;
            out[0] = l; out[1] = 13.0 * l * (up - kUn); out[2] = 13.0 * l * (vp - kVn);
            break;
        }
        default: {  // kRgbFromLuv
            const double l = in[0];
            if (l <= 0.0) { srgb_from_xyz(0.0, 0.0, 0.0, out); break; }
            const double y = l > 8.0 ? kYn * pow((l + 16.0) / 116.0, 3.0) : kYn * l / kKappa;
            const double up = in[1] / (13.0 * l) + kUn, vp = in[2] / (13.0 * l) + kVn;
            const double x = y * 9.0 * up / (4.0 * vp);
            const double z = y * (12.0 - 3.0 * up - 20.0 * vp) / (4.0 * vp);
            srgb_from_xyz(x, y, z, out);
        }
    }
}

// one pixel of conversion `conv` (wave-uniform on the device): in[channels_in], out[channels_out]
KH_HD void convert_pixel(int conv, const double* in, double* out) {
    switch (conv) {
        case kGrayFromRgb: out[0] = 0.299 * in[0] + 0.587 * in[1] + 0.114 * in[2]; break;
        case kRgbFromGray: out[0] = out[1] = out[2] = in[0]; break;
        case kHsvFromRgb: hsv_from_rgb(in, out); break;
        case kRgbFromHsv: rgb_from_hsv(in, out); break;
        case kHlsFromRgb: hls_from_rgb(in, out); break;
        case kRgbFromHls: rgb_from_hls(in, out); break;
        case kYcbcrFromRgb: ycc_from_rgb(in, out, false); break;
        case kRgbFromYcbcr: rgb_from_ycc(in, out, false); break;
        case kYuvFromRgb: ycc_from_rgb(in, out, true); break;
        case kRgbFromYuv: rgb_from_ycc(in, out, true); break;
        default: cie(conv, in, out); break;
    }
}

}  // namespace kh_f64
