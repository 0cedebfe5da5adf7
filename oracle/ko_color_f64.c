/*
 * CPU oracle — f64 colour conversions outside the CIE family.  TEST INFRASTRUCTURE: only tests/, smoke() and
 * bench.py's cpu_baseline may use it.
 *
 * Restates the reference's public f64 arms, written from the Rust sources (not from the product's device code):
 *   gray_from_rgb / rgb_from_gray   P/color/gray/mod.rs:41-49   `0.299*r + 0.587*g + 0.114*b`
 *   hsv_from_rgb / rgb_from_hsv     P/color/hsv/mod.rs:64-113   ([0,255] domain; Rust `%` on f64 is C fmod;
 *                                   f64::max / min ignore a NaN operand like fmax / fmin; `as i32` saturates, NaN -> 0)
 *   hls_from_rgb / rgb_from_hls     P/color/hls/mod.rs:64-130   (channel order H, L, S)
 *   ycbcr / yuv                     P/color/yuv/mod.rs:95-145   (the f64 arm uses 0.713 / 0.564 for BOTH chroma orders)
 * conv: 8 gray_from_rgb, 9 rgb_from_gray, 10 hsv_from_rgb, 11 rgb_from_hsv, 12 hls_from_rgb, 13 rgb_from_hls,
 *       14 ycbcr_from_rgb, 15 rgb_from_ycbcr, 16 yuv_from_rgb, 17 rgb_from_yuv   (0..7 = ko_cie_f64)
 */
#include <math.h>
#include <stddef.h>

#include "ko_oracle.h"

static double hue_of(double r, double g, double b, double max, double delta) {
    double h;
    if (max == r) h = 60.0 * fmod((g - b) / delta, 6.0);
    else if (max == g) h = 60.0 * (((b - r) / delta) + 2.0);
    else h = 60.0 * (((r - g) / delta) + 4.0);
    if (h < 0.0) h = h + 360.0;
    return h;
}

static void hsv_from_rgb64(const double* s, double* d) {
    double r = s[0] / 255.0, g = s[1] / 255.0, b = s[2] / 255.0;
    double max = fmax(fmax(r, g), b), min = fmin(fmin(r, g), b);
    double delta = max - min;
    double h = delta == 0.0 ? 0.0 : hue_of(r, g, b, max, delta);
    double sat = max == 0.0 ? 0.0 : (delta / max) * 255.0;
    d[0] = (h / 360.0) * 255.0; d[1] = sat; d[2] = max * 255.0;
}

static int rust_f64_as_i32(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)v;
}

static void rgb_from_hsv64(const double* in, double* d) {
    double s = in[1] / 255.0, v = in[2] / 255.0;
    double hh = (in[0] / 255.0) * 6.0;
    double c = v * s;
    double hmod2 = hh - 2.0 * floor(hh * 0.5);
    double x = c * (1.0 - fabs(hmod2 - 1.0));
    double m = v - c;
    double r1, g1, b1;
    switch (rust_f64_as_i32(floor(hh))) {
        case 0: r1 = c; g1 = x; b1 = 0.0; break;
        case 1: r1 = x; g1 = c; b1 = 0.0; break;
        case 2: r1 = 0.0; g1 = c; b1 = x; break;
        case 3: r1 = 0.0; g1 = x; b1 = c; break;
        case 4: r1 = x; g1 = 0.0; b1 = c; break;
        default: r1 = c; g1 = 0.0; b1 = x; break;
    }
    d[0] = (r1 + m) * 255.0; d[1] = (g1 + m) * 255.0; d[2] = (b1 + m) * 255.0;
}

static void hls_from_rgb64(const double* in, double* d) {
    double r = in[0] / 255.0, g = in[1] / 255.0, b = in[2] / 255.0;
    double max = fmax(fmax(r, g), b), min = fmin(fmin(r, g), b);
    double diff = max - min, sum = max + min;
    double l = sum * 0.5;
    double h = 0.0, s = 0.0;
    if (!(diff == 0.0)) {
        s = l <= 0.5 ? diff / sum : diff / (2.0 - sum);
        h = hue_of(r, g, b, max, diff);
    }
    d[0] = (h / 360.0) * 255.0; d[1] = l * 255.0; d[2] = s * 255.0;
}

static double hue2rgb64(double p, double q, double t) {
    t = t < 0.0 ? t + 1.0 : t;
    t = t > 1.0 ? t - 1.0 : t;
    if (t < 1.0 / 6.0) return p + (q - p) * 6.0 * t;
    if (t < 0.5) return q;
    if (t < 2.0 / 3.0) return p + (q - p) * (2.0 / 3.0 - t) * 6.0;
    return p;
}

static void rgb_from_hls64(const double* in, double* d) {
    double l = in[1] / 255.0, s = in[2] / 255.0;
    if (s == 0.0) { d[0] = l * 255.0; d[1] = l * 255.0; d[2] = l * 255.0; return; }
    double h_deg = (in[0] / 255.0) * 360.0;
    double q = l < 0.5 ? l * (1.0 + s) : l + s - l * s;
    double p = 2.0 * l - q;
    double hk = h_deg / 360.0;
    d[0] = hue2rgb64(p, q, hk + 1.0 / 3.0) * 255.0;
    d[1] = hue2rgb64(p, q, hk) * 255.0;
    d[2] = hue2rgb64(p, q, hk - 1.0 / 3.0) * 255.0;
}

static const double YR = 0.299, YG = 0.587, YB = 0.114, KCR = 0.713, KCB = 0.564;

static void ycc_from_rgb64(const double* s, double* d, int yuv_order) {
    double r = s[0], g = s[1], b = s[2];
    double y = YR * r + YG * g + YB * b;
    double cr = (r - y) * KCR + 0.5;
    double cb = (b - y) * KCB + 0.5;
    d[0] = y;
    if (yuv_order) { d[1] = cb; d[2] = cr; } else { d[1] = cr; d[2] = cb; }
}

static void rgb_from_ycc64(const double* s, double* d, int yuv_order) {
    double y = s[0];
    double cr = yuv_order ? s[2] : s[1], cb = yuv_order ? s[1] : s[2];
    double r = y + (cr - 0.5) / KCR;
    double b = y + (cb - 0.5) / KCB;
    double g = (y - YR * r - YB * b) / YG;
    d[0] = r; d[1] = g; d[2] = b;
}

int ko_color_f64(const double* src, double* dst, size_t npixels, int conv) {
    if (conv >= 0 && conv <= 7) { ko_cie_f64(src, dst, npixels, conv); return 0; }
    if (conv < 8 || conv > 17) return -1;
    for (size_t i = 0; i < npixels; ++i) {
        switch (conv) {
            case 8: dst[i] = 0.299 * src[3 * i] + 0.587 * src[3 * i + 1] + 0.114 * src[3 * i + 2]; break;
            case 9: dst[3 * i] = dst[3 * i + 1] = dst[3 * i + 2] = src[i]; break;
            case 10: hsv_from_rgb64(src + 3 * i, dst + 3 * i); break;
            case 11: rgb_from_hsv64(src + 3 * i, dst + 3 * i); break;
            case 12: hls_from_rgb64(src + 3 * i, dst + 3 * i); break;
            case 13: rgb_from_hls64(src + 3 * i, dst + 3 * i); break;
            case 14: ycc_from_rgb64(src + 3 * i, dst + 3 * i, 0); break;
            case 15: rgb_from_ycc64(src + 3 * i, dst + 3 * i, 0); break;
            case 16: ycc_from_rgb64(src + 3 * i, dst + 3 * i, 1); break;
            default: rgb_from_ycc64(src + 3 * i, dst + 3 * i, 1); break;
        }
    }
    return 0;
}
