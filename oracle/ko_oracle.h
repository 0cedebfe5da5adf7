/*
 * ko_oracle.h — CPU oracle for the kornia-rs imgproc hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load libkornia_oracle.so; nothing under kornia-rs_amd/ links, imports
 * or calls it, and the product path fails loudly when the HIP library is missing.
 *
 * Every function is a plain-C restatement of the reference's scalar arithmetic (the reference
 * itself cannot be built here: no cargo/rustc).  Each cites the reference file:line it follows
 * (paths relative to /root/reference: P/ = crates/kornia-imgproc/src/).  Built with
 * `gcc -O2 -ffp-contract=off -fno-fast-math` so no multiply-add is fused unless the reference
 * asks for `mul_add` (written as fmaf here).
 *
 * Pinning: tests/test_oracle_*.py check these functions against every known-answer vector the
 * reference's own tests hold for the path — see tests/golden/README.md.
 */
#ifndef KO_ORACLE_H
#define KO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Number of OpenMP threads the oracle will use (1 when built without OpenMP). */
int ko_max_threads(void);
void ko_set_threads(int n);

/* LCG fixture generator shared by the reference's GPU parity tests
 * (P/cuda/color/mod.rs:303-321 pattern_u8 / pattern_f32).                                    */
void ko_pattern_u8(uint8_t* out, size_t n);
void ko_pattern_f32(float* out, size_t n);

/* ---- fused camera preprocess (P/preprocess.rs:430-622, the CUDA kernel's arithmetic) ------ */
typedef struct ko_preprocess_params {
    float scale_x, scale_y, pad_x, pad_y;
    int32_t src_w, src_h, src_pitch, src_bpp, fmt;
    int32_t dst_w, dst_h;
    float mean[3];
    float inv_std[3];
    float pad_value;
    int32_t sampling;  /* 0 nearest, 1 bilinear, 2 lanczos */
    int32_t out_dtype; /* 0 f32, 1 f16 bits */
    int32_t nframes;
    int32_t flags;
    int64_t src_frame_stride; /* bytes */
    int64_t dst_frame_stride; /* elements */
} ko_preprocess_params;

void ko_preprocess_to_chw(const uint8_t* src, void* dst, const ko_preprocess_params* p);
/* Affine::new (P/preprocess.rs:350-369): mode 0 = Letterbox, 1 = Stretch; out = {sx, sy, px, py} */
void ko_preprocess_affine(int mode, int sw, int sh, int dw, int dh, float out[4]);
/* the kernel's manual f32 -> binary16 RNE (P/preprocess.rs:452-477) */
uint16_t ko_f2h(float f);
/* see ko_preprocess.c: exhaustively-checked x/255 shortcut used by the device fast path */
float ko_div255_fma(float x);
long long ko_div255_fma_mismatches(uint32_t lo_bits, uint32_t hi_bits);
int ko_plan_div_mismatches(float pad, float scale, int count);

/* ---- colour: gray (P/color/gray/kernels.rs) ----------------------------------------------- */
void ko_gray_from_rgb_u8(const uint8_t* src, uint8_t* dst, size_t npixels);
void ko_gray_from_rgb_f32(const float* src, float* dst, size_t npixels);
void ko_rgb_from_gray_u8(const uint8_t* src, uint8_t* dst, size_t npixels);
void ko_rgb_from_gray_f32(const float* src, float* dst, size_t npixels);

/* ---- colour: video decode / encode (P/color/yuv/kernels.rs Family B / C) ------------------ */
/* layout: 0 NV12, 1 NV21, 2 I420, 3 YV12.  `buf` = Y plane then chroma, tightly packed.       */
void ko_rgb_from_planar420(const uint8_t* buf, uint8_t* dst, int width, int height, int layout);
/* layout: 0 YUYV, 1 UYVY, 2 YVYU */
void ko_rgb_from_packed422(const uint8_t* src, uint8_t* dst, int width, int height, int layout);
void ko_nv12_from_rgb(const uint8_t* src, uint8_t* dst, int width, int height);
void ko_yuyv_from_rgb(const uint8_t* src, uint8_t* dst, int width, int height);

/* ---- colour: full-range YCbCr/YUV (Family A), HSV/HLS, swizzles, sepia, LUT -------------- */
void ko_ycc_from_rgb_u8(const uint8_t* src, uint8_t* dst, size_t npixels, int order);
void ko_rgb_from_ycc_u8(const uint8_t* src, uint8_t* dst, size_t npixels, int order);
void ko_ycc_from_rgb_f32(const float* src, float* dst, size_t npixels, int order);
void ko_rgb_from_ycc_f32(const float* src, float* dst, size_t npixels, int order);
void ko_hsv_from_rgb_f32(const float* src, float* dst, size_t npixels);
void ko_rgb_from_hsv_f32(const float* src, float* dst, size_t npixels);
void ko_hls_from_rgb_f32(const float* src, float* dst, size_t npixels);
void ko_rgb_from_hls_f32(const float* src, float* dst, size_t npixels);
void ko_bgr_from_rgb_u8(const uint8_t* src, uint8_t* dst, size_t n);
void ko_bgr_from_rgb_f32(const float* src, float* dst, size_t n);
void ko_rgba_from_rgb_u8(const uint8_t* src, uint8_t* dst, size_t n, int swap_rb);
void ko_rgba_from_rgb_f32(const float* src, float* dst, size_t n, int swap_rb);
void ko_rgb_from_rgba_u8(const uint8_t* src, uint8_t* dst, size_t n, int swap_rb, const uint8_t* bg);
void ko_sepia_from_rgb_f32(const float* src, float* dst, size_t n);
void ko_sepia_from_rgb_u8(const uint8_t* src, uint8_t* dst, size_t n);
void ko_apply_colormap_u8(const uint8_t* src, uint8_t* dst, size_t n, const uint8_t* lut);

/* ---- geometry, f32 (ko_geom.c): mode 0 nearest, 1 bilinear, 2 bicubic --------------------- */
void ko_resize_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int C, int mode);
/* the launchers' PixelMapping (0 HalfPixel, 1 AlignCorners) and the fused resize + normalise kernel; -1 / -2 on bad arguments */
int ko_pixel_mapping_coeffs(int mapping, int src_len, int dst_len, float out[2]);
int ko_resize_mapped_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int C, int mode, int mapping);
int ko_resize_bilinear_normalize_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, const float mean[3],
                                     const float std_dev[3], int mapping);
void ko_invert_affine_transform(const float m[6], float out[6]);
void ko_warp_affine_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int C, const float m[6], int mode);
int ko_invert_homography(const float m[9], float inv[9]);
int ko_warp_perspective_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int C, const float m[9], int mode);
void ko_remap_f32(const float* src, int sw, int sh, const float* map_x, const float* map_y, float* dst, int dw, int dh, int C, int mode);
void ko_correction_map_polynomial(const double intr[4], const double dist[8], int w, int h, float* map_x, float* map_y);

/* ---- separable filters, f32 (ko_filter.c) ---------------------------------------------------- */
void ko_box_blur_kernel_1d(int n, float* out);
void ko_gaussian_kernel_1d(int n, float sigma, float* out);
int ko_gradient_kernels_1d(int kind, int n, float* kx, float* ky);
int ko_gaussian_resolve(int k[2], float s[2]);
void ko_separable_filter_f32(const float* src, float* dst, int cols, int rows, int C, const float* kx, int nx, const float* ky, int ny);
void ko_gradient_magnitude_f32(const float* src, float* dst, int cols, int rows, int C, const float* kx, const float* ky, int n);

float ko_sin_pi(float x);
float ko_lanczos3(float x);
void ko_lanczos3_weights(float frac, float w[6]);
void ko_lanczos_axis(int src_len, int dst_len, int32_t* x0s, float* weights);

/* ---- u8 fixed-point twins (ko_u8.c) ---------------------------------------------------------- */
void ko_quantize_kernel_256(const float* k, int n, uint8_t* out);
void ko_separable_blur_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C, const uint8_t* kx, int nx, const uint8_t* ky, int ny);
void ko_binomial3_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C);
int ko_gaussian_blur_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C, int kx, int ky, float sx, float sy);
int ko_box_blur_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C, int kx, int ky);
void ko_remap_u8(const uint8_t* src, int sw, int sh, const float* map_x, const float* map_y, uint8_t* dst, int dw, int dh, int C, int mode);
void ko_warp_affine_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C, const float m[6]);
int ko_warp_perspective_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C, const float m[9]);

/* ---- u8 resize cascade + OpenCV-compatible resize (ko_resize_u8.c) -------------------------------- */
int ko_resize_contribs(int src_size, int dst_size, int filt, int antialias, int32_t* offsets, int32_t* weights, int max_ksize);
int ko_resize_fast_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C, int mode, int antialias);
int ko_resize_opencv_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int C, int mode);
int ko_resize_opencv_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int C, int mode);
void ko_normalize_params(const float mean[3], const float std[3], float scale[3], float bias[3]);
int ko_resize_normalize_to_chw(const uint8_t* src, int sw, int sh, float* dst, int dw, int dh, const float scale[3],
                               const float bias[3], int mode, int antialias);

void ko_fused_pipeline(const uint8_t* src, int sw, int sh, int rdw, int rdh, int dw, int dh, const int* map_kinds,
                       const float* map_params, int nmaps, int sink, float* dst);

/* ---- CIE colour spaces (ko_cie.c) -------------------------------------------------------------------- */
void ko_cie_f32(const float* src, float* dst, size_t npixels, int conv);
void ko_cie_f64(const double* src, double* dst, size_t npixels, int conv);
/* rgb_from_bayer (P/color/bayer/mod.rs:37): pattern 0 RGGB, 1 BGGR, 2 GRBG, 3 GBRG; -1 on a bad argument */
int ko_rgb_from_bayer(const uint8_t* src, uint8_t* dst, int cols, int rows, int pattern);
/* convert_yuyv_to_rgb_u8 (P/color/yuv/mod.rs:342): mode 0 Bt601Full, 1 Bt709Full, 2 Bt601Limited; -1 on a bad argument */
int ko_yuyv_to_rgb_mode(const uint8_t* src, uint8_t* dst, int w, int h, int mode);
/* f64 colour family (ko_color_f64.c): conv 0..7 = ko_cie_f64, 8..17 gray / hsv / hls / ycbcr / yuv; -1 on an unknown code */
int ko_color_f64(const double* src, double* dst, size_t npixels, int conv);

/* ---- pyramid + morphology (ko_pyramid_morph.c) ------------------------------------------------------- */
void ko_pyrdown_f32(const float* src, int sw, int sh, float* dst, int C);
void ko_pyrup_f32(const float* src, int sw, int sh, float* dst, int C);
void ko_pyrdown_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int C);
void ko_pyrup_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int C);
void ko_morph_kernel(int shape, int width, int height, uint8_t* out);
void ko_morphology_u8(const uint8_t* src, int w, int h, int C, uint8_t* dst, int op, const uint8_t* mask, int kw, int kh,
                      int border, const uint8_t* cval);

/* ---- the rest of the filter module (ko_filter_extra.c) ------------------------------------------------ */
/* spatial_gradient_float / scharr_spatial_gradient_float (P/filter/ops.rs:287-590): kind 0 = Sobel, 1 = Scharr */
void ko_spatial_gradient_f32(const float* src, float* gx, float* gy, int cols, int rows, int C, int kind);
void ko_box_blur_fast_kernels_1d(float sigma, int kernels, int* out);
/* -1 where the reference would index out of bounds (half >= cols); dst is the TRANSPOSED image */
int ko_fast_horizontal_filter(const float* src, float* dst, int cols, int rows, int C, int half);
int ko_box_blur_fast_f32(const float* src, float* dst, int cols, int rows, int C, float sigma_x, float sigma_y);
/* median_blur (P/filter/median.rs:174): ksize 3 | 5, else -1 */
int ko_median_blur_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int C, int ksize);
/* bilateral_filter (P/filter/bilateral.rs): cv2's tables and accumulation order, single-channel u8 */
float ko_v_exp_f32(float x);
int ko_bilateral_tables(int d, double sigma_color, double sigma_space, int capacity, int* radius_out, int* tap_dy, int* tap_dx,
                        float* space_weight, float* color_weight, int* simd_order);
int ko_bilateral_filter_u8(const uint8_t* src, uint8_t* dst, int cols, int rows, int d, double sigma_color, double sigma_space);

/* normalize_mean_std (P/normalize.rs:56-87): (src - mean[c]) / std[c], IEEE division; rows in parallel like par_iter_rows */
void ko_normalize_mean_std_f32(const float* src, float* dst, int width, int height, int channels, const float* mean,
                               const float* std);

#ifdef __cplusplus
}
#endif
#endif
