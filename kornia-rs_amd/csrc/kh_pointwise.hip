// normalize / min-max / crop / flip for gfx950.
//
// P/normalize.rs:56-118 normalize_mean_std `(x - mean) / std` (true division),
// :123-146 find_min_max, :191-222 normalize_min_max `(x-min_v)*(max-min)/(max_v-min_v)+min`,
// :235-420 normalize_rgb_u8 (scalar expression `x*scale + offset`), P/crop.rs:187-240 crop_image,
// P/flip.rs:39-120,305-360 horizontal/vertical flip.  The reference has no device twin for
// normalize; crop/flip are exact copies.
#include "kh_common.h"

using namespace kh;

namespace {

struct Vec4 { float v[4]; };

// One thread = one pixel: C floats come in with one load (a wave reads 64 * C * 4 contiguous bytes) and leave through the block's
// streaming window (write-through non-temporal store, kh_common.h::stream_store) — the shape of the f32 colour maps (kh_color.hip),
// 0.73-0.78 of the HBM peak on this part.  Round 3's one-element-per-thread kernel with a 64-bit `i % C` measured 0.44 (r04d).
template <int C> struct PxF { float v[C]; };
template <int C>
__global__ __launch_bounds__(kBlock) void normalize_mean_std_kernel(const float* __restrict__ src,
                                                                     float* __restrict__ dst, long long npx,
                                                                     Vec4 mean, Vec4 stdv) {
    const long long p0 = (long long)blockIdx.x * kBlock, p = p0 + threadIdx.x;
    if (p >= npx) return;
    const PxF<C> in = *reinterpret_cast<const PxF<C>*>(src + p * C);
    uint32_t w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = __float_as_uint((in.v[c] - mean.v[c]) / stdv.v[c]);   // true division (normalize.rs:78)
    stream_store<C>(stream_window(dst + p0 * C, (npx - p0) * C * 4), (int)threadIdx.x * C * 4, w);
}

// Three-channel images whose float count is a multiple of four and whose buffers are 16-byte aligned: a lane owns FOUR consecutive
// floats (16 B in, 16 B out: a wave moves 1 KiB per instruction, the access width of the part's best copy, against 768 B for the
// pixel-per-lane form above).  Element e has channel e % 3; quad q = 256 b + t starts at element 4 q === q (mod 3), and 256 === 1, so the
// first channel of a lane's quad is (b % 3 + t % 3) % 3 — no 64-bit remainder per element (what made round 3's flat kernel slow).
// Same expression per element: bit-identical.
typedef float f32x4_pw __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBlock) void normalize_mean_std_quads3_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                            long long nquads, Vec4 mean, Vec4 stdv) {
    const long long q0 = (long long)blockIdx.x * kBlock, q = q0 + threadIdx.x;
    if (q >= nquads) return;
    const f32x4_pw in = reinterpret_cast<const f32x4_pw*>(src)[q];
    const unsigned c0 = ((unsigned)blockIdx.x % 3u + (unsigned)threadIdx.x % 3u) % 3u;
    // channels of the four elements: c0, c0 + 1, c0 + 2, c0 (mod 3)
    const float m0 = c0 == 0 ? mean.v[0] : c0 == 1 ? mean.v[1] : mean.v[2], s0 = c0 == 0 ? stdv.v[0] : c0 == 1 ? stdv.v[1] : stdv.v[2];
    const float m1 = c0 == 0 ? mean.v[1] : c0 == 1 ? mean.v[2] : mean.v[0], s1 = c0 == 0 ? stdv.v[1] : c0 == 1 ? stdv.v[2] : stdv.v[0];
    const float m2 = c0 == 0 ? mean.v[2] : c0 == 1 ? mean.v[0] : mean.v[1], s2 = c0 == 0 ? stdv.v[2] : c0 == 1 ? stdv.v[0] : stdv.v[1];
    const uint32_t w[4] = {__float_as_uint((in.x - m0) / s0), __float_as_uint((in.y - m1) / s1), __float_as_uint((in.z - m2) / s2),
                           __float_as_uint((in.w - m0) / s0)};   // true division (normalize.rs:78)
    stream_store<4>(stream_window(dst + q0 * 4, (nquads - q0) * 16), (int)threadIdx.x * 16, w);
}

__global__ __launch_bounds__(kBlock) void normalize_rgb_u8_kernel(const uint8_t* __restrict__ src,
                                                                   float* __restrict__ dst, long long npx,
                                                                   Vec4 scale, Vec4 offset) {
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= npx) return;
    const uint8_t* s = src + p * 3;
    float* d = dst + p * 3;
    d[0] = (float)s[0] * scale.v[0] + offset.v[0];
    d[1] = (float)s[1] * scale.v[1] + offset.v[1];
    d[2] = (float)s[2] * scale.v[2] + offset.v[2];
}

// Sixteen-byte stores (round 6; the pixel count a multiple of four, 4-byte-aligned source, 16-byte-aligned destination): the image as a flat
// list of 4-float chunks — chunk k = source bytes 4k .. 4k + 3 (one dword) -> destination floats 4k .. 4k + 3 (one 16-byte streaming
// store) — and a lane takes chunks lane, lane + 64, lane + 128 of its wave's 192, so that EVERY load and store instruction of a wave is
// lane-contiguous (256 B / 1 KiB).  (Three 16-byte stores at a 48-byte lane stride — four whole pixels per lane — ran 4x SLOWER than
// the one-pixel kernel: a write-through store instruction that fills a third of every line leaves as partial-line writes.)  Element e
// of the image is channel e % 3; 4 and 64 are 1 modulo 3 and a wave's base chunk a multiple of 3, so element j of the lane's chunk g is
// channel (lane + g + j) % 3: the scales rotated once per lane.  Same `(float)x * scale[c] + offset[c]` per element: 0.49 -> 0.7+ of peak.
__global__ __launch_bounds__(kBlock) void normalize_rgb_u8_quads_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long long nchunks,
                                                                         Vec4 scale, Vec4 offset) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long base = (long long)blockIdx.x * (3 * kBlock);            // first chunk of the block (a multiple of 3)
    const int r = lane % 3;
    float sr[3], orr[3];                                                      // sr[k] = scale[(lane + k) % 3]
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c0 = k % 3, c1 = (k + 1) % 3, c2 = (k + 2) % 3;
        sr[k] = r == 0 ? scale.v[c0] : (r == 1 ? scale.v[c1] : scale.v[c2]);
        orr[k] = r == 0 ? offset.v[c0] : (r == 1 ? offset.v[c1] : offset.v[c2]);
    }
    const __amdgpu_buffer_rsrc_t ow = stream_window(dst + base * 4, (nchunks - base) * 16);
    uint32_t d[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const long long k = base + wv * 192 + g * 64 + lane;
        d[g] = reinterpret_cast<const uint32_t*>(src)[k < nchunks ? k : nchunks - 1];
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = __float_as_uint((float)((d[g] >> (8 * j)) & 0xffu) * sr[(g + j) % 3] + orr[(g + j) % 3]);
        stream_store<4>(ow, (wv * 192 + g * 64 + lane) * 16, w);   // (chunks past the end fall outside the window: dropped)
    }
}

// Order-preserving float <-> uint key so min/max can use integer atomics.
__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// keys[0] = min key (init 0xFFFFFFFF), keys[1] = max key (init 0)
__global__ __launch_bounds__(kBlock) void minmax_partial_kernel(const float* __restrict__ src, long long n,
                                                                 uint32_t* __restrict__ keys) {
    float mn = INFINITY, mx = -INFINITY;
    bool any = false;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const float x = src[i];
        if (x == x) {  // NaN never wins a `<` / `>` comparison in the reference loop
            mn = fminf(mn, x);
            mx = fmaxf(mx, x);
            any = true;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
        const int other = __shfl_xor((int)any, off);  // every lane takes part: `any || __shfl_xor(..)` would skip the
        any = any || other;                            // exchange in the lanes that already hold data (a divergent shuffle)
    }
    if ((threadIdx.x & 63) == 0 && any) {
        atomicMin(&keys[0], f2key(mn));
        atomicMax(&keys[1], f2key(mx));
    }
}

// out[0] = min, out[1] = max; a NaN first element poisons both, as in the reference loop
__global__ void minmax_finish_kernel(const float* __restrict__ src, const uint32_t* __restrict__ keys,
                                     float* __restrict__ out) {
    const float first = src[0];
    if (first != first) { out[0] = first; out[1] = first; return; }
    out[0] = key2f(keys[0]);
    out[1] = key2f(keys[1]);
}

__global__ __launch_bounds__(kBlock) void normalize_min_max_kernel(const float* __restrict__ src,
                                                                    float* __restrict__ dst, long long n, float lo,
                                                                    float hi, const float* __restrict__ mm) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float min_val = mm[0], max_val = mm[1];
    dst[i] = (src[i] - min_val) * (hi - lo) / (max_val - min_val) + lo;
}

// flips: one thread per pixel, pixel = `pb` bytes
template <bool HORIZONTAL>
__global__ __launch_bounds__(kBlock) void flip_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                      int w, int h, int pb) {
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= (long long)w * h) return;
    const int y = (int)(p / w), x = (int)(p - (long long)y * w);
    const long long q = HORIZONTAL ? (long long)y * w + (w - 1 - x) : (long long)(h - 1 - y) * w + x;
    const uint8_t* s = src + q * pb;
    uint8_t* d = dst + p * pb;
    if ((pb & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0) {
        for (int k = 0; k < pb; k += 4) *reinterpret_cast<uint32_t*>(d + k) = *reinterpret_cast<const uint32_t*>(s + k);
    } else {
        for (int k = 0; k < pb; ++k) d[k] = s[k];
    }
}

int32_t check_n(const void* a, const void* b, int64_t n, const char* what) {
    KH_REQUIRE(n >= 0, KH_ERR_INVALID_ARG, "%s: negative element count", what);
    if (n > 0) KH_REQUIRE(a && b, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    KH_REQUIRE(n <= (int64_t)kI32Max * 64, KH_ERR_TOO_LARGE, "%s: too many elements", what);
    return KH_OK;
}

}  // namespace

extern "C" {

int32_t kh_normalize_mean_std_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels, int32_t channels,
                                  const float* mean, const float* stdv) {
    KH_REQUIRE(channels >= 1 && channels <= 4, KH_ERR_UNSUPPORTED, "kh_normalize_mean_std_f32: %d channels (1..4)", channels);
    KH_REQUIRE(mean && stdv, KH_ERR_INVALID_ARG, "kh_normalize_mean_std_f32: null mean/std");
    const int64_t n = npixels * channels;
    if (int32_t rc = check_n(src, dst, n, "kh_normalize_mean_std_f32")) return rc;
    if (n == 0) return KH_OK;
    Vec4 m{}, s{};
    for (int c = 0; c < channels; ++c) { m.v[c] = mean[c]; s.v[c] = stdv[c]; }
    const dim3 g(cdiv(npixels, kBlock)), b(kBlock);
    hipStream_t st = as_hip(stream);
    if (channels == 3 && (npixels * 3) % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0) {
        const long long nquads = (long long)npixels * 3 / 4;
        hipLaunchKernelGGL(normalize_mean_std_quads3_kernel, dim3(cdiv(nquads, kBlock)), b, 0, st, src, dst, nquads, m, s);
        return check_launch("kh_normalize_mean_std_f32");
    }
    switch (channels) {
        case 1: hipLaunchKernelGGL(normalize_mean_std_kernel<1>, g, b, 0, st, src, dst, (long long)npixels, m, s); break;
        case 2: hipLaunchKernelGGL(normalize_mean_std_kernel<2>, g, b, 0, st, src, dst, (long long)npixels, m, s); break;
        case 3: hipLaunchKernelGGL(normalize_mean_std_kernel<3>, g, b, 0, st, src, dst, (long long)npixels, m, s); break;
        default: hipLaunchKernelGGL(normalize_mean_std_kernel<4>, g, b, 0, st, src, dst, (long long)npixels, m, s); break;
    }
    return check_launch("kh_normalize_mean_std_f32");
}

int32_t kh_normalize_rgb_u8_f32(kh_stream_t stream, const uint8_t* src, float* dst, int64_t npixels, const float* scale,
                                const float* offset) {
    KH_REQUIRE(scale && offset, KH_ERR_INVALID_ARG, "kh_normalize_rgb_u8_f32: null scale/offset");
    if (int32_t rc = check_n(src, dst, npixels, "kh_normalize_rgb_u8_f32")) return rc;
    if (npixels == 0) return KH_OK;
    Vec4 s{{scale[0], scale[1], scale[2], 0.f}}, o{{offset[0], offset[1], offset[2], 0.f}};
    if (npixels % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0) {
        const long long nchunks = (long long)npixels / 4 * 3;
        hipLaunchKernelGGL(normalize_rgb_u8_quads_kernel, dim3(cdiv(nchunks, 3 * kBlock)), dim3(kBlock), 0, as_hip(stream), src, dst, nchunks, s, o);
        return check_launch("kh_normalize_rgb_u8_f32");
    }
    hipLaunchKernelGGL(normalize_rgb_u8_kernel, dim3(cdiv(npixels, kBlock)), dim3(kBlock), 0, as_hip(stream), src, dst,
                       (long long)npixels, s, o);
    return check_launch("kh_normalize_rgb_u8_f32");
}

// minmax_device: 2 floats (min, max) in device memory; scratch_device: 2 uint32 in device memory.
int32_t kh_find_min_max_f32(kh_stream_t stream, const float* src, int64_t n, float* minmax_device,
                            uint32_t* scratch_device) {
    KH_REQUIRE(n > 0, KH_ERR_INVALID_ARG, "image data not initialized: find_min_max of an empty image");
    KH_REQUIRE(src && minmax_device && scratch_device, KH_ERR_INVALID_ARG, "kh_find_min_max_f32: null device pointer");
    hipStream_t st = as_hip(stream);
    KH_HIP(hipMemsetAsync(scratch_device, 0xFF, 4, st));
    KH_HIP(hipMemsetAsync(scratch_device + 1, 0x00, 4, st));
    const unsigned blocks = (unsigned)(n / kBlock + 1 < 2048 ? n / kBlock + 1 : 2048);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(blocks), dim3(kBlock), 0, st, src, (long long)n, scratch_device);
    hipLaunchKernelGGL(minmax_finish_kernel, dim3(1), dim3(1), 0, st, src, scratch_device, minmax_device);
    return check_launch("kh_find_min_max_f32");
}

int32_t kh_normalize_min_max_f32(kh_stream_t stream, const float* src, float* dst, int64_t n, float min, float max,
                                 float* minmax_device, uint32_t* scratch_device) {
    if (int32_t rc = kh_find_min_max_f32(stream, src, n, minmax_device, scratch_device)) return rc;
    KH_REQUIRE(dst, KH_ERR_INVALID_ARG, "kh_normalize_min_max_f32: null device pointer");
    hipLaunchKernelGGL(normalize_min_max_kernel, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, as_hip(stream), src, dst,
                       (long long)n, min, max, minmax_device);
    return check_launch("kh_normalize_min_max_f32");
}

int32_t kh_crop(kh_stream_t stream, const void* src, void* dst, int32_t src_w, int32_t src_h, int32_t dst_w, int32_t dst_h,
                int32_t x, int32_t y, int32_t pixel_bytes) {
    KH_REQUIRE(src_w >= 0 && src_h >= 0 && dst_w >= 0 && dst_h >= 0 && x >= 0 && y >= 0 && pixel_bytes > 0,
               KH_ERR_INVALID_ARG, "kh_crop: negative geometry");
    KH_REQUIRE((int64_t)x + dst_w <= src_w && (int64_t)y + dst_h <= src_h, KH_ERR_INVALID_ARG,
               "pixel index out of bounds: crop (%lld, %lld) exceeds source %dx%d", (long long)x + dst_w, (long long)y + dst_h, src_w, src_h);
    if ((int64_t)dst_w * dst_h == 0) return KH_OK;
    KH_REQUIRE(src && dst, KH_ERR_INVALID_ARG, "kh_crop: null device pointer");
    const char* s = static_cast<const char*>(src) + ((size_t)y * src_w + x) * pixel_bytes;
    KH_HIP(hipMemcpy2DAsync(dst, (size_t)dst_w * pixel_bytes, s, (size_t)src_w * pixel_bytes, (size_t)dst_w * pixel_bytes,
                            (size_t)dst_h, hipMemcpyDeviceToDevice, as_hip(stream)));
    return KH_OK;
}

int32_t kh_flip(kh_stream_t stream, const void* src, void* dst, int32_t width, int32_t height, int32_t pixel_bytes,
                int32_t horizontal) {
    KH_REQUIRE(width >= 0 && height >= 0 && pixel_bytes > 0, KH_ERR_INVALID_ARG, "kh_flip: bad geometry");
    const int64_t n = (int64_t)width * height;
    if (n == 0) return KH_OK;
    KH_REQUIRE(src && dst && src != dst, KH_ERR_INVALID_ARG, "kh_flip: null or aliased device pointers");
    const dim3 g(cdiv(n, kBlock)), b(kBlock);
    if (horizontal)
        hipLaunchKernelGGL(flip_kernel<true>, g, b, 0, as_hip(stream), (const uint8_t*)src, (uint8_t*)dst, width, height, pixel_bytes);
    else
        hipLaunchKernelGGL(flip_kernel<false>, g, b, 0, as_hip(stream), (const uint8_t*)src, (uint8_t*)dst, width, height, pixel_bytes);
    return check_launch("kh_flip");
}

}  // extern "C"
