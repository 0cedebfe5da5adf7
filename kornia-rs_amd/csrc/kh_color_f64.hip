// f64 colour conversions for gfx950 — the device twins of the reference's 18 f64 colour launchers
// (crates/kornia-imgproc/src/color/cuda_dispatch.rs:48-61, 111-135: gray, hsv / hls, the CIE family, YCbCr / YUV).
//
// The per-pixel arithmetic lives in kh_color_f64.h and is the reference's CPU f64 path operation for operation
// (its own CUDA twins multiply by 1/255 where the CPU divides and are held to 1e-3; following the CPU keeps the
// +,-,*,/ conversions bit-identical to it).  HBM-bound maps: one pixel = 24 B in / 8..24 B out per thread,
// consecutive lanes read consecutive pixels (1.5 KiB contiguous per wave instruction).
#include "kh_common.h"

#include "kh_color_f64.h"

using namespace kh;

namespace {

template <int CIN, int COUT>
__global__ __launch_bounds__(kBlock) void map_f64_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                         long long npx, int conv) {
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= npx) return;
    double in[3], out[3];
#pragma unroll
    for (int c = 0; c < CIN; ++c) in[c] = src[p * CIN + c];
    kh_f64::convert_pixel(conv, in, out);  // `conv` is uniform over the launch
#pragma unroll
    for (int c = 0; c < COUT; ++c) dst[p * COUT + c] = out[c];
}

}  // namespace

extern "C" int32_t kh_color_convert_f64(kh_stream_t stream, const double* src, double* dst, int64_t npixels,
                                        int32_t conversion) {
    const char* what = "kh_color_convert_f64";
    KH_REQUIRE(conversion >= 0 && conversion < kh_f64::kCount, KH_ERR_INVALID_ARG, "%s: unknown conversion %d", what, conversion);
    KH_REQUIRE(npixels >= 0, KH_ERR_INVALID_ARG, "%s: negative pixel count", what);
    if (npixels == 0) return KH_OK;
    KH_REQUIRE(src && dst, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    KH_REQUIRE(npixels <= kI32Max * 64, KH_ERR_TOO_LARGE, "%s: %lld pixels exceed the launch limit", what, (long long)npixels);
    const dim3 grid(cdiv(npixels, kBlock)), blk(kBlock);
    hipStream_t st = as_hip(stream);
    if (conversion == kh_f64::kGrayFromRgb)
        hipLaunchKernelGGL((map_f64_kernel<3, 1>), grid, blk, 0, st, src, dst, (long long)npixels, (int)conversion);
    else if (conversion == kh_f64::kRgbFromGray)
        hipLaunchKernelGGL((map_f64_kernel<1, 3>), grid, blk, 0, st, src, dst, (long long)npixels, (int)conversion);
    else
        hipLaunchKernelGGL((map_f64_kernel<3, 3>), grid, blk, 0, st, src, dst, (long long)npixels, (int)conversion);
    return check_launch(what);
}
