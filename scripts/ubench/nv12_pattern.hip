// Dev micro-benchmark #2 (not shipped): which part of the NV12->CHW access pattern costs the gap
// between the 6-stream store pattern (~5.2 TB/s) and a flat fill (~6.9 TB/s)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Args { int w, h; long long sfs, dfs; };
template <bool NT> __device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    if constexpr (NT) __builtin_nontemporal_store(v, (f32x4*)p); else *(f32x4*)p = v;
}
extern __shared__ char dyn_lds[];

// 4 px x ROWS rows per thread; READS: load the NV12 bytes or not.
template <bool NT, bool READS, int ROWS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_shape(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * (a.h / ROWS);
    const int g = blockIdx.x * BLOCK + threadIdx.x;
    if (g >= groups) return;
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const int rp = g / wq, xq = g - rp * wq, w = a.w;
    const long long plane = (long long)w * a.h;
    float f[3] = {1.0f, 2.0f, 3.0f};
    if constexpr (READS) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) f[r] = __uint_as_float(*(const uint32_t*)(src + (long long)(ROWS * rp + r) * w + 4 * xq));
        f[2] = __uint_as_float(*(const uint32_t*)(src + plane + (long long)((ROWS * rp) >> 1) * w + 4 * xq));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            st4<NT>(dst + c * plane + (long long)(ROWS * rp + r) * w + 4 * xq, f[0], f[1], f[2], f[c]);
}

// one block = one full row pair (w/4 quads, BLOCK >= w/4 threads), so each plane gets 2*w*4 contiguous bytes per block
template <bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_rowpair(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, xq = threadIdx.x, rp = blockIdx.x, w = a.w;
    if (xq >= wq) return;
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const long long plane = (long long)w * a.h;
    float f0 = __uint_as_float(*(const uint32_t*)(src + (long long)(2 * rp) * w + 4 * xq));
    float f1 = __uint_as_float(*(const uint32_t*)(src + (long long)(2 * rp + 1) * w + 4 * xq));
    float f2 = __uint_as_float(*(const uint32_t*)(src + plane + (long long)rp * w + 4 * xq));
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r)
            st4<NT>(dst + c * plane + (long long)(2 * rp + r) * w + 4 * xq, f0, f1, f2, f0);
}

template <bool NT>
__global__ __launch_bounds__(256) void k_fill_flat(float* __restrict__ db, long long n4, float v) {
    long long i = (long long)blockIdx.y * gridDim.x * 256 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) st4<NT>(db + 4 * i, v, v, v, v);
}
// flat fill, but consecutive blocks rotate over 3 "planes" of each frame (3 streams, same bytes)
template <bool NT>
__global__ __launch_bounds__(256) void k_fill_3stream(float* __restrict__ db, Args a) {
    const long long plane = (long long)a.w * a.h;        // floats
    const int chunks = (int)(plane / 1024);              // 4 KiB chunks per plane (1080p: 2025)
    const int b = blockIdx.x;                            // 0 .. 3*chunks-1
    const int c = b % 3, k = b / 3;
    if (k >= chunks) return;
    float* dst = db + (long long)blockIdx.y * a.dfs + c * plane + (long long)k * 1024 + 4 * threadIdx.x;
    st4<NT>(dst, 1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void k_read(const uint8_t* __restrict__ sb, float* __restrict__ out, long long n4) {
    long long i = (long long)blockIdx.y * gridDim.x * 256 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    uint32_t v = *(const uint32_t*)(sb + 4 * i);
    if (v == 0x12345677u) out[0] = 1.0f;  // practically never
}

int main(int argc, char** argv) {
    const int W = 1920, H = 1080, N = argc > 1 ? atoi(argv[1]) : 1024, ROUNDS = 5;
    const size_t fb = (size_t)W * H * 3 / 2, ob = (size_t)W * H * 3;
    uint8_t* src; float* dst;
    CK(hipMalloc(&src, fb * N)); CK(hipMalloc(&dst, ob * N * 4));
    {
        std::vector<uint8_t> h(fb + 31 * 64); uint32_t s = 0x12345678u;
        for (auto& b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
        for (int k = 0; k < N; ++k) CK(hipMemcpy(src + k * fb, h.data() + 31 * (k % 64), fb, hipMemcpyHostToDevice));
    }
    Args a{W, H, (long long)fb, (long long)ob};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int g2 = (W / 4) * (H / 2), g1 = (W / 4) * H;
    const long long n4 = (long long)ob * N / 4, r4 = (long long)fb * N / 4;
    const double full = (double)(fb + ob * 4) * N, wonly = (double)ob * 4 * N, ronly = (double)fb * N;
    struct V { std::string name; double bytes; std::function<void()> run; std::vector<float> ms; };
    std::vector<V> vs;
    auto G = [&](int groups, int blk) { return dim3((groups + blk - 1) / blk, N); };
    vs.push_back({"W-only 4x2 (6 streams) NT", wonly, [&] { hipLaunchKernelGGL((k_shape<true, false, 2, 256>), G(g2, 256), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"W-only 4x2 (6 streams) st", wonly, [&] { hipLaunchKernelGGL((k_shape<false, false, 2, 256>), G(g2, 256), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"W-only 4x1 (3 streams) st", wonly, [&] { hipLaunchKernelGGL((k_shape<false, false, 1, 256>), G(g1, 256), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"R+W 4x2 st", full, [&] { hipLaunchKernelGGL((k_shape<false, true, 2, 256>), G(g2, 256), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"R+W 4x1 st (UV re-read)", full, [&] { hipLaunchKernelGGL((k_shape<false, true, 1, 256>), G(g1, 256), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"R+W 4x1 NT (UV re-read)", full, [&] { hipLaunchKernelGGL((k_shape<true, true, 1, 256>), G(g1, 256), dim3(256), 0, st, src, dst, a); }, {}});
    for (int lds : {20, 40, 80, 160}) {
        int bytes = lds * 1024 - 256;
        vs.push_back({"R+W 4x2 st  lds=" + std::to_string(lds) + "K (" + std::to_string(160 / lds) + " blk/CU)", full,
                      [&, bytes] { hipLaunchKernelGGL((k_shape<false, true, 2, 256>), G(g2, 256), dim3(256), bytes, st, src, dst, a); }, {}});
    }
    for (int lds : {40, 80, 160}) {
        int bytes = lds * 1024 - 256;
        vs.push_back({"R+W 4x2 st b1024 lds=" + std::to_string(lds) + "K", full,
                      [&, bytes] { hipLaunchKernelGGL((k_shape<false, true, 2, 1024>), G(g2, 1024), dim3(1024), bytes, st, src, dst, a); }, {}});
    }
    vs.push_back({"R+W rowpair/block b512 st", full, [&] { hipLaunchKernelGGL((k_rowpair<false, 512>), dim3(H / 2, N), dim3(512), 0, st, src, dst, a); }, {}});
    vs.push_back({"R+W rowpair/block b512 NT", full, [&] { hipLaunchKernelGGL((k_rowpair<true, 512>), dim3(H / 2, N), dim3(512), 0, st, src, dst, a); }, {}});
    vs.push_back({"fill flat st", wonly, [&] { hipLaunchKernelGGL((k_fill_flat<false>), dim3(65536, (unsigned)((n4 + 65536LL * 256 - 1) / (65536LL * 256))), dim3(256), 0, st, dst, n4, 1.5f); }, {}});
    vs.push_back({"fill 3-stream rotate st", wonly, [&] { hipLaunchKernelGGL((k_fill_3stream<false>), dim3(3 * 2025, N), dim3(256), 0, st, dst, a); }, {}});
    for (int lds : {40, 80, 160}) {
        int bytes = lds * 1024 - 256;
        vs.push_back({"fill flat st lds=" + std::to_string(lds) + "K", wonly,
                      [&, bytes] { hipLaunchKernelGGL((k_fill_flat<false>), dim3(65536, (unsigned)((n4 + 65536LL * 256 - 1) / (65536LL * 256))), dim3(256), bytes, st, dst, n4, 1.5f); }, {}});
    }
    vs.push_back({"read-only src (dword/lane)", ronly, [&] { hipLaunchKernelGGL(k_read, dim3(65536, (unsigned)((r4 + 65536LL * 256 - 1) / (65536LL * 256))), dim3(256), 0, st, src, dst, r4); }, {}});

    CK(hipFuncSetAttribute((const void*)k_shape<false, true, 2, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    CK(hipFuncSetAttribute((const void*)k_shape<false, true, 2, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    CK(hipFuncSetAttribute((const void*)k_fill_flat<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    for (int r = 0; r < ROUNDS + 1; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, st)); v.run(); CK(hipGetLastError()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    printf("%-40s %9s %9s %9s\n", "variant (N frames of 1080p)", "med ms", "min ms", "GB/s@med");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2];
        printf("%-40s %9.3f %9.3f %9.0f\n", v.name.c_str(), med, v.ms[0], v.bytes / med / 1e6);
    }
    return 0;
}
