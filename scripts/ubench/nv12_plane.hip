// Dev micro-benchmark #4 (not shipped): plane-major block order (one store stream at a time, source
// re-read from L2/Infinity Cache) vs the 3-stream 4x1 mapping; 24-bit multiplies.
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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kCY = 1220542, kCUB = 2116026, kCUG = -409993, kCVG = -852492, kCVR = 1673527, kHalf20 = 1 << 19;
__device__ __forceinline__ int clamp255(int v) { return min(max(v, 0), 255); }
struct Args { int w, h; float m0, m1, m2, is0, is1, is2; long long sfs, dfs; };
template <bool NT> __device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    if constexpr (NT) __builtin_nontemporal_store(v, (f32x4*)p); else *(f32x4*)p = v;
}
extern __shared__ __attribute__((aligned(16))) char dyn_lds[];

template <int DIV>
__device__ __forceinline__ float norm1(int v, float m, float is) {
    const float x = (float)v;
    if constexpr (DIV == 1) {
        const float rc = 1.0f / 255.0f;
        float q = x * rc, r = __builtin_fmaf(-q, 255.0f, x);
        q = __builtin_fmaf(r, rc, q);
        return (q - m) * is;
    } else {
        return (x / 255.0f - m) * is;
    }
}

template <int DIV>
__device__ __forceinline__ void decode_row(uint32_t y4, const int tb[2], const int tg[2], const int tr[2], const Args& a, float o[3][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = max((int)((y4 >> (8 * j)) & 0xFFu) - 16, 0) * kCY;
        const int k = j >> 1;
        o[0][j] = norm1<DIV>(clamp255((yy + tr[k]) >> 20), a.m0, a.is0);
        o[1][j] = norm1<DIV>(clamp255((yy + tg[k]) >> 20), a.m1, a.is1);
        o[2][j] = norm1<DIV>(clamp255((yy + tb[k]) >> 20), a.m2, a.is2);
    }
}
__device__ __forceinline__ void chroma_terms(uint32_t uv4, int tb[2], int tg[2], int tr[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = (int)((uv4 >> (16 * k)) & 0xFFu) - 128, v = (int)((uv4 >> (16 * k + 8)) & 0xFFu) - 128;
        tb[k] = kCUB * u + kHalf20; tg[k] = kCUG * u + kCVG * v + kHalf20; tr[k] = kCVR * v + kHalf20;
    }
}


template <int MUL24>
__device__ __forceinline__ int mulc(int c, int v) { if constexpr (MUL24) return __mul24(c, v); else return c * v; }

template <int DIV, int MUL24>
__device__ __forceinline__ float chan(int c, int y, int u, int v, const Args& a) {
    const int yy = mulc<MUL24>(max(y - 16, 0), kCY);
    int t; float m, is;
    if (c == 0) { t = mulc<MUL24>(kCVR, v) + kHalf20; m = a.m0; is = a.is0; }
    else if (c == 1) { t = mulc<MUL24>(kCUG, u) + mulc<MUL24>(kCVG, v) + kHalf20; m = a.m1; is = a.is1; }
    else { t = mulc<MUL24>(kCUB, u) + kHalf20; m = a.m2; is = a.is2; }
    return norm1<DIV>(clamp255((yy + t) >> 20), m, is);
}

// 3-stream 4x1 reference (all channels per thread)
template <bool NT, int DIV, int MUL24>
__global__ __launch_bounds__(256) void k_4x1(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * a.h;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const int r = g / wq, xq = g - r * wq, w = a.w;
    const long long plane = (long long)w * a.h;
    const uint32_t y4 = *(const uint32_t*)(src + (long long)r * w + 4 * xq);
    const uint32_t uv4 = *(const uint32_t*)(src + plane + (long long)(r >> 1) * w + 4 * xq);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = chan<DIV, MUL24>(c, (y4 >> (8 * j)) & 0xFF, (int)((uv4 >> (16 * (j >> 1))) & 0xFF) - 128, (int)((uv4 >> (16 * (j >> 1) + 8)) & 0xFF) - 128, a);
        st4<NT>(dst + c * plane + (long long)r * w + 4 * xq, o[0], o[1], o[2], o[3]);
    }
}

// plane-major: block b -> plane c = b / chunks, 1024-px chunk k = b % chunks; 4 px per thread, ONE channel
template <bool NT, int DIV, int MUL24>
__global__ __launch_bounds__(256) void k_plane41(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int w = a.w;
    const long long plane = (long long)w * a.h;
    const int chunks = (int)((plane + 1023) >> 10);
    const int c = blockIdx.x / chunks, k = blockIdx.x - c * chunks;
    const long long p = (long long)k * 1024 + 4 * threadIdx.x;
    if (p >= plane) return;
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const int r = (int)(p / w), x = (int)(p - (long long)r * w);
    const uint32_t y4 = *(const uint32_t*)(src + p);
    const uint32_t uv4 = *(const uint32_t*)(src + plane + (long long)(r >> 1) * w + x);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        o[j] = chan<DIV, MUL24>(c, (y4 >> (8 * j)) & 0xFF, (int)((uv4 >> (16 * (j >> 1))) & 0xFF) - 128, (int)((uv4 >> (16 * (j >> 1) + 8)) & 0xFF) - 128, a);
    st4<NT>(dst + c * plane + p, o[0], o[1], o[2], o[3]);
}

// plane-major, 16 px per thread via one dwordx4 Y load + one dwordx4 UV load, 4 float4 stores
// (lane stride 64 B: each store instruction covers every 4th 16 B of a 4 KiB span)
template <bool NT, int DIV, int MUL24>
__global__ __launch_bounds__(256) void k_plane16(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int w = a.w;
    const long long plane = (long long)w * a.h;
    const int chunks = (int)((plane + 4095) >> 12);
    const int c = blockIdx.x / chunks, k = blockIdx.x - c * chunks;
    const long long p = (long long)k * 4096 + 16 * threadIdx.x;
    if (p >= plane) return;
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const int r = (int)(p / w), x = (int)(p - (long long)r * w);
    const u32x4 yv = *(const u32x4*)(src + p);
    const u32x4 uvv = *(const u32x4*)(src + plane + (long long)(r >> 1) * w + x);
    const uint32_t yw[4] = {yv.x, yv.y, yv.z, yv.w}, uw[4] = {uvv.x, uvv.y, uvv.z, uvv.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = chan<DIV, MUL24>(c, (yw[q] >> (8 * j)) & 0xFF, (int)((uw[q] >> (16 * (j >> 1))) & 0xFF) - 128, (int)((uw[q] >> (16 * (j >> 1) + 8)) & 0xFF) - 128, a);
        st4<NT>(dst + c * plane + p + 4 * q, o[0], o[1], o[2], o[3]);
    }
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
    Args a{W, H, 0.485f, 0.456f, 0.406f, 1.0f / 0.229f, 1.0f / 0.224f, 1.0f / 0.225f, (long long)fb, (long long)ob};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int g1 = (W / 4) * H;
    const int ch1 = (W * H + 1023) / 1024, ch16 = (W * H + 4095) / 4096;
    const double full = (double)(fb + ob * 4) * N;
    struct V { std::string name; std::function<void(int)> run; std::vector<float> ms; };
    std::vector<V> vs;
    auto G = [&](int groups, int n) { return dim3((groups + 255) / 256, n); };
    vs.push_back({"4x1 st div  mul32 (3 streams)", [&](int n) { hipLaunchKernelGGL((k_4x1<false, 0, 0>), G(g1, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 st rcpfma mul32", [&](int n) { hipLaunchKernelGGL((k_4x1<false, 1, 0>), G(g1, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 st rcpfma mul24", [&](int n) { hipLaunchKernelGGL((k_4x1<false, 1, 1>), G(g1, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 NT rcpfma mul24", [&](int n) { hipLaunchKernelGGL((k_4x1<true, 1, 1>), G(g1, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"plane-major 4px st div mul24", [&](int n) { hipLaunchKernelGGL((k_plane41<false, 0, 1>), dim3(3 * ch1, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"plane-major 4px st rcpfma mul24", [&](int n) { hipLaunchKernelGGL((k_plane41<false, 1, 1>), dim3(3 * ch1, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"plane-major 4px NT rcpfma mul24", [&](int n) { hipLaunchKernelGGL((k_plane41<true, 1, 1>), dim3(3 * ch1, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"plane-major 16px st rcpfma mul24", [&](int n) { hipLaunchKernelGGL((k_plane16<false, 1, 1>), dim3(3 * ch16, n), dim3(256), 0, st, src, dst, a); }, {}});
    for (int lds : {40, 80}) {
        int bytes = lds * 1024 - 256;
        vs.push_back({"plane-major 4px st rcpfma mul24 lds=" + std::to_string(lds) + "K", [&, bytes](int n) { hipLaunchKernelGGL((k_plane41<false, 1, 1>), dim3(3 * ch1, n), dim3(256), bytes, st, src, dst, a); }, {}});
    }
    CK(hipFuncSetAttribute((const void*)k_plane41<false, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));

    std::vector<float> want(ob * 2), got(ob * 2);
    vs[0].run(2);
    CK(hipMemcpyAsync(want.data(), dst, ob * 2 * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    for (auto& v : vs) {
        CK(hipMemsetAsync(dst, 0xFF, ob * 2 * 4, st));
        v.run(2); CK(hipGetLastError());
        CK(hipMemcpyAsync(got.data(), dst, ob * 2 * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        size_t bad = 0;
        for (size_t i = 0; i < want.size(); ++i) bad += (*(uint32_t*)&want[i] != *(uint32_t*)&got[i]);
        if (bad) printf("MISMATCH %-36s %zu elements\n", v.name.c_str(), bad);
    }
    for (int r = 0; r < ROUNDS + 1; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, st)); v.run(N); CK(hipGetLastError()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    printf("%-44s %9s %9s %9s\n", "variant (N frames of 1080p)", "med ms", "min ms", "GB/s@med");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2];
        printf("%-44s %9.3f %9.3f %9.0f\n", v.name.c_str(), med, v.ms[0], full / med / 1e6);
    }
    return 0;
}
