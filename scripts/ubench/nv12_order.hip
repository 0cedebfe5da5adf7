// Dev micro-benchmark #8 (not shipped): production 4x1 kernel vs (a) plane-rotating blocks (one store stream
// per block, source re-read from cache), (b) bigger blocks, (c) XCD-contiguous chunk order.
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


template <int DIV>
__device__ __forceinline__ float one(int c, int y, int tb, int tg, int tr, const Args& a) {
    const int yy = max(y - 16, 0) * kCY;
    if (c == 0) return norm1<DIV>(clamp255((yy + tr) >> 20), a.m0, a.is0);
    if (c == 1) return norm1<DIV>(clamp255((yy + tg) >> 20), a.m1, a.is1);
    return norm1<DIV>(clamp255((yy + tb) >> 20), a.m2, a.is2);
}

// production mapping: thread = 4 px of one row, 3 planes.  ORDER: 0 = linear, 1 = XCD-contiguous
template <bool NT, int BLOCK, int ORDER>
__global__ __launch_bounds__(BLOCK) void k_4x1(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * a.h;
    int blk = blockIdx.x;
    if constexpr (ORDER == 1) {
        // dispatch id b runs on XCD b % 8; give XCD x the x-th contiguous eighth of the frame's chunks
        const int nb = gridDim.x, per = (nb + 7) >> 3;
        blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (blk >= nb) return;
    }
    const int g = blk * BLOCK + threadIdx.x;
    if (g >= groups) return;
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const int r = g / wq, xq = g - r * wq, w = a.w;
    const long long plane = (long long)w * a.h, off = (long long)r * w + 4 * xq;
    const uint32_t y4 = *(const uint32_t*)(src + off);
    const uint32_t uv4 = *(const uint32_t*)(src + plane + (long long)(r >> 1) * w + 4 * xq);
    int tb[2], tg[2], tr[2];
    chroma_terms(uv4, tb, tg, tr);
    float o[3][4];
    decode_row<1>(y4, tb, tg, tr, a, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) st4<NT>(dst + c * plane + off, o[c][0], o[c][1], o[c][2], o[c][3]);
}

// plane-rotating: dispatch id b -> chunk b / 3, plane b % 3; thread = 4 px, ONE channel, one 16 B store
template <bool NT>
__global__ __launch_bounds__(256) void k_rot(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * a.h;
    const int c = blockIdx.x % 3, chunk = blockIdx.x / 3;
    const int g = chunk * 256 + threadIdx.x;
    if (g >= groups) return;
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const int r = g / wq, xq = g - r * wq, w = a.w;
    const long long plane = (long long)w * a.h, off = (long long)r * w + 4 * xq;
    const uint32_t y4 = *(const uint32_t*)(src + off);
    const uint32_t uv4 = *(const uint32_t*)(src + plane + (long long)(r >> 1) * w + 4 * xq);
    int tb[2], tg[2], tr[2];
    chroma_terms(uv4, tb, tg, tr);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = one<1>(c, (y4 >> (8 * j)) & 0xFF, tb[j >> 1], tg[j >> 1], tr[j >> 1], a);
    st4<NT>(dst + c * plane + off, o[0], o[1], o[2], o[3]);
}

int main(int argc, char** argv) {
    const int W = 1920, H = 1080, N = argc > 1 ? atoi(argv[1]) : 1024, ROUNDS = 7;
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
    const double full = (double)(fb + ob * 4) * N;
    struct V { std::string name; std::function<void(int)> run; std::vector<float> ms; };
    std::vector<V> vs;
    auto nb = [&](int blk) { return (g1 + blk - 1) / blk; };
    vs.push_back({"4x1 NT b256 (production)", [&](int n) { hipLaunchKernelGGL((k_4x1<true, 256, 0>), dim3(nb(256), n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 st b256", [&](int n) { hipLaunchKernelGGL((k_4x1<false, 256, 0>), dim3(nb(256), n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 NT b512", [&](int n) { hipLaunchKernelGGL((k_4x1<true, 512, 0>), dim3(nb(512), n), dim3(512), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 NT b1024", [&](int n) { hipLaunchKernelGGL((k_4x1<true, 1024, 0>), dim3(nb(1024), n), dim3(1024), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 NT b128", [&](int n) { hipLaunchKernelGGL((k_4x1<true, 128, 0>), dim3(nb(128), n), dim3(128), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 NT b64", [&](int n) { hipLaunchKernelGGL((k_4x1<true, 64, 0>), dim3(nb(64), n), dim3(64), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 NT b256 XCD-contiguous", [&](int n) { hipLaunchKernelGGL((k_4x1<true, 256, 1>), dim3(nb(256), n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 st b256 XCD-contiguous", [&](int n) { hipLaunchKernelGGL((k_4x1<false, 256, 1>), dim3(nb(256), n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"plane-rotate NT (1 store/thread)", [&](int n) { hipLaunchKernelGGL((k_rot<true>), dim3(3 * nb(256), n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"plane-rotate st (1 store/thread)", [&](int n) { hipLaunchKernelGGL((k_rot<false>), dim3(3 * nb(256), n), dim3(256), 0, st, src, dst, a); }, {}});

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
