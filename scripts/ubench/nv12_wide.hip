// Dev micro-benchmark #6 (not shipped): 1 KiB-contiguous dwordx4 source loads (a wave owns 1024 px x 2
// rows), transposed through LDS so that every store instruction is still 1 KiB contiguous.
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


template <bool NT, int DIV>
__global__ __launch_bounds__(256) void k_base(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * (a.h >> 1);
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const int rp = g / wq, xq = g - rp * wq, w = a.w;
    const long long plane = (long long)w * a.h;
    const uint32_t yt = *(const uint32_t*)(src + (long long)(2 * rp) * w + 4 * xq);
    const uint32_t yb = *(const uint32_t*)(src + (long long)(2 * rp + 1) * w + 4 * xq);
    const uint32_t uv4 = *(const uint32_t*)(src + plane + (long long)rp * w + 4 * xq);
    int tb[2], tg[2], tr[2];
    chroma_terms(uv4, tb, tg, tr);
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        float o[3][4];
        decode_row<DIV>(row ? yb : yt, tb, tg, tr, a, o);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            st4<NT>(dst + c * plane + (long long)(2 * rp + row) * w + 4 * xq, o[c][0], o[c][1], o[c][2], o[c][3]);
    }
}

// wave = 1024 px x 2 rows of one row pair.  Lane L loads 16 B at px 16L of Y-top, Y-bottom and UV
// (three 1 KiB-contiguous wave loads), parks them in the wave's LDS slab, then reads back the
// dword at px 256k + 4L for k = 0..3, so store instruction (c,row,k) writes px [256k, 256k+256).
// ROWS1: 4x1-like ordering is not applicable here; COMPUTE=false stores raw bits (traffic only).
template <bool NT, int DIV, bool COMPUTE>
__global__ __launch_bounds__(256) void k_wide(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    uint32_t* lds = (uint32_t*)dyn_lds;                  // [4 waves][3][256] dwords = 12 KiB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wpr = (a.w + 1023) >> 10;                  // waves per row pair
    const int gw = blockIdx.x * 4 + wv;
    const int rp = gw / wpr, seg = gw - rp * wpr;
    if (rp >= (a.h >> 1)) return;                        // whole wave exits; no block barrier used below
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const int w = a.w, x0 = seg << 10;
    const long long plane = (long long)w * a.h;
    uint32_t* slab = lds + wv * 768;
    const int xl = x0 + 16 * lane;
    if (xl < w) {                                        // w % 16 == 0 for this variant
        const u32x4 t = *(const u32x4*)(src + (long long)(2 * rp) * w + xl);
        const u32x4 b = *(const u32x4*)(src + (long long)(2 * rp + 1) * w + xl);
        const u32x4 u = *(const u32x4*)(src + plane + (long long)rp * w + xl);
        *(u32x4*)(slab + 4 * lane) = t;
        *(u32x4*)(slab + 256 + 4 * lane) = b;
        *(u32x4*)(slab + 512 + 4 * lane) = u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + 256 * k + 4 * lane;
        if (x >= w) break;
        const uint32_t yt = slab[64 * k + lane], yb = slab[256 + 64 * k + lane], uv4 = slab[512 + 64 * k + lane];
        if constexpr (COMPUTE) {
            int tb[2], tg[2], tr[2];
            chroma_terms(uv4, tb, tg, tr);
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                float o[3][4];
                decode_row<DIV>(row ? yb : yt, tb, tg, tr, a, o);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    st4<NT>(dst + c * plane + (long long)(2 * rp + row) * w + x, o[c][0], o[c][1], o[c][2], o[c][3]);
            }
        } else {
            const float f0 = __uint_as_float(yt), f1 = __uint_as_float(yb), f2 = __uint_as_float(uv4);
#pragma unroll
            for (int row = 0; row < 2; ++row)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    st4<NT>(dst + c * plane + (long long)(2 * rp + row) * w + x, f0, f1, f2, f0);
        }
    }
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
    const int g2 = (W / 4) * (H / 2);
    const int nw = ((W + 1023) / 1024) * (H / 2);
    const double full = (double)(fb + ob * 4) * N;
    struct V { std::string name; bool check; std::function<void(int)> run; std::vector<float> ms; };
    std::vector<V> vs;
    auto G = [&](int groups, int n) { return dim3((groups + 255) / 256, n); };
    vs.push_back({"base 4x2 st div", true, [&](int n) { hipLaunchKernelGGL((k_base<false, 0>), G(g2, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"base 4x2 st rcpfma", true, [&](int n) { hipLaunchKernelGGL((k_base<false, 1>), G(g2, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"base 4x2 NT rcpfma", true, [&](int n) { hipLaunchKernelGGL((k_base<true, 1>), G(g2, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"wide16 st div", true, [&](int n) { hipLaunchKernelGGL((k_wide<false, 0, true>), dim3((nw + 3) / 4, n), dim3(256), 12288, st, src, dst, a); }, {}});
    vs.push_back({"wide16 st rcpfma", true, [&](int n) { hipLaunchKernelGGL((k_wide<false, 1, true>), dim3((nw + 3) / 4, n), dim3(256), 12288, st, src, dst, a); }, {}});
    vs.push_back({"wide16 NT rcpfma", true, [&](int n) { hipLaunchKernelGGL((k_wide<true, 1, true>), dim3((nw + 3) / 4, n), dim3(256), 12288, st, src, dst, a); }, {}});
    vs.push_back({"wide16 st traffic-only (no decode)", false, [&](int n) { hipLaunchKernelGGL((k_wide<false, 1, false>), dim3((nw + 3) / 4, n), dim3(256), 12288, st, src, dst, a); }, {}});
    vs.push_back({"wide16 NT traffic-only (no decode)", false, [&](int n) { hipLaunchKernelGGL((k_wide<true, 1, false>), dim3((nw + 3) / 4, n), dim3(256), 12288, st, src, dst, a); }, {}});

    std::vector<float> want(ob * 2), got(ob * 2);
    vs[0].run(2);
    CK(hipMemcpyAsync(want.data(), dst, ob * 2 * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    for (auto& v : vs) {
        if (!v.check) continue;
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
