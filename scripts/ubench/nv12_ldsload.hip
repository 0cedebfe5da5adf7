// Dev micro-benchmark #9 (not shipped; written at the end of round 1, to be RUN at the start of round 2):
// the read side of the north-star kernel.  Production issues 8 dword-per-lane loads per 1024 pixels (dword
// loads read at only ~3.1 TB/s in isolation, profiles/r01b_ubench_nv12.txt); the staged variants fetch a
// 128 px x 8 row tile with ONE 16-byte and ONE 8-byte load per lane, pass it through a wave-private LDS
// tile, and keep the three-stream NT store pattern (two 512-B segments per store instruction).
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


// production mapping: thread = 4 px of one row, 3 planes
template <bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_4x1(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * a.h;
    const int g = blockIdx.x * BLOCK + threadIdx.x;
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

// staged: a wave owns a 128 px x 8 row tile (w % 128 == 0, h % 8 == 0).  Loads: lane l -> luma row l/8,
// bytes 16*(l%8) (dwordx4) and chroma row l/16, bytes 8*(l%16) (dwordx2).  Compute: lane l -> column group
// g = l % 32 (4 px), rows 4*(l/32) .. +3.  LDS_PAD staggers rows to dodge bank conflicts on the row reads.
template <bool NT, int LDS_PAD>
__global__ __launch_bounds__(256) void k_staged(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    constexpr int YS = 128 + LDS_PAD, TILE = 8 * YS + 4 * YS;  // bytes per wave: 8 luma rows + 4 chroma rows
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][TILE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles_x = a.w >> 7, tiles = tiles_x * (a.h >> 3);
    const int t = blockIdx.x * 4 + wv;
    if (t >= tiles) return;
    const int ty = t / tiles_x, tx = t - ty * tiles_x, w = a.w;
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const long long plane = (long long)w * a.h;
    const int x0 = tx * 128, y0 = ty * 8;
    // wide loads
    const u32x4 yv = *(const u32x4*)(src + (long long)(y0 + (lane >> 3)) * w + x0 + 16 * (lane & 7));
    const uint2 cv = *(const uint2*)(src + plane + (long long)((y0 >> 1) + (lane >> 4)) * w + x0 + 8 * (lane & 15));
    uint8_t* tile = lds[wv];
    *(u32x4*)(tile + (lane >> 3) * YS + 16 * (lane & 7)) = yv;
    *(uint2*)(tile + 8 * YS + (lane >> 4) * YS + 8 * (lane & 15)) = cv;
    __builtin_amdgcn_wave_barrier();
    const int g = lane & 31, rb = 4 * (lane >> 5);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {  // row pair: one chroma row
        const uint32_t uv4 = *(const uint32_t*)(tile + 8 * YS + ((rb >> 1) + pr) * YS + 4 * g);
        int tb[2], tg[2], tr[2];
        chroma_terms(uv4, tb, tg, tr);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = rb + 2 * pr + k;
            const uint32_t y4 = *(const uint32_t*)(tile + row * YS + 4 * g);
            float o[3][4];
            decode_row<1>(y4, tb, tg, tr, a, o);
            const long long off = (long long)(y0 + row) * w + x0 + 4 * g;
#pragma unroll
            for (int c = 0; c < 3; ++c) st4<NT>(dst + c * plane + off, o[c][0], o[c][1], o[c][2], o[c][3]);
        }
    }
}

// control: the staged tile mapping and store order, but with the production per-lane dword loads (no LDS) —
// separates the effect of the 128 x 8 store pattern from the effect of the wide loads
template <bool NT>
__global__ __launch_bounds__(256) void k_tile_direct(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles_x = a.w >> 7, tiles = tiles_x * (a.h >> 3);
    const int t = blockIdx.x * 4 + wv;
    if (t >= tiles) return;
    const int ty = t / tiles_x, tx = t - ty * tiles_x, w = a.w;
    const uint8_t* src = sb + (long long)blockIdx.y * a.sfs;
    float* dst = db + (long long)blockIdx.y * a.dfs;
    const long long plane = (long long)w * a.h;
    const int x0 = tx * 128, y0 = ty * 8, g = lane & 31, rb = 4 * (lane >> 5);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const uint32_t uv4 = *(const uint32_t*)(src + plane + (long long)(((y0 + rb) >> 1) + pr) * w + x0 + 4 * g);
        int tb[2], tg[2], tr[2];
        chroma_terms(uv4, tb, tg, tr);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long long off = (long long)(y0 + rb + 2 * pr + k) * w + x0 + 4 * g;
            const uint32_t y4 = *(const uint32_t*)(src + off);
            float o[3][4];
            decode_row<1>(y4, tb, tg, tr, a, o);
#pragma unroll
            for (int c = 0; c < 3; ++c) st4<NT>(dst + c * plane + off, o[c][0], o[c][1], o[c][2], o[c][3]);
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
    const int g1 = (W / 4) * H;
    const double full = (double)(fb + ob * 4) * N;
    struct V { std::string name; std::function<void(int)> run; std::vector<float> ms; };
    std::vector<V> vs;
    auto nb = [&](int blk) { return (g1 + blk - 1) / blk; };
    const int tiles = (W / 128) * (H / 8), tb4 = (tiles + 3) / 4;
    vs.push_back({"4x1 NT b512 (production)", [&](int n) { hipLaunchKernelGGL((k_4x1<true, 512>), dim3(nb(512), n), dim3(512), 0, st, src, dst, a); }, {}});
    vs.push_back({"4x1 NT b256", [&](int n) { hipLaunchKernelGGL((k_4x1<true, 256>), dim3(nb(256), n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"staged 128x8 NT pad0", [&](int n) { hipLaunchKernelGGL((k_staged<true, 0>), dim3(tb4, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"staged 128x8 NT pad16", [&](int n) { hipLaunchKernelGGL((k_staged<true, 16>), dim3(tb4, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"staged 128x8 st pad16", [&](int n) { hipLaunchKernelGGL((k_staged<false, 16>), dim3(tb4, n), dim3(256), 0, st, src, dst, a); }, {}});
    vs.push_back({"tile 128x8 direct loads NT", [&](int n) { hipLaunchKernelGGL((k_tile_direct<true>), dim3(tb4, n), dim3(256), 0, st, src, dst, a); }, {}});
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
