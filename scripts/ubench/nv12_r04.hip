// Dev micro-benchmark (round 4, not shipped).  profiles/r04b_store_grid.txt: with no loads at all, a wave that issues ONE 16-byte
// store runs at 3.52-3.59 ms per 25.5 GB (flat OR three planes), two stores 4.0 ms, three 4.15-4.35 ms — whatever the layout, the
// block size, the waits between the stores.  The production kernel issues three per wave.  Variants here give every wave exactly one
// store without tripling the loads:
//   lds<Q>    block = 3Q threads (Q = 64 / 128 / 256).  Threads 0..Q-1 load + decode quad gbase + t (all three planes) into LDS
//             ([3][Q] x 16 B), barrier, then thread (c, i) stores plane c of quad gbase + i: 12 / 6 / 3 waves, one store each.
//   ldsall<Q> the same, but the decode is spread over all 3Q threads: thread (c, i) loads quad i and decodes only channel c — no
//             LDS, no barrier, three times the load instructions (the r02 `plane1` shape, here with buffer loads / WT stores).
//   ldsx<Q>   lds<Q> with the decode threads taken from all waves (every third lane decodes... no: the first Q/3 lanes of... ) — not built.
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
constexpr int AUX = 19;  // sc0 sc1 nt
__device__ __forceinline__ int clamp255(int v) { return min(max(v, 0), 255); }
struct Args { int w, h; float m0, m1, m2, is0, is1, is2; long long sfs, dfs; };

__device__ __forceinline__ float norm1(int v, float m, float is) {
    const float x = (float)v, rc = 1.0f / 255.0f;
    float q = x * rc, r = __builtin_fmaf(-q, 255.0f, x);
    q = __builtin_fmaf(r, rc, q);
    return (q - m) * is;
}
__device__ __forceinline__ void decode4(uint32_t y4, uint32_t uv4, const Args& a, f32x4 o[3]) {
    int tb[2], tg[2], tr[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = (int)((uv4 >> (16 * k)) & 0xFFu) - 128, v = (int)((uv4 >> (16 * k + 8)) & 0xFFu) - 128;
        tb[k] = kCUB * u + kHalf20; tg[k] = kCUG * u + kCVG * v + kHalf20; tr[k] = kCVR * v + kHalf20;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = max((int)((y4 >> (8 * j)) & 0xFFu) - 16, 0) * kCY;
        const int k = j >> 1;
        o[0][j] = norm1(clamp255((yy + tr[k]) >> 20), a.m0, a.is0);
        o[1][j] = norm1(clamp255((yy + tg[k]) >> 20), a.m1, a.is1);
        o[2][j] = norm1(clamp255((yy + tb[k]) >> 20), a.m2, a.is2);
    }
}
#define RSRC_SRC(a) __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sb + (long long)blockIdx.y * a.sfs), 0, plane + plane / 2, 0x00020000)
#define RSRC_DST(a) __builtin_amdgcn_make_buffer_rsrc(db + (long long)blockIdx.y * a.dfs, 0, 12 * plane, 0x00020000)
constexpr int kDrop = 0x7fffffff;  // + 2 planes still wraps past num_records as an unsigned compare? no: keep below 2^31 (see off below)

// ---- base / bar / sgb: production mapping (thread = quad g of the frame, linear)
template <int BLOCK, int MODE>  // MODE 0 base, 1 barrier before the stores, 2 sched_group_barrier schedule
__global__ __launch_bounds__(BLOCK) void k_lin(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * a.h, plane = a.w * a.h;
    const __amdgpu_buffer_rsrc_t rl = RSRC_SRC(a);
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST(a);
    const int g0 = blockIdx.x * BLOCK + threadIdx.x, g = min(g0, groups - 1);
    const int r = g / wq, xq = g - r * wq;
    const uint32_t y4 = __builtin_amdgcn_raw_buffer_load_b32(rl, r * a.w + 4 * xq, 0, 0);
    const uint32_t uv4 = __builtin_amdgcn_raw_buffer_load_b32(rl, plane + (r >> 1) * a.w + 4 * xq, 0, 0);
    const int off = g0 < groups ? 16 * g : kDrop - 8 * plane;
    f32x4 o[3];
    decode4(y4, uv4, a, o);
    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]));
    if constexpr (MODE == 1) __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[c]), rs, off + c * (4 * plane), 0, AUX);
    if constexpr (MODE == 2) {
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // 2 VMEM reads first
        __builtin_amdgcn_sched_group_barrier(0x002, 200, 0); // then all VALU
        __builtin_amdgcn_sched_group_barrier(0x040, 3, 0);   // then the 3 VMEM writes
    }
}

// ---- lds<Q>: one store per wave, decode by the first Q threads, hand-over through LDS
template <int Q, int MODE>  // MODE 0: decode by threads 0..Q-1;  1: every thread loads, thread (c, i) decodes channel c only (no LDS)
__global__ __launch_bounds__(3 * Q) void k_split(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * a.h, plane = a.w * a.h;
    const __amdgpu_buffer_rsrc_t rl = RSRC_SRC(a);
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST(a);
    const int t = threadIdx.x, gbase = blockIdx.x * Q;
    const int c = t / Q, i = t - c * Q;   // wave-uniform c (Q is a multiple of 64)
    f32x4 v;
    if constexpr (MODE == 0) {
        __shared__ f32x4 tile[3][Q];
        if (t < Q) {
            const int g = min(gbase + t, groups - 1);
            const int r = g / wq, xq = g - r * wq;
            const uint32_t y4 = __builtin_amdgcn_raw_buffer_load_b32(rl, r * a.w + 4 * xq, 0, 0);
            const uint32_t uv4 = __builtin_amdgcn_raw_buffer_load_b32(rl, plane + (r >> 1) * a.w + 4 * xq, 0, 0);
            f32x4 o[3];
            decode4(y4, uv4, a, o);
            tile[0][t] = o[0]; tile[1][t] = o[1]; tile[2][t] = o[2];
        }
        __syncthreads();
        v = tile[c][i];
    } else {
        const int g = min(gbase + i, groups - 1);
        const int r = g / wq, xq = g - r * wq;
        const uint32_t y4 = __builtin_amdgcn_raw_buffer_load_b32(rl, r * a.w + 4 * xq, 0, 0);
        const uint32_t uv4 = __builtin_amdgcn_raw_buffer_load_b32(rl, plane + (r >> 1) * a.w + 4 * xq, 0, 0);
        f32x4 o[3];
        decode4(y4, uv4, a, o);   // WRONG as measured in round 4: `c` is not known wave-uniform, so all three channels are decoded and then selected (159 VALU per
                                  // thread); redone with readfirstlane + a scalar branch in nv12_r06.hip (profiles/r06a_ubench_nv12_one_store.txt)
        v = c == 0 ? o[0] : (c == 1 ? o[1] : o[2]);
    }
    const int g0 = gbase + i;
    const int off = g0 < groups ? 16 * g0 + c * (4 * plane) : kDrop - 8 * plane;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, AUX);
}

// ---- rp: row-pair mapping.  BLOCK = 2 * QB threads; grid.x = ceil(wq / QB) * (h / 2)
template <int QB, int PACE, int BAR>
__global__ __launch_bounds__(2 * QB) void k_rp(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a, int nqc) {
    const int wq = a.w >> 2, plane = a.w * a.h;
    const __amdgpu_buffer_rsrc_t rl = RSRC_SRC(a);
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST(a);
    const int p = blockIdx.x / nqc, qc = blockIdx.x - p * nqc;       // wave-uniform (scalar) division
    const int half = threadIdx.x >= QB ? 1 : 0;
    int xq0 = qc * QB + (threadIdx.x - half * QB);
    if constexpr (PACE) {  // the production kernel's per-lane integer division, result folded in so that it cannot be dropped
        const int g = (2 * p + half) * wq + xq0;
        const int rr = g / wq;
        xq0 = g - rr * wq;
    }
    const int xq = min(xq0, wq - 1), r = 2 * p + half;
    const uint32_t y4 = __builtin_amdgcn_raw_buffer_load_b32(rl, r * a.w + 4 * xq, 0, 0);
    const uint32_t uv4 = __builtin_amdgcn_raw_buffer_load_b32(rl, plane + p * a.w + 4 * xq, 0, 0);
    const int off = xq0 < wq ? 16 * (r * wq + xq) : kDrop - 8 * plane;
    f32x4 o[3];
    decode4(y4, uv4, a, o);
    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]));
    if constexpr (BAR) __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[c]), rs, off + c * (4 * plane), 0, AUX);
}

// ---- k2w: thread = quads g and g + 64 of a 128-quad wave segment
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_k2w(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * a.h, plane = a.w * a.h;
    const __amdgpu_buffer_rsrc_t rl = RSRC_SRC(a);
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gA0 = blockIdx.x * (2 * BLOCK) + wave * 128 + lane;
    uint32_t y4[2], uv4[2]; int off[2];
    const int gA = min(gA0, groups - 1);
    int r = gA / wq, xq = gA - r * wq;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k == 1) { xq += 64; if (xq >= wq) { xq -= wq; ++r; } }
        const bool ok = gA0 + 64 * k < groups;
        const int rr = ok ? r : a.h - 1, xx = ok ? xq : wq - 1;
        y4[k] = __builtin_amdgcn_raw_buffer_load_b32(rl, rr * a.w + 4 * xx, 0, 0);
        uv4[k] = __builtin_amdgcn_raw_buffer_load_b32(rl, plane + (rr >> 1) * a.w + 4 * xx, 0, 0);
        off[k] = ok ? 16 * (gA0 + 64 * k) : kDrop - 8 * plane;
    }
    f32x4 o[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) decode4(y4[k], uv4[k], a, o[k]);
#pragma unroll
    for (int k = 0; k < 2; ++k) asm volatile("" : "+v"(o[k][0]), "+v"(o[k][1]), "+v"(o[k][2]));
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 2; ++k) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[k][c]), rs, off[k] + c * (4 * plane), 0, AUX);
}

// ---- fills (the two ceilings bench.py also times in-process through libkornia_hip_diag.so)
__global__ __launch_bounds__(256) void f_flat(float* __restrict__ db, long long n4) {
    const long long i = (long long)blockIdx.y * gridDim.x * 256 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const long long base = i & ~((1ll << 26) - 1);  // one V# per 1 GiB window
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + 4 * base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, 4u}, rs, (int)(16 * (i - base)), 0, AUX);
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void f_3plane(float* __restrict__ db, Args a) {
    const int groups = (a.w >> 2) * a.h, g = blockIdx.x * BLOCK + threadIdx.x, plane = a.w * a.h;
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST(a);
    const int off = g < groups ? 16 * g : kDrop - 8 * plane;
#pragma unroll
    for (int c = 0; c < 3; ++c) __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, (unsigned)c}, rs, off + c * plane * 4, 0, AUX);
}

__global__ void k_diff(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, long long n, unsigned long long* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (; i < n; i += stride) if (a[i] != b[i]) ++bad;
    if (bad) atomicAdd(out, bad);
}

int main(int argc, char** argv) {
    const int W = 1920, H = 1080, N = argc > 1 ? atoi(argv[1]) : 1024, ROUNDS = argc > 2 ? atoi(argv[2]) : 7;
    const int NCHK = std::min(N, 16);
    const size_t fb = (size_t)W * H * 3 / 2, ob = (size_t)W * H * 3;
    uint8_t* src; float *dst, *ref;
    hipMemPool_t mp; CK(hipDeviceGetDefaultMemPool(&mp, 0)); uint64_t thr = UINT64_MAX; CK(hipMemPoolSetAttribute(mp, hipMemPoolAttrReleaseThreshold, &thr));
    CK(hipMallocAsync((void**)&src, fb * N, 0)); CK(hipMallocAsync((void**)&dst, ob * N * 4, 0)); CK(hipDeviceSynchronize());
    CK(hipMalloc(&ref, ob * NCHK * 4));
    {
        std::vector<uint8_t> h(fb + 31 * 64); uint32_t s = 0x12345678u;
        for (auto& b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
        for (int k = 0; k < N; ++k) CK(hipMemcpy(src + k * fb, h.data() + 31 * (k % 64), fb, hipMemcpyHostToDevice));
    }
    Args a{W, H, 0.485f, 0.456f, 0.406f, 1.0f / 0.229f, 1.0f / 0.224f, 1.0f / 0.225f, (long long)fb, (long long)ob};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long* dbad; CK(hipMalloc(&dbad, 16));
    const int wq = W / 4, groups = wq * H;
    const long long n4 = (long long)ob * N / 4;
    const double full = (double)(fb + ob * 4) * N, wonly = (double)ob * 4 * N;
    struct V { std::string name; double bytes; bool check; std::function<void()> run; std::vector<float> ms; long long bad; };
    std::vector<V> vs;
    auto G = [&](int per_block) { return dim3((groups + per_block - 1) / per_block, N); };
#define LIN(B, MODE, NAME) vs.push_back({NAME, full, MODE != 0 || B != 512, [&] { hipLaunchKernelGGL((k_lin<B, MODE>), G(B), dim3(B), 0, st, src, dst, a); }, {}, 0});
    LIN(512, 0, "base b512 (production shape)")
    LIN(512, 2, "sgb  b512 sched_group_barrier loads|valu|stores")
#define SPLIT(Q, MODE, NAME) vs.push_back({NAME, full, true, [&] { hipLaunchKernelGGL((k_split<Q, MODE>), G(Q), dim3(3 * Q), 0, st, src, dst, a); }, {}, 0});
    SPLIT(256, 0, "lds256  768 thr: 4 waves decode -> LDS -> 12 waves x 1 store")
    SPLIT(128, 0, "lds128  384 thr: 2 waves decode -> LDS -> 6 waves x 1 store")
    SPLIT(64, 0, "lds64   192 thr: 1 wave decodes -> LDS -> 3 waves x 1 store")
    SPLIT(256, 1, "all256  768 thr: every thread loads, decodes 1 channel, 1 store")
    SPLIT(128, 1, "all128  384 thr: every thread loads, decodes 1 channel, 1 store")
    SPLIT(64, 1, "all64   192 thr: every thread loads, decodes 1 channel, 1 store")
#define RP(QB, PACE, BAR, NAME) vs.push_back({NAME, full, true, [&] { const int nqc = (wq + QB - 1) / QB; hipLaunchKernelGGL((k_rp<QB, PACE, BAR>), dim3(nqc * (H / 2), N), dim3(2 * QB), 0, st, src, dst, a, nqc); }, {}, 0});
    vs.push_back({"F0 fill flat [sc0 sc1 nt]", wonly, false, [&] { hipLaunchKernelGGL(f_flat, dim3(65536, (unsigned)((n4 + 65536LL * 256 - 1) / (65536LL * 256))), dim3(256), 0, st, dst, n4); }, {}, 0});
    vs.push_back({"F1 W-only 3 planes/thread b512 [sc0 sc1 nt]", wonly, false, [&] { hipLaunchKernelGGL((f_3plane<512>), G(512), dim3(512), 0, st, dst, a); }, {}, 0});

    vs[0].run(); CK(hipGetLastError()); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(ref, dst, ob * NCHK * 4, hipMemcpyDeviceToDevice));
    for (auto& v : vs) {
        if (!v.check) continue;
        CK(hipMemsetAsync(dst, 0xCD, ob * NCHK * 4, st));
        v.run(); CK(hipGetLastError());
        CK(hipMemsetAsync(dbad, 0, 8, st));
        hipLaunchKernelGGL(k_diff, dim3(4096), dim3(256), 0, st, (const uint32_t*)ref, (const uint32_t*)dst, (long long)ob * NCHK, dbad);
        unsigned long long bad; CK(hipMemcpyAsync(&bad, dbad, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        v.bad = (long long)bad;
    }
    for (int r = 0; r < ROUNDS + 1; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, st)); v.run(); CK(hipGetLastError()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    printf("# N=%d frames of 1920x1080, %d rounds interleaved; GB/s = algorithmic bytes (R+W 28.67 GB, W-only 25.48 GB at N=1024) / median\n", N, ROUNDS);
    printf("%-56s %9s %9s %9s  %s\n", "variant", "med ms", "min ms", "GB/s@med", "vs base");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2];
        printf("%-56s %9.3f %9.3f %9.0f  %s\n", v.name.c_str(), med, v.ms[0], v.bytes / med / 1e6,
               !v.check ? "-" : (v.bad ? ("MISMATCH " + std::to_string(v.bad)).c_str() : "bit-equal"));
    }
    return 0;
}
