// Round 6, second file: three<WPC, CH> — one channel per WAVE again, but CH consecutive 1 KiB chunks of that plane per wave (3 or 6 KiB in flight per
// wave, the stores of a wave ADJACENT: the shape that ran at the fill rate in normalize_rgb_u8), the three channel groups of the same pixels in ONE block
// (their source loads share L1 / L2).
// Dev micro-benchmark (round 6, not shipped).  VERDICT r05 "What's weak" 2: round 4's one-store-per-wave variant (nv12_r04.hip
// k_split<Q,1>) decoded all three channels in every thread (c = t / Q was not known wave-uniform to the compiler: 159 VALU per thread,
// 3.2x the vector work of the production shape), so its 8.0 ms was an ALU artefact.  This file redoes it correctly:
//   one<Q>      block = 3Q threads, c = readfirstlane(t / Q), decode ONLY channel c behind a scalar branch; the row / column of a quad
//               from a scalar division of the block base + one conditional wrap (Q <= w / 4), no per-lane integer division; buffer loads,
//               write-through non-temporal stores.  Every wave issues two 4-byte loads and ONE 16-byte store.
//   onepf<Q>    the same + lanes 0..L-1 of wave 0 touch one dword of each 128-byte source line of the block `dist` places further down
//               this XCD's queue (linear block id + 8 * dist), so that the real loads of that block hit this XCD's L2.
//   chan<B>     the channel is a property of the BLOCK: linear block id = ((group * 3 + c) * 8 + xcd); the three blocks that decode the
//               three channels of a chunk run on ONE XCD (its L2 serves the second and third read of the source), B threads each.
//   base        the production shape (thread = quad, three stores), as in kh_preprocess.hip.
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
// ONE channel of a quad: the same integer expressions as decode4, channel C known at compile time.
template <int C>
__device__ __forceinline__ f32x4 decode1(uint32_t y4, uint32_t uv4, const Args& a) {
    int t[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = (int)((uv4 >> (16 * k)) & 0xFFu) - 128, v = (int)((uv4 >> (16 * k + 8)) & 0xFFu) - 128;
        t[k] = C == 0 ? kCVR * v + kHalf20 : (C == 1 ? kCUG * u + kCVG * v + kHalf20 : kCUB * u + kHalf20);
    }
    const float m = C == 0 ? a.m0 : (C == 1 ? a.m1 : a.m2), is = C == 0 ? a.is0 : (C == 1 ? a.is1 : a.is2);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = max((int)((y4 >> (8 * j)) & 0xFFu) - 16, 0) * kCY;
        o[j] = norm1(clamp255((yy + t[j >> 1]) >> 20), m, is);
    }
    return o;
}
#define RSRC_SRC_F(a, f) __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sb + (long long)(f) * a.sfs), 0, plane + plane / 2, 0x00020000)
#define RSRC_DST_F(a, f) __builtin_amdgcn_make_buffer_rsrc(db + (long long)(f) * a.dfs, 0, 12 * plane, 0x00020000)
constexpr int kDrop = 0x7fffffff;

// ---- base: production mapping (thread = quad g of the frame, linear, per-lane division kept: it paces the loads, r02)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_lin(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * a.h, plane = a.w * a.h;
    const __amdgpu_buffer_rsrc_t rl = RSRC_SRC_F(a, blockIdx.y);
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST_F(a, blockIdx.y);
    const int g0 = blockIdx.x * BLOCK + threadIdx.x, g = min(g0, groups - 1);
    const int r = g / wq, xq = g - r * wq;
    const uint32_t y4 = __builtin_amdgcn_raw_buffer_load_b32(rl, r * a.w + 4 * xq, 0, 0);
    const uint32_t uv4 = __builtin_amdgcn_raw_buffer_load_b32(rl, plane + (r >> 1) * a.w + 4 * xq, 0, 0);
    const int off = g0 < groups ? 16 * g : kDrop - 8 * plane;
    f32x4 o[3];
    decode4(y4, uv4, a, o);
    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]));
#pragma unroll
    for (int c = 0; c < 3; ++c) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[c]), rs, off + c * (4 * plane), 0, AUX);
}

// ---- tab<K>: the production shape with the frame base taken from a table of K device pointers passed BY VALUE in the kernel arguments
// (8 KiB of kernarg at K = 1024): the reference's run_raw_batch signature is a slice of separately allocated frame buffers
// (P/preprocess.rs:1258-1282); one launch for N arbitrary buffers needs the bases somewhere the kernel can read with a scalar load.
template <int K> struct Tab { const uint8_t* p[K]; };
template <int BLOCK, int K>
__global__ __launch_bounds__(BLOCK) void k_tab(Tab<K> tab, float* __restrict__ db, Args a, int first) {
    const int wq = a.w >> 2, groups = wq * a.h, plane = a.w * a.h;
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(tab.p[blockIdx.y]), 0, plane + plane / 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST_F(a, first + blockIdx.y);
    const int g0 = blockIdx.x * BLOCK + threadIdx.x, g = min(g0, groups - 1);
    const int r = g / wq, xq = g - r * wq;
    const uint32_t y4 = __builtin_amdgcn_raw_buffer_load_b32(rl, r * a.w + 4 * xq, 0, 0);
    const uint32_t uv4 = __builtin_amdgcn_raw_buffer_load_b32(rl, plane + (r >> 1) * a.w + 4 * xq, 0, 0);
    const int off = g0 < groups ? 16 * g : kDrop - 8 * plane;
    f32x4 o[3];
    decode4(y4, uv4, a, o);
    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]));
#pragma unroll
    for (int c = 0; c < 3; ++c) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[c]), rs, off + c * (4 * plane), 0, AUX);
}

// ---- one<Q>: one store per wave, channel = wave-uniform scalar, single-channel decode
// DIV: 0 = scalar division of the block base + one wrap; 1 = the production kernel's per-lane integer division (pacing)
template <int Q, int PF, int DIV>
__global__ __launch_bounds__(3 * Q) void k_one(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a, int dist) {
    const int wq = a.w >> 2, groups = wq * a.h, plane = a.w * a.h;
    const __amdgpu_buffer_rsrc_t rl = RSRC_SRC_F(a, blockIdx.y);
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST_F(a, blockIdx.y);
    const int t = threadIdx.x;
    const int c = __builtin_amdgcn_readfirstlane(t / Q);   // Q is a multiple of 64: uniform per wave, and now the compiler knows it
    const int i = t - c * Q, gbase = blockIdx.x * Q;
    uint32_t pf = 0;
    if constexpr (PF) {
        // the block `dist` places further down this XCD's queue; its source segment is [gq * 4, gq * 4 + 4 Q) luma bytes (may straddle two
        // rows) + the matching chroma bytes: touch one dword per 128-byte line of both
        constexpr int kLines = (4 * Q + 127) / 128 + 1;
        if (t < 2 * kLines) {
            unsigned lin = blockIdx.y * gridDim.x + blockIdx.x + 8u * (unsigned)dist;
            const unsigned fy = lin / gridDim.x, fx = lin - fy * gridDim.x;
            if (fy < gridDim.y) {
                const __amdgpu_buffer_rsrc_t rp = RSRC_SRC_F(a, fy);
                const int gq = fx * Q, rr = gq / wq, xx = gq - rr * wq;
                const int l = t >> 1;
                const int luma = rr * a.w + 4 * xx + 128 * l;                       // linear in the luma plane across the row wrap
                const int r2 = (4 * xx + 128 * l) >= a.w ? rr + 1 : rr;
                const int chroma = plane + (r2 >> 1) * a.w + (4 * xx + 128 * l) % a.w;
                pf = __builtin_amdgcn_raw_buffer_load_b32(rp, (t & 1) ? chroma : luma, 0, 0);
            }
        }
    }
    int r, xq;
    const int g0 = gbase + i;
    if constexpr (DIV == 0) {
        const int r0 = gbase / wq, x0 = gbase - r0 * wq;   // scalar
        xq = x0 + i; r = r0;
        if (xq >= wq) { xq -= wq; ++r; }
        if (g0 >= groups) { r = a.h - 1; xq = wq - 1; }
    } else {
        const int g = min(g0, groups - 1);
        r = g / wq; xq = g - r * wq;
    }
    const uint32_t y4 = __builtin_amdgcn_raw_buffer_load_b32(rl, r * a.w + 4 * xq, 0, 0);
    const uint32_t uv4 = __builtin_amdgcn_raw_buffer_load_b32(rl, plane + (r >> 1) * a.w + 4 * xq, 0, 0);
    f32x4 v;
    if (c == 0) v = decode1<0>(y4, uv4, a);
    else if (c == 1) v = decode1<1>(y4, uv4, a);
    else v = decode1<2>(y4, uv4, a);
    const int off = g0 < groups ? 16 * g0 + c * (4 * plane) : kDrop - 8 * plane;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, AUX);
    if constexpr (PF) asm volatile("" :: "v"(pf));
}

// ---- chan<B>: channel per BLOCK; the three blocks of a chunk share an XCD.  grid.x = 3 * chunks rounded up to a multiple of 24, 1-D
// over the whole batch (chunk id = frame * cpf + chunk)
template <int B>
__global__ __launch_bounds__(B) void k_chan(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a, unsigned cpf, unsigned chunks) {
    const int wq = a.w >> 2, groups = wq * a.h, plane = a.w * a.h;
    const unsigned b = blockIdx.x, xcd = b & 7u, slot = b >> 3;          // slot = group * 3 + c
    const unsigned grp = slot / 3u;
    const int c = (int)(slot - grp * 3u);
    const unsigned chunk = grp * 8u + xcd;
    if (chunk >= chunks) return;
    const unsigned frame = chunk / cpf, ch = chunk - frame * cpf;
    const __amdgpu_buffer_rsrc_t rl = RSRC_SRC_F(a, frame);
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST_F(a, frame);
    const int gbase = ch * B, i = threadIdx.x, g0 = gbase + i;
    const int r0 = gbase / wq, x0 = gbase - r0 * wq;   // scalar
    int xq = x0 + i, r = r0;
    if (xq >= wq) { xq -= wq; ++r; }
    if (xq >= wq) { xq -= wq; ++r; }                   // B up to 2 * wq
    if (g0 >= groups) { r = a.h - 1; xq = wq - 1; }
    const uint32_t y4 = __builtin_amdgcn_raw_buffer_load_b32(rl, r * a.w + 4 * xq, 0, 0);
    const uint32_t uv4 = __builtin_amdgcn_raw_buffer_load_b32(rl, plane + (r >> 1) * a.w + 4 * xq, 0, 0);
    f32x4 v;
    if (c == 0) v = decode1<0>(y4, uv4, a);
    else if (c == 1) v = decode1<1>(y4, uv4, a);
    else v = decode1<2>(y4, uv4, a);
    const int off = g0 < groups ? 16 * g0 + c * (4 * plane) : kDrop - 8 * plane;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, AUX);
}

// ---- three<WPC, CH>: block = 3 * WPC waves; wave -> (channel c = wave / WPC, column group w4 = wave % WPC); a wave owns CH consecutive
// 1 KiB chunks of plane c (CH * 256 pixels); a thread decodes channel c of CH quads 256 pixels apart.
template <int WPC, int CH>
__global__ __launch_bounds__(192 * WPC) void k_three(const uint8_t* __restrict__ sb, float* __restrict__ db, Args a) {
    const int wq = a.w >> 2, groups = wq * a.h, plane = a.w * a.h;
    const __amdgpu_buffer_rsrc_t rl = RSRC_SRC_F(a, blockIdx.y);
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST_F(a, blockIdx.y);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = wave / WPC, w4 = wave - c * WPC;
    const int gbase = blockIdx.x * (WPC * 64 * CH) + w4 * (64 * CH);
    uint32_t y4[CH], uv4[CH];
    int off[CH];
#pragma unroll
    for (int g = 0; g < CH; ++g) {
        const int G = gbase + g * 64 + lane, Gc = min(G, groups - 1);
        const int r = Gc / wq, xq = Gc - r * wq;
        y4[g] = __builtin_amdgcn_raw_buffer_load_b32(rl, r * a.w + 4 * xq, 0, 0);
        uv4[g] = __builtin_amdgcn_raw_buffer_load_b32(rl, plane + (r >> 1) * a.w + 4 * xq, 0, 0);
        off[g] = G < groups ? 16 * G + c * (4 * plane) : kDrop - 8 * plane;
    }
    f32x4 v[CH];
    if (c == 0) { for (int g = 0; g < CH; ++g) v[g] = decode1<0>(y4[g], uv4[g], a); }
    else if (c == 1) { for (int g = 0; g < CH; ++g) v[g] = decode1<1>(y4[g], uv4[g], a); }
    else { for (int g = 0; g < CH; ++g) v[g] = decode1<2>(y4[g], uv4[g], a); }
#pragma unroll
    for (int g = 0; g < CH; ++g) asm volatile("" : "+v"(v[g]));
#pragma unroll
    for (int g = 0; g < CH; ++g) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[g]), rs, off[g], 0, AUX);
}

// ---- fills (ceilings)
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
    const __amdgpu_buffer_rsrc_t rs = RSRC_DST_F(a, blockIdx.y);
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
    vs.push_back({"base b512 (production shape, 3 stores / wave)", full, false, [&] { hipLaunchKernelGGL((k_lin<512>), G(512), dim3(512), 0, st, src, dst, a); }, {}, 0});
#define THREE(WPC, CH, NAME) vs.push_back({NAME, full, true, [&] { hipLaunchKernelGGL((k_three<WPC, CH>), G(WPC * 64 * CH), dim3(192 * WPC), 0, st, src, dst, a); }, {}, 0});
    THREE(1, 3, "three w1 c3: 192 thr, a wave = 3 KiB of one plane")
    THREE(2, 3, "three w2 c3: 384 thr")
    THREE(4, 3, "three w4 c3: 768 thr")
    THREE(1, 6, "three w1 c6: 192 thr, a wave = 6 KiB of one plane")
    THREE(2, 6, "three w2 c6: 384 thr")
    THREE(4, 6, "three w4 c6: 768 thr")
    THREE(2, 2, "three w2 c2: 384 thr, a wave = 2 KiB")
    THREE(4, 1, "three w4 c1: 768 thr, a wave = 1 KiB (= one256)")
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
    printf("%-64s %9s %9s %9s %7s  %s\n", "variant", "med ms", "min ms", "GB/s@med", "frac", "vs base");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2];
        printf("%-64s %9.3f %9.3f %9.0f %7.3f  %s\n", v.name.c_str(), med, v.ms[0], v.bytes / med / 1e6, v.bytes / med / 1e6 / 8000.0,
               !v.check ? "-" : (v.bad ? ("MISMATCH " + std::to_string(v.bad)).c_str() : "bit-equal"));
    }
    return 0;
}
