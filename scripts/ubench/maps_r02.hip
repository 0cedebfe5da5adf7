// Dev micro-benchmark (round 2, not shipped): pointwise colour maps.  The production f32 map is one pixel per thread
// (dwordx3 load, COUT-dword store): for gray a wave stores only 256 B per instruction.  Variants:
//   v0  production shape (global load / store)
//   v1  same, buffer load + write-through non-temporal buffer store
//   v2  wave = 256 consecutive pixels: 4 rounds of wave-contiguous dwordx3 loads, results transposed through a wave-private
//       LDS slice, ONE dwordx4 store per lane (1 KiB contiguous per wave), global / buffer-WT
//   v3  thread = 4 consecutive pixels: three dwordx4 loads at a 48-byte lane stride, one dwordx4 store
// for f32 RGB -> gray (12 B in, 4 B out), f32 RGB -> RGB "hsv-like" (12 in, 12 out) and u8 RGB -> gray (3 in, 1 out).
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
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
constexpr int kAux = 19;  // sc0 sc1 nt
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float gray(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }

// chunked addressing: blockIdx.y selects a 1 GiB-ish chunk so buffer offsets stay 32-bit
struct Span { long long px_per_chunk; long long npx; };

template <bool BUF>
__global__ __launch_bounds__(256) void gray_v01(const float* __restrict__ src, float* __restrict__ dst, Span s) {
    const long long base = (long long)blockIdx.y * s.px_per_chunk;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= s.px_per_chunk || base + p >= s.npx) return;
    if constexpr (BUF) {
        const auto rl = rsrc(src + base * 3, (unsigned)(s.px_per_chunk * 12)), rs = rsrc(dst + base, (unsigned)(s.px_per_chunk * 4));
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rl, p * 12, 0, 0);
        const float o = gray(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z));
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), rs, p * 4, 0, kAux);
    } else {
        const f32x3 v = *(const f32x3*)(src + (base + p) * 3);
        dst[base + p] = gray(v.x, v.y, v.z);
    }
}
template <int BLOCK, bool BUF>
__global__ __launch_bounds__(BLOCK) void gray_v2(const float* __restrict__ src, float* __restrict__ dst, Span s) {
    __shared__ float lds[BLOCK / 64][256];
    const long long base = (long long)blockIdx.y * s.px_per_chunk;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p0 = (blockIdx.x * (BLOCK / 64) + wv) * 256;  // first pixel of this wave within the chunk
    if (p0 >= s.px_per_chunk) return;
    const long long left = min(s.px_per_chunk - p0, s.npx - base - p0);
    const auto rl = rsrc(src + (base + p0) * 3, (unsigned)(min(left, 256ll) * 12)), rs = rsrc(dst + base + p0, (unsigned)(min(left, 256ll) * 4));
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = j * 64 + lane;
        if constexpr (BUF) {
            const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rl, p * 12, 0, 0);  // out of range -> zeros
            o[j] = gray(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z));
        } else {
            const long long q = min((long long)p, left - 1);
            const f32x3 v = *(const f32x3*)(src + (base + p0 + q) * 3);
            o[j] = gray(v.x, v.y, v.z);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) lds[wv][j * 64 + lane] = o[j];
    __builtin_amdgcn_wave_barrier();
    const f32x4 v = *(const f32x4*)&lds[wv][4 * lane];
    if constexpr (BUF) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, lane * 16, 0, kAux);
    else if (4 * lane + 3 < left) *(f32x4*)(dst + base + p0 + 4 * lane) = v;
    else for (int k = 0; k < 4; ++k) if (4 * lane + k < left) dst[base + p0 + 4 * lane + k] = v[k];
}
template <bool BUF>
__global__ __launch_bounds__(256) void gray_v3(const float* __restrict__ src, float* __restrict__ dst, Span s) {
    const long long base = (long long)blockIdx.y * s.px_per_chunk;
    const int q = blockIdx.x * 256 + threadIdx.x;  // group of 4 pixels
    if (4ll * q + 3 >= s.px_per_chunk || base + 4ll * q + 3 >= s.npx) return;  // (sizes here are multiples of 4)
    f32x4 a, b, c;
    if constexpr (BUF) {
        const auto rl = rsrc(src + base * 3, (unsigned)(s.px_per_chunk * 12));
        a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, q * 48, 0, 0));
        b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, q * 48 + 16, 0, 0));
        c = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, q * 48 + 32, 0, 0));
    } else {
        const f32x4* p = (const f32x4*)(src + (base + 4ll * q) * 3);
        a = p[0]; b = p[1]; c = p[2];
    }
    const f32x4 o = {gray(a.x, a.y, a.z), gray(a.w, b.x, b.y), gray(b.z, b.w, c.x), gray(c.y, c.z, c.w)};
    if constexpr (BUF) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsrc(dst + base, (unsigned)(s.px_per_chunk * 4)), q * 16, 0, kAux);
    else *(f32x4*)(dst + base + 4ll * q) = o;
}

// 3 -> 3 f32 map (stand-in for hsv / ycc: same traffic, a few dozen flops)
__device__ __forceinline__ f32x3 map33(f32x3 v) {
    const float mx = fmaxf(v.x, fmaxf(v.y, v.z)), mn = fminf(v.x, fminf(v.y, v.z)), d = mx - mn;
    return f32x3{d > 0.f ? (v.y - v.z) / d : 0.f, mx > 0.f ? d / mx : 0.f, mx};
}
template <int MODE>  // 0 global, 1 buffer WT store, 2 buffer load + WT store
__global__ __launch_bounds__(256) void m33(const float* __restrict__ src, float* __restrict__ dst, Span s) {
    const long long base = (long long)blockIdx.y * s.px_per_chunk;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= s.px_per_chunk || base + p >= s.npx) return;
    f32x3 v;
    if constexpr (MODE == 2) v = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(rsrc(src + base * 3, (unsigned)(s.px_per_chunk * 12)), p * 12, 0, 0));
    else v = *(const f32x3*)(src + (base + p) * 3);
    const f32x3 o = map33(v);
    if constexpr (MODE == 0) *(f32x3*)(dst + (base + p) * 3) = o;
    else __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, o), rsrc(dst + base * 3, (unsigned)(s.px_per_chunk * 12)), p * 12, 0, kAux);
}

// u8 RGB -> gray, thread = 4 px (3 dwords in, 1 dword out)
__device__ __forceinline__ uint32_t gray4(u32x3 w) {
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int b = 3 * j + k; c[k] = (w[b >> 2] >> (8 * (b & 3))) & 0xFFu; }
        out |= ((4899u * c[0] + 9617u * c[1] + 1868u * c[2] + 8192u) >> 14) << (8 * j);
    }
    return out;
}
template <int MODE>  // 0 global, 1 buffer WT
__global__ __launch_bounds__(256) void g8_v01(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, Span s) {
    const long long base = (long long)blockIdx.y * s.px_per_chunk;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (4ll * q + 3 >= s.px_per_chunk || base + 4ll * q + 3 >= s.npx) return;
    if constexpr (MODE == 0) {
        const u32x3 w = *(const u32x3*)(src + (base + 4ll * q) * 3);
        *(uint32_t*)(dst + base + 4ll * q) = gray4(w);
    } else {
        const u32x3 w = __builtin_amdgcn_raw_buffer_load_b96(rsrc(src + base * 3, (unsigned)(s.px_per_chunk * 3)), q * 12, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(gray4(w), rsrc(dst + base, (unsigned)s.px_per_chunk), q * 4, 0, kAux);
    }
}
template <int BLOCK, int MODE>  // wave = 1024 px: 4 rounds of wave-contiguous dwordx3 loads, LDS transpose, one dwordx4 store per lane
__global__ __launch_bounds__(BLOCK) void g8_v2(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, Span s) {
    __shared__ uint32_t lds[BLOCK / 64][256];
    const long long base = (long long)blockIdx.y * s.px_per_chunk;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q0 = (blockIdx.x * (BLOCK / 64) + wv) * 256;  // first 4-px group of this wave
    if (4ll * q0 >= s.px_per_chunk) return;
    const long long left = min(s.px_per_chunk - 4ll * q0, s.npx - base - 4ll * q0);  // pixels
    const auto rl = rsrc(src + (base + 4ll * q0) * 3, (unsigned)(min(left, 1024ll) * 3)), rs = rsrc(dst + base + 4ll * q0, (unsigned)min(left, 1024ll));
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = j * 64 + lane;
        if constexpr (MODE == 0) {
            const long long qq = min((long long)q, left / 4 - 1);
            o[j] = gray4(*(const u32x3*)(src + (base + 4ll * (q0 + qq)) * 3));
        } else {
            o[j] = gray4(__builtin_amdgcn_raw_buffer_load_b96(rl, q * 12, 0, 0));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) lds[wv][j * 64 + lane] = o[j];
    __builtin_amdgcn_wave_barrier();
    const u32x4 v = *(const u32x4*)&lds[wv][4 * lane];
    if constexpr (MODE == 0) { if (16 * lane + 15 < left) *(u32x4*)(dst + base + 4ll * q0 + 16 * lane) = v; }
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, 0, kAux);
}

__global__ void k_diff(const uint32_t* a, const uint32_t* b, long long n, unsigned long long* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (; i < n; i += stride) bad += a[i] != b[i];
    if (bad) atomicAdd(out, bad);
}

int main(int argc, char** argv) {
    const long long NPX = 1920ll * 1080 * (argc > 1 ? atoi(argv[1]) : 512);
    const int ROUNDS = argc > 2 ? atoi(argv[2]) : 5;
    float *src, *dst, *ref;
    CK(hipMalloc(&src, NPX * 12)); CK(hipMalloc(&dst, NPX * 12)); CK(hipMalloc(&ref, NPX * 12));
    {
        std::vector<float> h(1 << 22); uint32_t st = 0x12345678u;
        for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (float)(st >> 24) / 255.0f; }
        for (long long off = 0; off < NPX * 3; off += (long long)h.size())
            CK(hipMemcpy(src + off, h.data(), std::min<long long>(h.size(), NPX * 3 - off) * 4, hipMemcpyHostToDevice));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long* dbad; CK(hipMalloc(&dbad, 8));
    const long long CH = 1920ll * 1080 * 32;  // pixels per chunk: 32 frames (796 MB of f32 RGB)
    const Span s{CH, NPX};
    const unsigned chunks = (unsigned)((NPX + CH - 1) / CH);
    auto G = [&](long long per_block) { return dim3((unsigned)((CH + per_block - 1) / per_block), chunks); };
    struct V { std::string name; int fam; double bytes; std::function<void()> run; std::vector<float> ms; long long bad; };
    std::vector<V> vs;
    const double b_gray = NPX * 16.0, b_33 = NPX * 24.0, b_g8 = NPX * 4.0;
    vs.push_back({"gray f32 v0 1px/thread global (production shape)", 0, b_gray, [&] { hipLaunchKernelGGL((gray_v01<false>), G(256), dim3(256), 0, st, src, dst, s); }, {}, 0});
    vs.push_back({"gray f32 v1 1px/thread buffer + WT store", 0, b_gray, [&] { hipLaunchKernelGGL((gray_v01<true>), G(256), dim3(256), 0, st, src, dst, s); }, {}, 0});
    vs.push_back({"gray f32 v2 wave=256px LDS-transposed dwordx4 store, global b256", 0, b_gray, [&] { hipLaunchKernelGGL((gray_v2<256, false>), G(1024), dim3(256), 0, st, src, dst, s); }, {}, 0});
    vs.push_back({"gray f32 v2 wave=256px LDS-transposed dwordx4 store, buffer WT b256", 0, b_gray, [&] { hipLaunchKernelGGL((gray_v2<256, true>), G(1024), dim3(256), 0, st, src, dst, s); }, {}, 0});
    vs.push_back({"gray f32 v2 ... buffer WT b128", 0, b_gray, [&] { hipLaunchKernelGGL((gray_v2<128, true>), G(512), dim3(128), 0, st, src, dst, s); }, {}, 0});
    vs.push_back({"gray f32 v2 ... buffer WT b512", 0, b_gray, [&] { hipLaunchKernelGGL((gray_v2<512, true>), G(2048), dim3(512), 0, st, src, dst, s); }, {}, 0});
    vs.push_back({"gray f32 v3 4px/thread 3x dwordx4 strided loads, global", 0, b_gray, [&] { hipLaunchKernelGGL((gray_v3<false>), G(1024), dim3(256), 0, st, src, dst, s); }, {}, 0});
    vs.push_back({"gray f32 v3 4px/thread ... buffer WT", 0, b_gray, [&] { hipLaunchKernelGGL((gray_v3<true>), G(1024), dim3(256), 0, st, src, dst, s); }, {}, 0});
    vs.push_back({"map 3->3 f32 global (production shape)", 1, b_33, [&] { hipLaunchKernelGGL((m33<0>), G(256), dim3(256), 0, st, src, dst, s); }, {}, 0});
    vs.push_back({"map 3->3 f32 global load + WT buffer store", 1, b_33, [&] { hipLaunchKernelGGL((m33<1>), G(256), dim3(256), 0, st, src, dst, s); }, {}, 0});
    vs.push_back({"map 3->3 f32 buffer load + WT buffer store", 1, b_33, [&] { hipLaunchKernelGGL((m33<2>), G(256), dim3(256), 0, st, src, dst, s); }, {}, 0});
    const uint8_t* s8 = (const uint8_t*)src; uint8_t* d8 = (uint8_t*)dst;
    vs.push_back({"gray u8 v0 4px/thread global (production shape)", 2, b_g8, [&] { hipLaunchKernelGGL((g8_v01<0>), G(1024), dim3(256), 0, st, s8, d8, s); }, {}, 0});
    vs.push_back({"gray u8 v1 4px/thread buffer + WT store", 2, b_g8, [&] { hipLaunchKernelGGL((g8_v01<1>), G(1024), dim3(256), 0, st, s8, d8, s); }, {}, 0});
    vs.push_back({"gray u8 v2 wave=1024px LDS-transposed dwordx4 store, global b256", 2, b_g8, [&] { hipLaunchKernelGGL((g8_v2<256, 0>), G(4096), dim3(256), 0, st, s8, d8, s); }, {}, 0});
    vs.push_back({"gray u8 v2 ... buffer WT b256", 2, b_g8, [&] { hipLaunchKernelGGL((g8_v2<256, 1>), G(4096), dim3(256), 0, st, s8, d8, s); }, {}, 0});
    vs.push_back({"gray u8 v2 ... buffer WT b512", 2, b_g8, [&] { hipLaunchKernelGGL((g8_v2<512, 1>), G(8192), dim3(512), 0, st, s8, d8, s); }, {}, 0});

    const long long words[3] = {NPX, NPX * 3, NPX / 4};
    int fam = -1;
    for (auto& v : vs) {
        CK(hipMemsetAsync(dst, 0xCD, words[v.fam] * 4, st));
        v.run(); CK(hipGetLastError());
        if (v.fam != fam) { fam = v.fam; CK(hipMemcpyAsync(ref, dst, words[fam] * 4, hipMemcpyDeviceToDevice, st)); CK(hipStreamSynchronize(st)); continue; }
        CK(hipMemsetAsync(dbad, 0, 8, st));
        hipLaunchKernelGGL(k_diff, dim3(4096), dim3(256), 0, st, (const uint32_t*)ref, (const uint32_t*)dst, words[fam], dbad);
        unsigned long long bad; CK(hipMemcpyAsync(&bad, dbad, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        v.bad = (long long)bad;
    }
    for (int r = 0; r < ROUNDS + 1; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, st)); v.run(); CK(hipGetLastError()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    printf("# %lld pixels (%lld 1080p frames), %d rounds interleaved; GB/s = (read + written) bytes / median\n", NPX, NPX / (1920 * 1080), ROUNDS);
    printf("%-72s %9s %9s %9s  %s\n", "variant", "med ms", "min ms", "GB/s@med", "vs first of family");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        printf("%-72s %9.3f %9.3f %9.0f  %s\n", v.name.c_str(), med, v.ms[0], v.bytes / med / 1e6, v.bad ? ("MISMATCH " + std::to_string(v.bad)).c_str() : "equal");
    }
    return 0;
}
