// Dev micro-benchmark (round 4, not shipped): WHY do the north star's three plane stores cost 18 % more than a flat fill of the
// same bytes?    Every variant writes the same 1024 x 24 883 200 B with the production store policy
// (write-through non-temporal 16-B buffer stores) and nothing else — no loads, no decode — so any difference is the STORE SHAPE.
// Each variant is its own kernel symbol, so a rocprofv3 --pmc pass attributes counters per variant.
//
//   flat256        flat fill, 256-thread blocks, one store per thread                       (khd_flat_fill)
//   flat512x3      flat fill, 512-thread blocks, THREE stores per thread into three consecutive 8 KiB chunks: the production
//                  kernel's stores-per-thread and block size, but ONE stream
//   planes         production shape: a thread stores 16 B into each of the three planes of its frame, planes 8 294 400 B apart
//   pad256/4352/33024   the same with the plane stride padded by that many bytes (NOT within the CHW contract: diagnostic only —
//                  does the 8 294 400-B spacing alias channels / banks?)
//   rotwave        production shape, plane order rotated per wave: (c + wave) % 3                    (within contract)
//   rotblock       production shape, plane order rotated per block: (c + block) % 3                  (within contract)
//   oneplane       three streams but ONE store per thread: blockIdx.z picks the plane (3x the blocks)
//   frames3        one store per thread per FRAME-sized stream... i.e. three streams 24 883 200 B apart (a thread writes the
//                  same offset of three consecutive frames' flat bytes): three streams at a different spacing, flat content
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX = 19;  // sc0 sc1 nt
constexpr int W = 1920, H = 1080, PLANE = W * H, GROUPS = PLANE / 4;

__global__ __launch_bounds__(256) void flat256(float* __restrict__ db, long long n4) {
    const long long i = (long long)blockIdx.y * gridDim.x * 256 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const long long base = i & ~((1ll << 26) - 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + 4 * base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, 4u}, rs, (int)(16 * (i - base)), 0, AUX);
}

// one stream, three stores per thread: block b of frame f owns bytes [b * 24 KiB, (b + 1) * 24 KiB) of the frame
__global__ __launch_bounds__(512) void flat512x3(float* __restrict__ db, long long dfs) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + (long long)blockIdx.y * dfs, 0, 12 * PLANE, 0x00020000);
    const int off = blockIdx.x * (3 * 8192) + 16 * threadIdx.x;
#pragma unroll
    for (int c = 0; c < 3; ++c) __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, (unsigned)c}, rs, off + c * 8192, 0, AUX);
}

template <int ROT>  // 0 none, 1 per wave, 2 per block
__global__ __launch_bounds__(512) void planes(float* __restrict__ db, long long dfs, int plane_stride_bytes) {
    const int g = blockIdx.x * 512 + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + (long long)blockIdx.y * dfs, 0, 3 * plane_stride_bytes, 0x00020000);
    const int off = g < GROUPS ? 16 * g : 0x7fffffff - 2 * plane_stride_bytes - 16;
    const int rot = ROT == 1 ? (int)((threadIdx.x >> 6) + 8 * blockIdx.x) % 3 : ROT == 2 ? (int)(blockIdx.x % 3) : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int p = c + rot;
        p = p >= 3 ? p - 3 : p;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, (unsigned)p}, rs, off + p * plane_stride_bytes, 0, AUX);
    }
}

__global__ __launch_bounds__(512) void oneplane(float* __restrict__ db, long long dfs) {
    const int g = blockIdx.x * 512 + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + (long long)blockIdx.y * dfs, 0, 12 * PLANE, 0x00020000);
    const int off = g < GROUPS ? 16 * g + (int)blockIdx.z * 4 * PLANE : 0x7ffffff0;
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, blockIdx.z}, rs, off, 0, AUX);
}

// three streams one FRAME (24 883 200 B) apart: blockIdx.y = frame triple, a thread writes the same flat offset of three frames
__global__ __launch_bounds__(512) void frames3(float* __restrict__ db, long long dfs, int chunks_per_frame) {
    const int q = blockIdx.x * 512 + threadIdx.x;       // 16-B quad index within a frame's flat bytes (3 * GROUPS of them)
    if (q >= 3 * GROUPS) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + (long long)(3 * blockIdx.y + c) * dfs, 0, 12 * PLANE, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, (unsigned)c}, rs, 16 * q, 0, AUX);
    }
}

// ---- the shape grid: ONE flat stream; block size B, S stores per thread, three layouts, optional serialisation
//   LAYOUT 0  block-contiguous: store s of thread t at block_base + s * (16 B) + 16 t   (S chunks of 16 B bytes per block)
//   LAYOUT 1  wave-contiguous:  a wave writes S KiB contiguous: wave_base + s * 1 KiB + 16 lane
//   GAP 0 back to back, 1 = s_waitcnt vmcnt(0) after every store, 2 = s_sleep 8 between stores
template <int B, int S, int LAYOUT, int GAP>
__global__ __launch_bounds__(B) void flat_bs(float* __restrict__ db, long long n4) {
    const long long q0 = LAYOUT == 0 ? (long long)blockIdx.x * (B * S) + threadIdx.x
                                      : ((long long)blockIdx.x * (B / 64) + (threadIdx.x >> 6)) * (64 * S) + (threadIdx.x & 63);
    constexpr int STEP = LAYOUT == 0 ? B : 64;
    const long long base = q0 & ~((1ll << 26) - 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + 4 * base, 0, 0x7fffffff, 0x00020000);
    const int off = (int)(16 * (q0 - base));
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (q0 + (long long)s * STEP < n4) __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, (unsigned)s}, rs, off + 16 * s * STEP, 0, AUX);
        if (GAP == 1) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
        if (GAP == 2 && s + 1 < S) __builtin_amdgcn_s_sleep(8);
    }
}
// occupancy-limited flat512x3: LDS bytes per block chosen by the host (160 KiB / CU)
template <int B, int S>
__global__ __launch_bounds__(B) void flat_bs_lds(float* __restrict__ db, long long n4) {
    extern __shared__ uint32_t pad_[];
    if (n4 < 0) pad_[threadIdx.x] = 1;
    const long long q0 = (long long)blockIdx.x * (B * S) + threadIdx.x;
    const long long base = q0 & ~((1ll << 26) - 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + 4 * base, 0, 0x7fffffff, 0x00020000);
    const int off = (int)(16 * (q0 - base));
#pragma unroll
    for (int s = 0; s < S; ++s)
        if (q0 + (long long)s * B < n4) __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, (unsigned)s}, rs, off + 16 * s * B, 0, AUX);
}
template <int B>
__global__ __launch_bounds__(B) void oneplane_b(float* __restrict__ db, long long dfs) {
    const int g = blockIdx.x * B + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + (long long)blockIdx.y * dfs, 0, 12 * PLANE, 0x00020000);
    const int off = g < GROUPS ? 16 * g + (int)blockIdx.z * 4 * PLANE : 0x7ffffff0;
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, blockIdx.z}, rs, off, 0, AUX);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1024, ROUNDS = argc > 2 ? atoi(argv[2]) : 7;
    const char* only = argc > 3 ? argv[3] : "";
    const int pads[3] = {256, 4352, 33024};
    const long long frame_f = 3LL * PLANE, pad_max = 33024;
    float* dst;
    const size_t bytes = (size_t)N * (12 * (size_t)PLANE + 3 * pad_max) + (1 << 20);
    CK(hipMalloc(&dst, bytes));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long n4 = frame_f * N / 4;
    const dim3 G((GROUPS + 511) / 512, N);
    struct V { std::string name; std::function<void()> run; std::vector<float> ms; double frac = 1.0; };
    std::vector<V> vs;
    vs.push_back({"flat256", [&] { hipLaunchKernelGGL(flat256, dim3(65536, (unsigned)((n4 + 65536LL * 256 - 1) / (65536LL * 256))), dim3(256), 0, st, dst, n4); }, {}});
    vs.push_back({"flat512x3", [&] { hipLaunchKernelGGL(flat512x3, dim3(12 * PLANE / (3 * 8192), N), dim3(512), 0, st, dst, frame_f); }, {}});  // 1012.5 -> 1012 chunks (tail skipped: 0.05 %)
    vs.push_back({"planes", [&] { hipLaunchKernelGGL(planes<0>, G, dim3(512), 0, st, dst, frame_f, 4 * PLANE); }, {}});
    for (int p : pads)
        vs.push_back({"pad" + std::to_string(p), [&, p] { hipLaunchKernelGGL(planes<0>, G, dim3(512), 0, st, dst, frame_f + 3 * p / 4, 4 * PLANE + p); }, {}});
    vs.push_back({"rotwave", [&] { hipLaunchKernelGGL(planes<1>, G, dim3(512), 0, st, dst, frame_f, 4 * PLANE); }, {}});
    vs.push_back({"rotblock", [&] { hipLaunchKernelGGL(planes<2>, G, dim3(512), 0, st, dst, frame_f, 4 * PLANE); }, {}});
    vs.push_back({"oneplane", [&] { hipLaunchKernelGGL(oneplane, dim3(G.x, N, 3), dim3(512), 0, st, dst, frame_f); }, {}});
    vs.push_back({"frames3", [&] { hipLaunchKernelGGL(frames3, dim3((3 * GROUPS + 511) / 512, N / 3), dim3(512), 0, st, dst, frame_f, 0); }, {}});
    vs.back().frac = (double)(N / 3 * 3) / N;
    for (auto& v : vs) if (v.name == "flat512x3") v.frac = 1012.0 / 1012.5;
#define FBS(B, S, L, G, NAME) vs.push_back({NAME, [&] { hipLaunchKernelGGL((flat_bs<B, S, L, G>), dim3((unsigned)((n4 + (long long)B * S - 1) / ((long long)B * S))), dim3(B), 0, st, dst, n4); }, {}});
    FBS(256, 1, 0, 0, "g256x1") FBS(512, 1, 0, 0, "g512x1") FBS(1024, 1, 0, 0, "g1024x1")
    FBS(256, 2, 0, 0, "g256x2") FBS(512, 2, 0, 0, "g512x2") FBS(1024, 2, 0, 0, "g1024x2")
    FBS(256, 3, 0, 0, "g256x3") FBS(512, 3, 0, 0, "g512x3") FBS(1024, 3, 0, 0, "g1024x3")
    FBS(64, 1, 0, 0, "g64x1") FBS(64, 3, 0, 0, "g64x3") FBS(128, 3, 0, 0, "g128x3") FBS(256, 6, 0, 0, "g256x6")
    FBS(256, 3, 1, 0, "w256x3") FBS(512, 3, 1, 0, "w512x3") FBS(512, 6, 1, 0, "w512x6")
    FBS(512, 3, 0, 1, "g512x3wait") FBS(512, 3, 0, 2, "g512x3sleep") FBS(256, 3, 0, 1, "g256x3wait")
#define FLDS(B, S, KIB, NAME) vs.push_back({NAME, [&] { hipLaunchKernelGGL((flat_bs_lds<B, S>), dim3((unsigned)((n4 + (long long)B * S - 1) / ((long long)B * S))), dim3(B), KIB * 1024, st, dst, n4); }, {}});
    FLDS(512, 3, 40, "g512x3occ4") FLDS(512, 3, 64, "g512x3occ2") FLDS(512, 3, 100, "g512x3occ1") FLDS(256, 1, 40, "g256x1occ4") FLDS(256, 1, 20, "g256x1occ8")
    vs.push_back({"oneplane256", [&] { hipLaunchKernelGGL(oneplane_b<256>, dim3((GROUPS + 255) / 256, N, 3), dim3(256), 0, st, dst, frame_f); }, {}});
    vs.push_back({"oneplane1024", [&] { hipLaunchKernelGGL(oneplane_b<1024>, dim3((GROUPS + 1023) / 1024, N, 3), dim3(1024), 0, st, dst, frame_f); }, {}});
    if (*only) vs.erase(std::remove_if(vs.begin(), vs.end(), [&](const V& v) { return !strstr(only, v.name.c_str()); }), vs.end());
    for (int r = 0; r < ROUNDS + 1; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, st)); v.run(); CK(hipGetLastError()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    const double wbytes = 12.0 * PLANE * N;
    printf("# store shapes, N=%d frames x 24 883 200 B, %d rounds interleaved, policy sc0 sc1 nt\n", N, ROUNDS);
    printf("%-14s %9s %9s %9s %8s\n", "variant", "med ms", "min ms", "GB/s@med", "vs flat");
    float flat = 0;
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        if (v.name == "flat256") flat = med;
        printf("%-14s %9.3f %9.3f %9.0f %8.3f\n", v.name.c_str(), med, v.ms[0], wbytes * v.frac / med / 1e6, flat > 0 ? med / flat : 0.0f);
    }
    return 0;
}
