// Dev micro-benchmark (round 5, not shipped): for COPIES the cost of several accesses per wave depends on the distance between a wave's own
// consecutive accesses (profiles/r05o).  Does the same hold for the north star's STORE shape?  Every variant writes the same 25.5 GB with
// three 16-byte stores per thread (512-thread blocks, sc0 sc1 nt), the three 8 KiB block chunks `stride` bytes apart.  The CHW contract
// fixes the real kernel at stride = plane size (8 294 400 B); this only asks whether a better distance exists at all.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// total = 3 * region bytes; block b owns chunk b of each of three regions that start `stride` apart inside a super-block of 3 * stride bytes
__global__ __launch_bounds__(512) void x3(float* __restrict__ db, long long stride16, long long chunks_per_region, long long nblocks) {
    const long long b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    if (b >= nblocks) return;
    const long long sup = b / chunks_per_region, c = b - sup * chunks_per_region;      // super-block, chunk inside each of its three regions
    const long long q0 = sup * 3 * stride16 + c * 512 + threadIdx.x;                   // in 16-byte units
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const long long q = q0 + k * stride16, base = q & ~((1ll << 26) - 1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + 4 * base, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, (unsigned)k}, rs, (int)(16 * (q - base)), 0, 19);
    }
}
__global__ __launch_bounds__(256) void x1(float* __restrict__ db, long long n4) {
    const long long i = (long long)blockIdx.y * gridDim.x * 256 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const long long base = i & ~((1ll << 26) - 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(db + 4 * base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, 4u}, rs, (int)(16 * (i - base)), 0, 19);
}
int main() {
    const long long total = 1024LL * 24883200;   // bytes
    float* dst; CK(hipMalloc(&dst, total + (64 << 20)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long strides[] = {8192, 65536, 262144, 1 << 20, 4 << 20, 8294400, 8 << 20, 32 << 20, 256 << 20, 1LL << 30, total / 3 / 8192 * 8192};
    struct R { long long stride; std::vector<float> ms; double frac; };
    std::vector<R> rs;
    for (long long s : strides) rs.push_back({s, {}, 1.0});
    std::vector<float> flat;
    for (int round = 0; round < 6; ++round) {
        { CK(hipEventRecord(e0, st)); const long long n4 = total / 16; hipLaunchKernelGGL(x1, dim3(65536, (unsigned)((n4 + 65536LL * 256 - 1) / (65536LL * 256))), dim3(256), 0, st, dst, n4);
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (round) flat.push_back(ms); }
        for (auto& r : rs) {
            const long long cpr = r.stride / 8192;                       // 8 KiB chunks per region
            const long long sups = total / (3 * r.stride);               // whole super-blocks only
            const long long blocks = sups * cpr;
            r.frac = (double)(sups * 3 * r.stride) / total;
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(x3, dim3(65536, (unsigned)((blocks + 65535) / 65536)), dim3(512), 0, st, dst, r.stride / 16, cpr, blocks);
            CK(hipGetLastError());
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (round) r.ms.push_back(ms);
        }
    }
    std::sort(flat.begin(), flat.end());
    printf("# 25.48 GB written, 5 rounds interleaved; GB/s scaled to the bytes a variant covers (whole super-blocks of 3 x stride)\n");
    printf("%-44s %9s %9s %7s\n", "variant", "med ms", "GB/s", "frac");
    printf("%-44s %9.3f %9.0f %7.3f\n", "one store per wave (flat fill)", flat[flat.size() / 2], total / flat[flat.size() / 2] / 1e6, total / flat[flat.size() / 2] / 1e6 / 8000);
    for (auto& r : rs) {
        std::sort(r.ms.begin(), r.ms.end());
        const float med = r.ms[r.ms.size() / 2];
        char name[96]; snprintf(name, sizeof name, "three stores per thread, %lld B apart%s", r.stride, r.stride == 8294400 ? " (= planes)" : "");
        // note: grid rounds up to 65536-block rows; extra blocks write past `covered` but inside the allocation only if blocks fit — keep exact:
        printf("%-44s %9.3f %9.0f %7.3f\n", name, med, total * r.frac / med / 1e6, total * r.frac / med / 1e6 / 8000);
    }
    return 0;
}
