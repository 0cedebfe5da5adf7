// Dev micro-benchmark (round 4, not shipped): calibrate rocprofv3's read counters on KNOWN byte counts per access shape
// (MI355X_MICROARCH.md "HBM": FETCH_SIZE x 2 is calibrated only for wide coalesced 16 B / lane reads —
// "other access widths: calibrate on a known byte count in your own access pattern").
// Every kernel reads from an 8 GiB buffer (32x the Infinity Cache), touches each address at most once per launch, and its
// distinct bytes / 32-B sectors / 64-B halves / 128-B lines are printed, to be set against FETCH_SIZE and the raw
// TCC_EA0_RDREQ{,_32B,_64B,_128B} request counts of the same launch.  One kernel symbol per shape.
//   wide16      16 B per lane, coalesced, every byte                     (the guide's calibrated shape)
//   dword4      4 B per lane, coalesced, every byte
//   px12        12 B per lane (three dwords of one f32x3 pixel), coalesced, every byte      (f32 HWC maps)
//   stride128   one dword per 128-B line     stride64: one dword per 64 B     stride32: one dword per 32 B
//   pair24_103  a 24-B tap pair (two f32x3 pixels) every 103.2 B on average — the C2 resize row pattern (8.57x decimation)
//   u8tap8_31   8 unaligned bytes at byte offset 31*i + 1 (u8 bilinear tap pair, overlapping lines)      (u8 gathers)
//   u8tap6_9    6 unaligned bytes at byte offset 9*i + 1 (RGB8 tap pair at 3x decimation)
//   lds16       16 B per lane through `buffer_load ... lds`-free staging: global -> LDS -> sum (LDS-staged quads of the u8 warps)
// A checksum is accumulated so nothing is optimised away.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint16_t u16_unaligned __attribute__((aligned(1)));

__device__ __forceinline__ void sink(uint32_t v, uint32_t* out) { if (v == 0x9E3779B9u) atomicAdd(out, 1u); }

__global__ __launch_bounds__(256) void wide16(const u32x4* __restrict__ p, long long n, uint32_t* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32x4 v = __builtin_nontemporal_load(p + i);
    sink(v.x ^ v.y ^ v.z ^ v.w, out);
}
__global__ __launch_bounds__(256) void dword4(const uint32_t* __restrict__ p, long long n, uint32_t* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    sink(p[i], out);
}
__global__ __launch_bounds__(256) void px12(const uint32_t* __restrict__ p, long long n, uint32_t* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    sink(p[3 * i] ^ p[3 * i + 1] ^ p[3 * i + 2], out);
}
template <int STRIDE>
__global__ __launch_bounds__(256) void strided(const uint8_t* __restrict__ p, long long n, uint32_t* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    sink(*reinterpret_cast<const uint32_t*>(p + i * STRIDE), out);
}
// lane i of a row reads pixels x0 = floor((i + 0.5) * 8.5714 - 0.5) and x0 + 1 of a 1920-pixel f32x3 row; rows are 4.82 apart
// (1080 -> 224): the source-row pattern of resize 1920x1080 -> 224x224.  One launch = `rows` output rows x 224 lanes.
__global__ __launch_bounds__(256) void pair24_103(const float* __restrict__ p, int rows, uint32_t* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int r = i / 224, x = i - r * 224;
    if (r >= rows) return;
    const int x0 = (int)(((float)x + 0.5f) * (1920.0f / 224.0f) - 0.5f);
    const long long row = (long long)r * 5;  // every 5th source row: no 128-B line is shared between two output rows
    const float* q = p + (row * 1920 + x0) * 3;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc ^= __builtin_bit_cast(uint32_t, q[k]);
    sink(acc, out);
}
template <int BYTES, int STEP>
__global__ __launch_bounds__(256) void u8tap(const uint8_t* __restrict__ p, long long n, uint32_t* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* q = p + i * STEP + 1;
    uint32_t acc = *reinterpret_cast<const u32_unaligned*>(q);
    if (BYTES == 8) acc ^= *reinterpret_cast<const u32_unaligned*>(q + 4);
    if (BYTES == 6) acc ^= *reinterpret_cast<const u16_unaligned*>(q + 4);
    sink(acc, out);
}
__global__ __launch_bounds__(256) void lds16(const u32x4* __restrict__ p, long long n, uint32_t* out) {
    __shared__ u32x4 tile[256];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    tile[threadIdx.x] = i < n ? p[i] : u32x4{0, 0, 0, 0};
    __syncthreads();
    const u32x4 v = tile[threadIdx.x ^ 37];
    sink(v.x ^ v.y ^ v.z ^ v.w, out);
}

int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : "";
    const size_t BYTES = 8ull << 30;
    uint8_t* buf; uint32_t* out;
    CK(hipMalloc(&buf, BYTES + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0x5A, BYTES + 4096)); CK(hipMemset(out, 0, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    struct V { std::string name; double bytes, s32, s64, l128; std::function<void()> run; };
    std::vector<V> vs;
    auto blocks = [](long long n) { return dim3((unsigned)((n + 255) / 256)); };
    {
        const long long n = BYTES / 16;
        vs.push_back({"wide16", (double)BYTES, BYTES / 32.0, BYTES / 64.0, BYTES / 128.0, [=] { hipLaunchKernelGGL(wide16, blocks(n), dim3(256), 0, st, (const u32x4*)buf, n, out); }});
        vs.push_back({"lds16", (double)BYTES, BYTES / 32.0, BYTES / 64.0, BYTES / 128.0, [=] { hipLaunchKernelGGL(lds16, blocks(n), dim3(256), 0, st, (const u32x4*)buf, n, out); }});
    }
    {
        const long long n = BYTES / 4;
        vs.push_back({"dword4", (double)BYTES, BYTES / 32.0, BYTES / 64.0, BYTES / 128.0, [=] { hipLaunchKernelGGL(dword4, blocks(n), dim3(256), 0, st, (const uint32_t*)buf, n, out); }});
    }
    {
        const long long n = BYTES / 12;
        vs.push_back({"px12", 12.0 * n, 12.0 * n / 32, 12.0 * n / 64, 12.0 * n / 128, [=] { hipLaunchKernelGGL(px12, blocks(n), dim3(256), 0, st, (const uint32_t*)buf, n, out); }});
    }
    {
        const long long n128 = BYTES / 128, n64 = BYTES / 64 / 2, n32 = BYTES / 32 / 4;  // stride64 covers half, stride32 a quarter of the buffer
        vs.push_back({"stride128", 4.0 * n128, (double)n128, (double)n128, (double)n128, [=] { hipLaunchKernelGGL(strided<128>, blocks(n128), dim3(256), 0, st, buf, n128, out); }});
        vs.push_back({"stride64", 4.0 * n64, (double)n64, (double)n64, n64 / 2.0, [=] { hipLaunchKernelGGL(strided<64>, blocks(n64), dim3(256), 0, st, buf, n64, out); }});
        vs.push_back({"stride32", 4.0 * n32, (double)n32, n32 / 2.0, n32 / 4.0, [=] { hipLaunchKernelGGL(strided<32>, blocks(n32), dim3(256), 0, st, buf, n32, out); }});
    }
    {
        const int rows = (int)(BYTES / (1920ull * 12 * 5));
        // host-side count of the sectors one output row touches
        double s32 = 0, s64 = 0, l128 = 0; long long last32 = -1, last64 = -1, last128 = -1;
        for (int x = 0; x < 224; ++x) {
            const int x0 = (int)(((float)x + 0.5f) * (1920.0f / 224.0f) - 0.5f);
            for (long long b = (long long)x0 * 12; b < (long long)x0 * 12 + 24; ++b) {
                if (b / 32 != last32) { last32 = b / 32; ++s32; }
                if (b / 64 != last64) { last64 = b / 64; ++s64; }
                if (b / 128 != last128) { last128 = b / 128; ++l128; }
            }
        }
        // rows start at multiples of 1920*12*5 = 115200 B = 900 lines: line-aligned, so the per-row count holds for every row
        vs.push_back({"pair24_103", 24.0 * 224 * rows, s32 * rows, s64 * rows, l128 * rows, [=] { hipLaunchKernelGGL(pair24_103, blocks(224LL * rows), dim3(256), 0, st, (const float*)buf, rows, out); }});
    }
    {
        const long long n31 = (BYTES - 64) / 31, n9 = (BYTES / 4 - 64) / 9;
        // 8 bytes every 31: distinct bytes 8n; every sector / line of the span is touched (31 < 32)
        vs.push_back({"u8tap8_31", 8.0 * n31, 31.0 * n31 / 32, 31.0 * n31 / 64, 31.0 * n31 / 128, [=] { hipLaunchKernelGGL((u8tap<8, 31>), blocks(n31), dim3(256), 0, st, buf, n31, out); }});
        vs.push_back({"u8tap6_9", 6.0 * n9, 9.0 * n9 / 32, 9.0 * n9 / 64, 9.0 * n9 / 128, [=] { hipLaunchKernelGGL((u8tap<6, 9>), blocks(n9), dim3(256), 0, st, buf, n9, out); }});
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# known-byte read shapes over an 8 GiB buffer; 2 launches each (the counters are per launch)\n");
    printf("%-12s %14s %14s %14s %14s %9s\n", "shape", "distinct_B", "x32B_sectors_B", "x64B_halves_B", "x128B_lines_B", "ms");
    for (auto& v : vs) {
        if (*only && !strstr(only, v.name.c_str())) continue;
        float ms = 0;
        for (int r = 0; r < 2; ++r) {
            CK(hipEventRecord(e0, st)); v.run(); CK(hipGetLastError()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%-12s %14.0f %14.0f %14.0f %14.0f %9.3f\n", v.name.c_str(), v.bytes, v.s32 * 32, v.s64 * 64, v.l128 * 128, ms);
    }
    return 0;
}
