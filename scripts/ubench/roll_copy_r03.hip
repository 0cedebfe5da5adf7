// Dev micro-benchmark (round 3, not shipped): what the ACCESS PATTERN of the rolling-wave RGB8 kernels (u8 blur, dilate / erode,
// pyramids) costs with no arithmetic at all — a copy of 256 RGB8 4K images in the same shape: a wave walks down a strip, a lane
// loads 12 bytes per row with K rows in flight, 62 (or 60) of 64 lanes store.  Variants change one thing at a time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef uint32_t u32u __attribute__((aligned(1)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct A { const uint8_t* src; uint8_t* dst; int w, h, th; long long fs; };

// lane owns 4 px (12 B); wave covers WPX output px (+ halo quads either side); MODE 0: direct 12-B stores; 1: LDS -> 16-B chunks (WPX = 240)
template <int K, int WPX, int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void roll12(A a) {
    __shared__ __attribute__((aligned(16))) uint32_t xp[WAVES][2][192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p0 = ((int)blockIdx.x * WAVES + wv) * WPX;
    if (p0 >= a.w) return;
    const int y0 = blockIdx.y * a.th;
    const uint8_t* src = a.src + (long long)blockIdx.z * a.fs;
    uint8_t* dst = a.dst + (long long)blockIdx.z * a.fs;
    constexpr int HL = (256 - WPX) / 8;   // halo lanes either side
    const int p = p0 - 4 * HL + 4 * lane, pc = min(max(p, 0), a.w - 4);
    const bool writer = lane >= HL && lane < 64 - HL && p < a.w;
    const int rowb = a.w * 3, nrows = min(a.th, a.h - y0) + K - 1;
    int pf = y0 - K / 2;
    uint32_t q[K][3];
    uint32_t acc = 0;
    auto prefetch = [&](uint32_t (&d)[3]) {
        if (MODE == 3) { d[0] = pf; d[1] = lane; d[2] = 7; ++pf; return; }   // store-only
        const uint8_t* rp = src + (long long)min(max(pf, 0), a.h - 1) * rowb + 3 * pc;
        d[0] = *(const u32u*)rp; d[1] = *(const u32u*)(rp + 4); d[2] = *(const u32u*)(rp + 8); ++pf;
    };
#pragma unroll
    for (int i = 0; i < K; ++i) prefetch(q[i]);
    long long off = (long long)(y0 - (K - 1)) * rowb + 3 * p;
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const int r = rb + s;
            uint32_t d0 = q[s][0], d1 = q[s][1], d2 = q[s][2];
            prefetch(q[s]);
            // a token of arithmetic so that the row cannot be forwarded untouched: rotate the lane's neighbour in
            d0 ^= (uint32_t)__shfl_up((int)d1, 1) & 0u;
            if (MODE == 2) { acc ^= d0 ^ d1 ^ d2; }   // load-only
            else if (r >= K - 1 && r < nrows) {
                if (MODE == 4) {   // non-temporal stores
                    if (writer) { uint8_t* o = dst + off; __builtin_nontemporal_store(d0, (u32u*)o); __builtin_nontemporal_store(d1, (u32u*)(o + 4)); __builtin_nontemporal_store(d2, (u32u*)(o + 8)); }
                } else if (MODE == 0 || MODE == 3) {
                    if (writer) { uint8_t* o = dst + off; *(u32u*)o = d0; *(u32u*)(o + 4) = d1; *(u32u*)(o + 8) = d2; }
                } else {
                    uint32_t* x = xp[wv][s & 1];
                    if (lane >= HL && lane < 64 - HL) { x[3 * (lane - HL)] = d0; x[3 * (lane - HL) + 1] = d1; x[3 * (lane - HL) + 2] = d2; }
                    // 60 lanes x 12 B = 720 B = 45 chunks of 16 B
                    const long long ob = (long long)(y0 - (K - 1) + r) * rowb + 3 * p0 + 16 * lane;
                    if (lane < 45 && 3 * p0 + 16 * lane + 16 <= rowb) *(u32x4*)(dst + ob) = *(const u32x4*)(x + 4 * lane);
                }
            }
            off += rowb;
        }
    }
    if (MODE == 2 && acc == 0x12345u) dst[p] = 1;
}
// lane owns 16 B, no halo, K rows in flight
template <int K, int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES) void roll16(A a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rowb = a.w * 3;
    const int b0 = (((int)blockIdx.x * WAVES + wv) * 64 + lane) * 16;
    if (b0 >= rowb) return;
    const int y0 = blockIdx.y * a.th, nrows = min(a.th, a.h - y0);
    const uint8_t* src = a.src + (long long)blockIdx.z * a.fs + (long long)y0 * rowb + b0;
    uint8_t* dst = a.dst + (long long)blockIdx.z * a.fs + (long long)y0 * rowb + b0;
    u32x4 q[K];
    uint32_t acc = 0;
    int pf = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) { q[i] = *(const u32x4*)(src + (long long)min(pf, nrows - 1) * rowb); ++pf; }
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const u32x4 d = q[s];
            if (MODE != 3) q[s] = *(const u32x4*)(src + (long long)min(pf, nrows - 1) * rowb);
            ++pf;
            if (MODE == 2) acc ^= d.x ^ d.y ^ d.z ^ d.w;
            else if (rb + s < nrows) *(u32x4*)(dst + (long long)(rb + s) * rowb) = d;
        }
    }
    if (MODE == 2 && acc == 0x12345u) dst[0] = 1;
}
// the staged gather's store shape: a 512-thread block owns a TW x TH pixel tile of NB consecutive images, a thread 4 pixels (12 bytes),
// threads row-major in the tile: a wave's store instruction covers 256 / TW row segments of 3 TW bytes
template <int TW, int NB>
__global__ __launch_bounds__(512) void tile_store(A a, int copy) {
    constexpr int TH = 2048 / TW, TXN = TW / 4;
    const int tx = threadIdx.x % TXN, ty = threadIdx.x / TXN;
    const int x4 = blockIdx.x * TW + 4 * tx, y = blockIdx.y * TH + ty;
    if (x4 >= a.w || y >= a.h) return;
    const long long off = (long long)y * a.w * 3 + 3 * x4;
    for (int b = 0; b < NB; ++b) {
        const long long f = ((long long)blockIdx.z * NB + b) * a.fs + off;
        uint32_t d0 = threadIdx.x, d1 = b, d2 = 7;
        if (copy) { d0 = *(const u32u*)(a.src + f); d1 = *(const u32u*)(a.src + f + 4); d2 = *(const u32u*)(a.src + f + 8); }
        *(u32u*)(a.dst + f) = d0; *(u32u*)(a.dst + f + 4) = d1; *(u32u*)(a.dst + f + 8) = d2;
    }
}
// a 64-px-wide tile PAIR per block: ORDER 0 = per image, left tile then right tile; 1 = all images of the left tile, then of the right
template <int ORDER>
__global__ __launch_bounds__(512) void tile_pair_store(A a) {
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    const int y = blockIdx.y * 32 + ty;
    if (y >= a.h) return;
    for (int i = 0; i < 16; ++i) {
        const int b = ORDER == 0 ? i >> 1 : i & 7, half = ORDER == 0 ? i & 1 : i >> 3;
        const int x4 = blockIdx.x * 128 + half * 64 + 4 * tx;
        if (x4 >= a.w) continue;
        const long long f = ((long long)blockIdx.z * 8 + b) * a.fs + (long long)y * a.w * 3 + 3 * x4;
        *(u32u*)(a.dst + f) = threadIdx.x; *(u32u*)(a.dst + f + 4) = b; *(u32u*)(a.dst + f + 8) = 7;
    }
}
__global__ __launch_bounds__(256) void flat16(const u32x4* __restrict__ s, u32x4* __restrict__ d, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d[i] = s[i];
}

int main(int argc, char** argv) {
    const int W = 3840, H = 2160, N = argc > 1 ? atoi(argv[1]) : 256, ROUNDS = 7;
    const size_t fb = (size_t)W * H * 3;
    uint8_t *src, *dst;
    CK(hipMalloc(&src, fb * N)); CK(hipMalloc(&dst, fb * N));
    CK(hipMemset(src, 0x5a, fb * N)); CK(hipMemset(dst, 0, fb * N));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { std::string name; std::function<void()> run; std::vector<float> ms; };
    std::vector<V> vs;
    auto cd = [](int a, int b) { return (a + b - 1) / b; };
#define R12(K, WPX, MODE, WAVES, TH, NAME) vs.push_back({NAME, [&] { A a{src, dst, W, H, TH, (long long)fb}; hipLaunchKernelGGL((roll12<K, WPX, MODE, WAVES>), dim3(cd(W, WPX * WAVES), cd(H, TH), N), dim3(64 * WAVES), 0, st, a); }, {}});
    R12(5, 248, 0, 4, 360, "roll12 K5 248px 4 waves th360 direct 12-B stores (= dilate 5x5)")
    R12(3, 248, 0, 4, 360, "roll12 K3")
    R12(7, 248, 0, 4, 360, "roll12 K7")
    R12(5, 248, 0, 4, 135, "roll12 K5 th135")
    R12(5, 248, 0, 4, 1080, "roll12 K5 th1080")
    R12(5, 248, 0, 1, 360, "roll12 K5 1 wave per block")
    R12(5, 248, 0, 2, 360, "roll12 K5 2 waves per block")
    R12(5, 240, 0, 4, 360, "roll12 K5 240px direct")
    R12(5, 240, 1, 4, 360, "roll12 K5 240px LDS -> 16-B chunk stores")
    R12(5, 240, 1, 4, 135, "roll12 K5 240px LDS -> 16-B chunk stores th135")
    R12(5, 248, 0, 8, 360, "roll12 K5 248px 8 waves per block")
    R12(5, 256, 0, 4, 360, "roll12 K5 256px no halo (768 B per wave row)")
    R12(5, 256, 4, 4, 360, "roll12 K5 256px no halo, NON-TEMPORAL stores")
    R12(5, 256, 4, 4, 135, "roll12 K5 256px no halo, NON-TEMPORAL stores th135")
    R12(5, 256, 0, 4, 135, "roll12 K5 256px no halo th135")
    R12(5, 256, 0, 8, 360, "roll12 K5 256px no halo 8 waves per block")
    R12(5, 256, 0, 2, 360, "roll12 K5 256px no halo 2 waves per block")
    R12(3, 256, 0, 4, 360, "roll12 K3 256px no halo")
    R12(7, 256, 0, 4, 360, "roll12 K7 256px no halo")
    R12(5, 248, 2, 4, 360, "roll12 K5 248px LOAD only")
    R12(5, 256, 2, 4, 360, "roll12 K5 256px no halo LOAD only")
    R12(5, 248, 3, 4, 360, "roll12 K5 248px STORE only")
    R12(5, 256, 3, 4, 360, "roll12 K5 256px no halo STORE only")
    R12(5, 240, 3, 4, 360, "roll12 K5 240px STORE only")
#define R16(K, WAVES, TH, MODE, NAME) vs.push_back({NAME, [&] { A a{src, dst, W, H, TH, (long long)fb}; hipLaunchKernelGGL((roll16<K, WAVES, MODE>), dim3(cd(W * 3, 1024 * WAVES), cd(H, TH), N), dim3(64 * WAVES), 0, st, a); }, {}});
    R16(5, 4, 360, 0, "roll16 K5 4 waves th360 (16 B per lane, no halo)")
    R16(5, 4, 135, 0, "roll16 K5 th135")
    R16(3, 4, 360, 0, "roll16 K3")
    R16(5, 8, 360, 0, "roll16 K5 8 waves per block")
    R16(5, 4, 360, 2, "roll16 K5 LOAD only")
    R16(5, 4, 360, 3, "roll16 K5 STORE only")
#define TS(TW, COPY, NAME) vs.push_back({NAME, [&] { A a{src, dst, W, H, 0, (long long)fb}; hipLaunchKernelGGL((tile_store<TW, 8>), dim3(cd(W, TW), cd(H, 2048 / TW), N / 8), dim3(512), 0, st, a, COPY); }, {}});
    TS(64, 0, "tile 64 x 32 px x 8 images STORE only (the staged gather's shape: 192-byte segments)")
    TS(128, 0, "tile 128 x 16 STORE only (384-byte segments)")
    TS(256, 0, "tile 256 x 8 STORE only (768-byte segments)")
    TS(512, 0, "tile 512 x 4 STORE only")
    vs.push_back({"tile pair 2 x (64 x 32) per block, per image left then right, STORE only", [&] { A a{src, dst, W, H, 0, (long long)fb}; hipLaunchKernelGGL((tile_pair_store<0>), dim3(cd(W, 128), cd(H, 32), N / 8), dim3(512), 0, st, a); }, {}});
    vs.push_back({"tile pair 2 x (64 x 32) per block, 8 images left then 8 images right, STORE only", [&] { A a{src, dst, W, H, 0, (long long)fb}; hipLaunchKernelGGL((tile_pair_store<1>), dim3(cd(W, 128), cd(H, 32), N / 8), dim3(512), 0, st, a); }, {}});
    TS(64, 1, "tile 64 x 32 copy")
    TS(128, 1, "tile 128 x 16 copy")
    TS(256, 1, "tile 256 x 8 copy")
    vs.push_back({"flat16 grid-stride copy", [&] { hipLaunchKernelGGL(flat16, dim3(256 * 32), dim3(256), 0, st, (const u32x4*)src, (u32x4*)dst, (long long)(fb * N / 16)); }, {}});
    for (int r = 0; r < ROUNDS + 1; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, st)); v.run(); CK(hipGetLastError()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    printf("# %d RGB8 images of %dx%d, copy (R + W = %.2f GB), %d rounds interleaved\n", N, W, H, 2.0 * fb * N / 1e9, ROUNDS);
    printf("%-72s %9s %9s %9s\n", "variant", "med ms", "min ms", "GB/s@med");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-72s %9.3f %9.3f %9.0f\n", v.name.c_str(), v.ms[v.ms.size() / 2], v.ms[0], 2.0 * fb * N / 1e6 / v.ms[v.ms.size() / 2]);
    }
    // spot check of the LDS variant against the source
    std::vector<uint8_t> h(fb); CK(hipMemset(dst, 0, fb)); vs[8].run(); CK(hipStreamSynchronize(st)); CK(hipMemcpy(h.data(), dst, fb, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < fb; ++i) bad += h[i] != 0x5a;
    printf("# LDS-chunk variant: %zu bytes differ from the source\n", bad);
    return 0;
}
