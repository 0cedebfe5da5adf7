// Round 6 addition to strip_copy_r05.hip: strip_copy_sync — the strip walk with the waves of one BAND (the 45 waves that walk down the same
// rows of one image side by side) kept within SLACK epochs of EVERY rows of each other through a counter in global memory, so that a band
// moves through memory as one contiguous front instead of 45 fronts that drift apart.
// Dev micro-benchmark (round 5, not shipped): the C4 gaussian takes the same time with its arithmetic and its LDS traffic removed
// (profiles/r05l): what bounds it is the ACCESS PATTERN of the rolling kernels — a wave walks down a strip, reads and writes LW floats
// per lane per row, K rows of loads in flight — which moves 51 GB at 0.63-0.72 of 8 TB/s where a flat 1R + 1W map of the same bytes
// reaches 0.78.  This is that pattern as a pure copy (256 f32x3 4K images), one knob at a time.
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
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWLEN = 3840 * 3, ROWS = 2160;          // floats per row, rows
constexpr long long IMG = (long long)ROWLEN * ROWS;    // floats per image

struct P { const float* src; float* dst; int th, tiles_x, strips, images, order; unsigned* cnt; };

// block id -> (tile_x, strip, image).  order 0: x fastest, then strip, then image.  order 1: the same list dealt to the 8 XCDs in
// contiguous eighths (block b runs on XCD b % 8: XCD k walks ids [k * total / 8, (k + 1) * total / 8)), as kh_common.h::xcd_tile does.
// order 2: strip fastest, then x, then image (vertically adjacent strips launch together).  order 3: image fastest.
__device__ __forceinline__ bool decode(const P& p, unsigned b, int& tx, int& ty, int& tz) {
    const unsigned total = (unsigned)p.tiles_x * p.strips * p.images;
    unsigned id = b;
    if (p.order == 1) {
        const unsigned per = (total + 7) / 8, xcd = b % 8, slot = b / 8;
        if (slot >= per) return false;
        id = xcd * per + slot;
    }
    if (id >= total) return false;
    if (p.order == 2) { ty = id % p.strips; id /= p.strips; tx = id % p.tiles_x; tz = id / p.tiles_x; return true; }
    if (p.order == 3) { tz = id % p.images; id /= p.images; tx = id % p.tiles_x; ty = id / p.tiles_x; return true; }
    tx = id % p.tiles_x; id /= p.tiles_x; ty = id % p.strips; tz = id / p.strips;
    return true;
}

// LW4: float4 per lane per row (1 or 2); K rows of loads in flight; WAVES per block; ST 1 = write-through nt buffer stores, 0 = plain
template <int LW4, int K, int WAVES, int ST>
__global__ __launch_bounds__(64 * WAVES) void strip_copy(P p) {
    extern __shared__ float pad_[];
    if (p.th < 0) pad_[threadIdx.x] = 1.0f;
    int tx, ty, tz;
    if (!decode(p, blockIdx.x, tx, ty, tz)) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gx = (tx * WAVES + wv) * (256 * LW4) + 4 * LW4 * lane;
    if (gx >= ROWLEN) return;
    const int y0 = ty * p.th, nrows = min(p.th, ROWS - y0);
    const float* src = p.src + (long long)tz * IMG + gx;
    float* dst = p.dst + (long long)tz * IMG + gx;
    f32x4 q[K][LW4];
    int pf = y0;
    auto prefetch = [&](f32x4 (&d)[LW4]) {
        const int r = min(pf, ROWS - 1);
#pragma unroll
        for (int j = 0; j < LW4; ++j) d[j] = *reinterpret_cast<const f32x4*>(src + (long long)r * ROWLEN + 4 * j);
        ++pf;
    };
#pragma unroll
    for (int i = 0; i < K; ++i) prefetch(q[i]);
    // wave-uniform window over this strip's rows of the image (a per-lane base would be a 64-trip waterfall loop)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.dst + (long long)tz * IMG + (long long)y0 * ROWLEN, 0, (int)min((long long)nrows * ROWLEN * 4, 0x7fffffffll), 0x00020000);
    int off = gx * 4;
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            if (rb + i < nrows) {
                f32x4 v[LW4];
#pragma unroll
                for (int j = 0; j < LW4; ++j) v[j] = q[i][j];
                prefetch(q[i]);
#pragma unroll
                for (int j = 0; j < LW4; ++j) {
                    if (ST) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j]), rs, off + 16 * j, 0, 19);
                    else *reinterpret_cast<f32x4*>(dst + (long long)(y0 + rb + i) * ROWLEN + 4 * j) = v[j];
                }
            }
            off += ROWLEN * 4;
        }
    }
}


template <int K, int EVERY, int SLACK>
__global__ __launch_bounds__(256) void strip_copy_sync(P p) {
    int tx, ty, tz;
    if (!decode(p, blockIdx.x, tx, ty, tz)) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gx = (tx * 4 + wv) * 256 + 4 * lane;
    if (gx >= ROWLEN) return;
    constexpr unsigned NW = ROWLEN / 256;   // waves per band (45)
    unsigned* cnt = p.cnt + (tz * p.strips + ty);
    const int y0 = ty * p.th, nrows = min(p.th, ROWS - y0);
    const float* src = p.src + (long long)tz * IMG + gx;
    f32x4 q[K];
    int pf = y0;
    auto prefetch = [&](f32x4& d) { d = *reinterpret_cast<const f32x4*>(src + (long long)min(pf, ROWS - 1) * ROWLEN); ++pf; };
#pragma unroll
    for (int i = 0; i < K; ++i) prefetch(q[i]);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.dst + (long long)tz * IMG + (long long)y0 * ROWLEN, 0, (int)min((long long)nrows * ROWLEN * 4, 0x7fffffffll), 0x00020000);
    int off = gx * 4;
    unsigned epoch = 0;
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            if (rb + i < nrows) {
                const f32x4 v = q[i];
                prefetch(q[i]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 19);
            }
            off += ROWLEN * 4;
        }
        if ((rb / K + 1) % EVERY == 0) {   // an epoch = EVERY trips of K rows
            ++epoch;
            if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (epoch > (unsigned)SLACK) {
                const unsigned need = NW * (epoch - SLACK);
                int spins = 0;
                while ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need && ++spins < 20000) __builtin_amdgcn_s_sleep(8);
            }
        }
    }
}

__global__ __launch_bounds__(256) void flat_copy(const float* __restrict__ src, float* __restrict__ dst, long long n4) {
    const long long i = (long long)blockIdx.y * gridDim.x * 256 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    const long long base = i & ~((1ll << 26) - 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst + 4 * base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)(16 * (i - base)), 0, 19);
}

// flat copies with S chunks of 1 KiB per WAVE.  MODE 0: chunk ids w + k * G (grid-stride); 1: S adjacent chunks per wave; 2: ONE chunk
// per wave at a permuted position (short-lived waves, scattered 1 KiB accesses).  PIPE 1: all S loads first, then S stores; 0: one at a time.
template <int S, int MODE, int PIPE>
__global__ __launch_bounds__(256) void flat_multi(const float* __restrict__ src, float* __restrict__ dst, long long nchunks, int group = 1) {
    const long long w = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);   // wave id
    const int lane = threadIdx.x & 63;
    const long long G = (nchunks + S - 1) / S;   // waves in the grid (rounded)
    if (w >= G) return;
    long long id[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        if (MODE == 0) id[k] = w + k * G;
        else if (MODE == 1) id[k] = w * S + k;
        else if (MODE == 3) id[k] = (w / group) * ((long long)group * S) + (long long)k * group + (w % group);   // `group` consecutive waves stream one contiguous region together
        else { const long long h = (w * 2654435761ll) % nchunks; id[k] = h < 0 ? h + nchunks : h; }   // odd multiplier: a permutation when nchunks is a power of two; close enough otherwise (a few chunks written twice)
        if (id[k] >= nchunks) id[k] = nchunks - 1;
    }
    f32x4 v[S];
    if (PIPE) {
#pragma unroll
        for (int k = 0; k < S; ++k) v[k] = reinterpret_cast<const f32x4*>(src)[id[k] * 64 + lane];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const long long e = id[k] * 64 + lane, base = e & ~((1ll << 26) - 1);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst + 4 * base, 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[k]), rs, (int)(16 * (e - base)), 0, 19);
        }
    } else {
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const long long e = id[k] * 64 + lane;
            const f32x4 t = reinterpret_cast<const f32x4*>(src)[e];
            reinterpret_cast<f32x4*>(dst)[e] = t;
            __builtin_amdgcn_s_waitcnt(0x0f70);
        }
    }
}

// the strip walk with its stores held back: SB output rows are kept in registers and written back to back (the 45 waves of a strip row then
// write SB rows = SB x 45 KiB of consecutive addresses together instead of one row per step)
template <int K, int SB>
__global__ __launch_bounds__(256) void strip_copy_sb(P p) {
    int tx, ty, tz;
    if (!decode(p, blockIdx.x, tx, ty, tz)) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gx = (tx * 4 + wv) * 256 + 4 * lane;
    if (gx >= ROWLEN) return;
    const int y0 = ty * p.th, nrows = min(p.th, ROWS - y0);
    const float* src = p.src + (long long)tz * IMG + gx;
    f32x4 q[K];
    int pf = y0;
    auto prefetch = [&](f32x4& d) { d = *reinterpret_cast<const f32x4*>(src + (long long)min(pf, ROWS - 1) * ROWLEN); ++pf; };
#pragma unroll
    for (int i = 0; i < K; ++i) prefetch(q[i]);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.dst + (long long)tz * IMG + (long long)y0 * ROWLEN, 0, (int)min((long long)nrows * ROWLEN * 4, 0x7fffffffll), 0x00020000);
    int off = gx * 4;
    static_assert((K * SB) % K == 0, "");
    for (int rb = 0; rb < nrows; rb += K * SB) {
#pragma unroll
        for (int g = 0; g < K; ++g) {          // K groups of SB rows per trip (so the prefetch ring index stays a compile-time constant)
            f32x4 hold[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                constexpr int dummy = 0; (void)dummy;
                const int i = (g * SB + j) % K;
                hold[j] = q[i];
                prefetch(q[i]);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                if (rb + g * SB + j < nrows) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hold[j]), rs, off, 0, 19);
                off += ROWLEN * 4;
            }
        }
    }
}

// the strip walk with ROW PHASES: a block is 4 column groups x RP waves; wave (c, j) copies rows y0 + RP t + j of column group c, so that the
// waves of a strip move RP consecutive image rows (RP x 45 KiB of consecutive addresses) per step instead of one (best case of a filter whose
// waves would share their horizontal results through LDS: no LDS and no synchronisation here)
template <int K, int RP>
__global__ __launch_bounds__(64 * 4 * RP) void strip_copy_rp(P p) {
    int tx, ty, tz;
    if (!decode(p, blockIdx.x, tx, ty, tz)) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = wv % 4, j = wv / 4;
    const int gx = (tx * 4 + c) * 256 + 4 * lane;
    if (gx >= ROWLEN) return;
    const int y0 = ty * p.th, nrows = min(p.th, ROWS - y0);
    const float* src = p.src + (long long)tz * IMG + gx;
    f32x4 q[K];
    int pf = y0 + j;
    auto prefetch = [&](f32x4& d) { d = *reinterpret_cast<const f32x4*>(src + (long long)min(pf, ROWS - 1) * ROWLEN); pf += RP; };
#pragma unroll
    for (int i = 0; i < K; ++i) prefetch(q[i]);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.dst + (long long)tz * IMG + (long long)y0 * ROWLEN, 0, (int)min((long long)nrows * ROWLEN * 4, 0x7fffffffll), 0x00020000);
    int off = gx * 4 + j * ROWLEN * 4;
    for (int rb = j; rb < nrows; rb += K * RP) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const f32x4 v = q[i];
            prefetch(q[i]);
            if (rb + i * RP < nrows) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 19);
            off += RP * ROWLEN * 4;
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 256, ROUNDS = argc > 2 ? atoi(argv[2]) : 5;
    const char* only = argc > 3 ? argv[3] : "";
    float *src, *dst;
    CK(hipMalloc(&src, IMG * N * 4)); CK(hipMalloc(&dst, IMG * N * 4));
    CK(hipMemset(src, 0x3c, IMG * N * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned* cnt; CK(hipMalloc(&cnt, 1 << 20)); CK(hipMemset(cnt, 0, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { std::string name; std::function<void()> run; std::vector<float> ms; };
    std::vector<V> vs;
    const long long n4 = IMG * N / 4;
    vs.push_back({"flat copy 16 B / lane (the 0.78 row)", [&] { hipLaunchKernelGGL(flat_copy, dim3(65536, (unsigned)((n4 + 65536LL * 256 - 1) / (65536LL * 256))), dim3(256), 0, st, src, dst, n4); }, {}});
    const long long nchunks = n4 / 64;
#define FM(S, MODE, PIPE, NAME) vs.push_back({NAME, [&] { const long long G = (nchunks + S - 1) / S, blocks = (G + 3) / 4; hipLaunchKernelGGL((flat_multi<S, MODE, PIPE>), dim3(65536, (unsigned)((blocks + 65535) / 65536)), dim3(256), 0, st, src, dst, nchunks); }, {}});
    FM(1, 0, 1, "flat 1 chunk per wave (in order)")
    FM(1, 2, 1, "flat 1 chunk per wave, permuted positions (scattered 1 KiB)")
    FM(2, 0, 1, "flat 2 chunks per wave, grid-stride, loads first")
    FM(2, 1, 1, "flat 2 chunks per wave, adjacent, loads first")
    FM(4, 0, 1, "flat 4 chunks per wave, grid-stride, loads first")
    FM(8, 0, 1, "flat 8 chunks per wave, grid-stride, loads first")
    FM(8, 1, 1, "flat 8 chunks per wave, adjacent, loads first")
    FM(8, 0, 0, "flat 8 chunks per wave, grid-stride, one at a time")
    FM(8, 1, 0, "flat 8 chunks per wave, adjacent, one at a time")
    FM(32, 0, 1, "flat 32 chunks per wave, grid-stride, loads first")
    for (int g : {4, 16, 45, 128, 512, 2048, 8192, 32768}) {
        vs.push_back({"flat 8 chunks per wave, groups of " + std::to_string(g) + " waves stream together (" + std::to_string(g) + " KiB per step)",
                      [&, g] { const long long G = (nchunks + 7) / 8, blocks = (G + 3) / 4; hipLaunchKernelGGL((flat_multi<8, 3, 1>), dim3(65536, (unsigned)((blocks + 65535) / 65536)), dim3(256), 0, st, src, dst, nchunks, g); }, {}});
    }
    auto add = [&](const std::string& name, auto kernel, int lw4, int waves, int th, int order, int lds) {
        P p{src, dst, th, (ROWLEN + 256 * lw4 * waves - 1) / (256 * lw4 * waves), (ROWS + th - 1) / th, N, order, cnt};
        unsigned total = (unsigned)p.tiles_x * p.strips * p.images;
        if (order == 1) total = (total + 7) / 8 * 8;
        vs.push_back({name, [=] { hipLaunchKernelGGL(kernel, dim3(total), dim3(64 * waves), lds, st, p); }, {}});
    };
    add("strip LW4 K7 4w th360 xcd-eighth (= production shape)", strip_copy<1, 7, 4, 1>, 1, 4, 360, 1, 0);
    add("strip LW4 K7 4w th360 linear order", strip_copy<1, 7, 4, 1>, 1, 4, 360, 0, 0);
    add("strip LW4 K7 4w th360 strip-fastest order", strip_copy<1, 7, 4, 1>, 1, 4, 360, 2, 0);
    add("strip LW4 K7 4w th360 image-fastest order", strip_copy<1, 7, 4, 1>, 1, 4, 360, 3, 0);
    add("strip LW4 K7 4w th2160 xcd-eighth", strip_copy<1, 7, 4, 1>, 1, 4, 2160, 1, 0);
    add("strip LW4 K7 4w th90 xcd-eighth", strip_copy<1, 7, 4, 1>, 1, 4, 90, 1, 0);
    add("strip LW4 K7 4w th360 xcd-eighth plain stores", strip_copy<1, 7, 4, 0>, 1, 4, 360, 1, 0);
    add("strip LW4 K3 4w th360 xcd-eighth", strip_copy<1, 3, 4, 1>, 1, 4, 360, 1, 0);
    add("strip LW4 K2 4w th360 xcd-eighth", strip_copy<1, 2, 4, 1>, 1, 4, 360, 1, 0);
    add("strip LW4 K1 4w th360 xcd-eighth", strip_copy<1, 1, 4, 1>, 1, 4, 360, 1, 0);
    add("strip LW4 K4 4w th360 xcd-eighth", strip_copy<1, 4, 4, 1>, 1, 4, 360, 1, 0);
    add("strip LW4 K7 1w th360 xcd-eighth", strip_copy<1, 7, 1, 1>, 1, 1, 360, 1, 0);
    add("strip LW4 K7 8w th360 xcd-eighth", strip_copy<1, 7, 8, 1>, 1, 8, 360, 1, 0);
    add("strip LW4 K7 12w th360 xcd-eighth (whole row per block)", strip_copy<1, 7, 12, 1>, 1, 12, 360, 1, 0);
    add("strip LW8 K7 4w th360 xcd-eighth", strip_copy<2, 7, 4, 1>, 2, 4, 360, 1, 0);
    add("strip LW8 K4 4w th360 xcd-eighth", strip_copy<2, 4, 4, 1>, 2, 4, 360, 1, 0);
    add("strip LW8 K4 6w th360 xcd-eighth (whole row per block)", strip_copy<2, 4, 6, 1>, 2, 6, 360, 1, 0);
    add("strip LW4 K7 4w th360 xcd-eighth occupancy 4 blocks/CU", strip_copy<1, 7, 4, 1>, 1, 4, 360, 1, 40 * 1024);
    add("strip LW4 K7 4w th360 xcd-eighth occupancy 2 blocks/CU", strip_copy<1, 7, 4, 1>, 1, 4, 360, 1, 64 * 1024);
    add("strip LW4 K7 4w th360 xcd-eighth occupancy 1 block/CU", strip_copy<1, 7, 4, 1>, 1, 4, 360, 1, 100 * 1024);
#define SBV(K, SB, NAME) { P p{src, dst, 360, (ROWLEN + 1023) / 1024, (ROWS + 359) / 360, N, 1, cnt}; unsigned total = ((unsigned)p.tiles_x * p.strips * p.images + 7) / 8 * 8; \
        vs.push_back({NAME, [=] { hipLaunchKernelGGL((strip_copy_sb<K, SB>), dim3(total), dim3(256), 0, st, p); }, {}}); }
    SBV(7, 1, "stores held: K7 SB1 (control)") SBV(7, 2, "stores held: K7 SB2") SBV(7, 3, "stores held: K7 SB3") SBV(7, 4, "stores held: K7 SB4") SBV(7, 8, "stores held: K7 SB8")
    SBV(4, 4, "stores held: K4 SB4") SBV(8, 8, "stores held: K8 SB8") SBV(2, 16, "stores held: K2 SB16")
#define RPV(K, RP, NAME) { P p{src, dst, 360, (ROWLEN + 1023) / 1024, (ROWS + 359) / 360, N, 1, cnt}; unsigned total = ((unsigned)p.tiles_x * p.strips * p.images + 7) / 8 * 8; \
        vs.push_back({NAME, [=] { hipLaunchKernelGGL((strip_copy_rp<K, RP>), dim3(total), dim3(64 * 4 * RP), 0, st, p); }, {}}); }
    RPV(7, 1, "row phases: K7 RP1 (control, 256 thr)") RPV(7, 2, "row phases: K7 RP2 (512 thr)") RPV(7, 3, "row phases: K7 RP3 (768 thr)") RPV(7, 4, "row phases: K7 RP4 (1024 thr)")
    RPV(4, 3, "row phases: K4 RP3") RPV(3, 4, "row phases: K3 RP4") RPV(2, 4, "row phases: K2 RP4")
#define SYV(K, EV, SL, TH, NAME) { P p{src, dst, TH, (ROWLEN + 1023) / 1024, (ROWS + TH - 1) / TH, N, 1, cnt}; unsigned total = ((unsigned)p.tiles_x * p.strips * p.images + 7) / 8 * 8; \
        vs.push_back({NAME, [=] { hipMemsetAsync(cnt, 0, 1 << 20, st); hipLaunchKernelGGL((strip_copy_sync<K, EV, SL>), dim3(total), dim3(256), 0, st, p); }, {}}); }
    SYV(7, 1, 1, 360, "band sync: K7 every 7 rows slack 1 th360") SYV(7, 1, 2, 360, "band sync: K7 every 7 rows slack 2 th360") SYV(7, 2, 1, 360, "band sync: K7 every 14 rows slack 1 th360")
    SYV(7, 4, 1, 360, "band sync: K7 every 28 rows slack 1 th360") SYV(4, 1, 1, 360, "band sync: K4 every 4 rows slack 1 th360") SYV(4, 1, 2, 360, "band sync: K4 every 4 rows slack 2 th360")
    SYV(2, 1, 1, 360, "band sync: K2 every 2 rows slack 1 th360") SYV(2, 2, 1, 360, "band sync: K2 every 4 rows slack 1 th360") SYV(7, 1, 1, 2160, "band sync: K7 every 7 rows slack 1 th2160")
    SYV(7, 1, 100000, 360, "band sync: K7 counter only, never waits (control)")
    if (*only) vs.erase(std::remove_if(vs.begin(), vs.end(), [&](const V& v) { return !strstr(v.name.c_str(), only) && v.name.rfind("flat copy", 0) != 0; }), vs.end());
    for (int r = 0; r < ROUNDS + 1; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, st)); v.run(); CK(hipGetLastError()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    const double bytes = 2.0 * IMG * N * 4;
    printf("# %d f32x3 4K images, copy (R + W = %.2f GB), %d rounds interleaved; frac = GB/s / 8000\n", N, bytes / 1e9, ROUNDS);
    printf("%-66s %9s %9s %9s %6s\n", "variant", "med ms", "min ms", "GB/s@med", "frac");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        printf("%-66s %9.3f %9.3f %9.0f %6.3f\n", v.name.c_str(), med, v.ms[0], bytes / med / 1e6, bytes / med / 1e6 / 8000);
    }
    return 0;
}
