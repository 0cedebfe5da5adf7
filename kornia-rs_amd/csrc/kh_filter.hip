// Separable f32 filters for gfx950: gaussian_blur, box_blur, sobel, scharr, separable_filter.
//
// Behavioural contract: P/filter/separable_filter.rs:87-164 — a horizontal pass into an f32
// intermediate, then a vertical pass; `acc += v * k` in ascending tap order, uncontracted; taps
// that fall outside the image are skipped (zero border, no renormalisation).  The reference's
// device path runs that as TWO launches through a full-size scratch image
// (P/cuda/filter.rs:361-385: 2 reads + 2 writes per element) and sobel/scharr as FIVE launches
// with three scratch images (P/filter/cuda.rs:185-237).
//
// Here each filter is ONE kernel: a 256-thread block stages a (TH + ky - 1) x (256 + halo)
// tile of the flat row-major float image in LDS, runs the horizontal pass LDS -> LDS (the f32
// intermediate is kept, so every rounding of the two-pass reference survives), then the vertical
// pass LDS -> registers -> global with each thread owning one column for 8 consecutive rows
// (ky + 7 LDS reads per 8 outputs).  HBM traffic is 1 read (+ halo) + 1 write per element.
// The image is treated as H rows of W*C floats; a horizontal tap is a flat offset of +-C floats,
// which is in-bounds iff the flat index stays inside the row — so any channel count works.
// Skipping an out-of-image tap equals adding (0 * k): the accumulator starts at +0 and can never
// become -0, so zero-filling the halo is bit-identical to the reference's `if in-bounds` test.
#include <math.h>
#include <stdlib.h>

#include "kh_common.h"

using namespace kh;

namespace {

constexpr int kTF = 256;      // tile width in flat floats (one float per thread per row)
constexpr int kVR = 8;        // rows per thread in the vertical pass
constexpr int kMaxTaps = 63;

struct Taps {
    float k[64];
    int n;
};

struct FilterArgs {
    const float* src;
    float* dst;
    int rows, rowlen, C;  // rowlen = cols * C
    int th;               // output rows per tile (multiple of kVR)
    long long src_stride, dst_stride;
    XcdTiles tiles;  // (column tile, row strip, image), XCD-contiguous order
    int listed;      // image bases from the launch's PtrList (kh_common.h) instead of base + k * stride
    int xlo, xhi, ylo, yhi;   // MASKED rolling launches (unequal tap counts): the real taps of each pass inside the K padded ones
    int plain;                // rolling kernels: write-back stores instead of the streaming policy (kh_common.h::plain_row_stores)
};


extern __shared__ __attribute__((aligned(16))) float lds_f[];

// GRAD = false: dst = V_ky(H_kx(src)).  GRAD = true: dst = sqrt(gx^2 + gy^2) with
// gx = V_ky(H_kx(src)), gy = V_kx(H_ky(src))  (sobel / scharr, P/filter/ops.rs:174-247).
template <bool GRAD>
__global__ __launch_bounds__(kBlock) void sep_filter_kernel(FilterArgs a, Taps kx, Taps ky, PtrList lst) {
    const int tid = threadIdx.x;
    const int hx = kx.n / 2, hy = ky.n / 2;
    const int halo = (GRAD ? max(hx, hy) : hx) * a.C;  // flat floats on each side
    const int vhalo = GRAD ? max(hx, hy) : hy;
    const int in_w = kTF + 2 * halo;
    const int in_h = a.th + 2 * vhalo;
    float* tile = lds_f;                   // [in_h][in_w]   input
    float* mid = tile + in_h * in_w;       // [in_h][kTF]    H-pass output (gx path)
    float* mid2 = mid + in_h * kTF;        // [in_h][kTF]    (GRAD only: gy path)

    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;  // block-uniform
    const int x0 = tx * kTF, y0 = ty * a.th;
    const float* src = list_src(lst, a.listed, a.src, a.src_stride, bz);
    float* dst = list_dst(lst, a.listed, a.dst, a.dst_stride, bz);

    // 1. stage the input tile, zero outside the image
    for (int r = 0; r < in_h; ++r) {
        const int gy = y0 - vhalo + r;
        const bool row_ok = gy >= 0 && gy < a.rows;
        const float* srow = src + (long long)gy * a.rowlen;
        for (int i = tid; i < in_w; i += kBlock) {
            const int gx = x0 - halo + i;
            tile[r * in_w + i] = (row_ok && gx >= 0 && gx < a.rowlen) ? srow[gx] : 0.0f;
        }
    }
    __syncthreads();

    // 2. horizontal pass: thread = one flat column, all tile rows
    for (int r = 0; r < in_h; ++r) {
        const float* trow = tile + r * in_w + halo + tid;
        float acc = 0.0f;
        for (int i = 0; i < kx.n; ++i) acc += trow[(i - hx) * a.C] * kx.k[i];
        mid[r * kTF + tid] = acc;
        if constexpr (GRAD) {
            float acc2 = 0.0f;
            for (int i = 0; i < ky.n; ++i) acc2 += trow[(i - hy) * a.C] * ky.k[i];
            mid2[r * kTF + tid] = acc2;
        }
    }
    __syncthreads();

    // 3. vertical pass: thread = one flat column, kVR consecutive rows per step
    const int gx = x0 + tid;
    if (gx >= a.rowlen) return;
    for (int rb = 0; rb < a.th; rb += kVR) {
        float acc[kVR], acc2[kVR];
#pragma unroll
        for (int j = 0; j < kVR; ++j) { acc[j] = 0.0f; acc2[j] = 0.0f; }
        // out row (rb + j) uses mid rows (rb + j + vhalo - hy + i), i ascending
        {
            const int base = rb + vhalo - hy;
            for (int t = 0; t < ky.n + kVR - 1; ++t) {
                const float v = mid[(base + t) * kTF + tid];
#pragma unroll
                for (int j = 0; j < kVR; ++j) {
                    const int i = t - j;
                    if (i >= 0 && i < ky.n) acc[j] += v * ky.k[i];
                }
            }
        }
        if constexpr (GRAD) {
            const int base = rb + vhalo - hx;
            for (int t = 0; t < kx.n + kVR - 1; ++t) {
                const float v = mid2[(base + t) * kTF + tid];
#pragma unroll
                for (int j = 0; j < kVR; ++j) {
                    const int i = t - j;
                    if (i >= 0 && i < kx.n) acc2[j] += v * kx.k[i];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kVR; ++j) {
            const int gy = y0 + rb + j;
            if (gy < a.rows) {
                float o = acc[j];
                if constexpr (GRAD) o = sqrtf(acc[j] * acc[j] + acc2[j] * acc2[j]);
                dst[(long long)gy * a.rowlen + gx] = o;
            }
        }
    }
}


// ---- rolling-column kernel (the fast path) -------------------------------------------------------
// A wave owns 64 adjacent flat columns and walks down a strip of rows.  Per input row it loads its
// 64 floats (+ the horizontal halo, by the first 2*halo lanes), parks them in a 512-byte wave-private
// LDS row, computes the horizontal pass from LDS (K conflict-free ds_read_b32) and pushes the result
// into a K-deep REGISTER ring; the vertical pass is K multiply-adds on that ring.  No block barrier,
// 2 KiB of LDS per block, every input row read once per strip (+ ky-1 warm-up rows per strip), K rows
// of global loads in flight per lane; the walk is unrolled K times so ring slots are static registers.  Ascending-tap `acc += v*k` order and the f32
// intermediate are exactly those of the two-pass reference.
constexpr int kRollStripMax = 360;  // tallest strip (output rows)

struct TapsK { float k[16]; };

// MASKED (round 6): kx and ky of DIFFERENT lengths (gaussian (3, 7), a one-dimensional blur (9, 1), ...) centred in K = the longer one.
// A padded tap must not see its pixel: 0 * Inf would put a NaN where the reference's shorter window never looks — so the value is
// replaced by 0 outside the real taps [lo, hi) of the pass (a wave-uniform select per tap), and `acc + (+0)` leaves every accumulator
// as it is (it starts at +0 and cannot become -0).  These launches took the LDS-tile kernel before, 8-10x slower per image.
template <int K, bool GRAD, bool MASKED = false>
__global__ __launch_bounds__(kBlock) void sep_roll_kernel(FilterArgs a, TapsK kx, TapsK ky, PtrList lst) {
    __shared__ float rowbuf[4][160];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int H = K / 2;
    const int halo = H * a.C;  // <= 32 (checked on the host)
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int gx0 = tx * kTF + wv * 64;  // first flat column of this wave
    if (gx0 >= a.rowlen) return;         // whole wave idle (no block barrier below)
    const int y0 = ty * a.th;
    const float* __restrict__ src = list_src(lst, a.listed, a.src, a.src_stride, bz);
    float* __restrict__ dst = list_dst(lst, a.listed, a.dst, a.dst_stride, bz);
    float* buf = rowbuf[wv];

    const int gx = gx0 + lane;
    // halo lanes: [0, halo) fetch the left neighbours, [halo, 2*halo) the right ones
    const bool is_halo = lane < 2 * halo;
    const int hgx = lane < halo ? gx0 - halo + lane : gx0 + 64 + (lane - halo);
    const int hslot = lane < halo ? lane : 64 + lane;  // left [0,halo), main [halo,halo+64), right after
    const bool gx_ok = gx < a.rowlen, hgx_ok = is_halo && hgx >= 0 && hgx < a.rowlen;
    const int nrows = min(a.th, a.rows - y0) + 2 * H;  // input rows to walk (incl. warm-up)

    // Loads are UNCONDITIONAL (clamped addresses, zero selected afterwards): with a load inside a
    // divergent branch the compiler falls back to `s_waitcnt vmcnt(0)` at every use, which drains
    // the K rows of loads in flight each step (measured 3x slower).  Non-halo lanes re-load their
    // own main element as the "halo" value (an L1 hit) so the instruction needs no exec mask.
    const int cx_m = min(gx, a.rowlen - 1);
    const int cx_h = is_halo ? min(max(hgx, 0), a.rowlen - 1) : cx_m;
    int pf_row = y0 - H;  // image row of the next prefetch (may be outside the image: clamped)

    float qm[K], qh[K];  // K rows of loads in flight per lane
    // The queue holds RAW loaded values; the zero-select for out-of-image rows/columns happens at
    // use time, K steps later, so nothing consumes a load right after it is issued.
    auto prefetch = [&](float& m, float& hv) {
        const int base = min(max(pf_row, 0), a.rows - 1) * a.rowlen;  // 32-bit: host-checked
        m = src[base + cx_m];
        hv = src[base + cx_h];
        ++pf_row;
    };
#pragma unroll
    for (int p = 0; p < K; ++p) prefetch(qm[p], qh[p]);

    uint32_t xm[MASKED ? K : 1], ym[MASKED ? K : 1];   // MASKED: all ones for the real taps of a pass, zero for the padded ones (scalar registers)
    if constexpr (MASKED) {
#pragma unroll
        for (int i = 0; i < K; ++i) { xm[i] = (i >= a.xlo && i < a.xhi) ? 0xffffffffu : 0u; ym[i] = (i >= a.ylo && i < a.yhi) ? 0xffffffffu : 0u; }
    }
    float ring[K], ring2[GRAD ? K : 1];  // slot p holds the horizontal result of walk step == p (mod K)
#pragma unroll
    for (int i = 0; i < K; ++i) { ring[i] = 0.0f; if constexpr (GRAD) ring2[i] = 0.0f; }

    // The walk is branch-free except for the store predicate: steps past `nrows` (the last,
    // partial group of K) recompute clamped rows and store nothing; the first 2H steps are the
    // ring warm-up and store nothing either.
    const int hs = is_halo ? hslot : 159;  // non-halo lanes park their duplicate in a slot nobody reads
    // Output: a streaming window (kh_common.h) over this strip's rows of the destination — wave-uniform base, write-through
    // non-temporal dword stores (same-box A/B on C4: 9.11 vs 9.34 ms, profiles/r02o_ab_stores.txt); the offsets of the 2H warm-up
    // steps are negative and never stored.
    const __amdgpu_buffer_rsrc_t ow = stream_window(dst + (long long)y0 * a.rowlen, (long long)(a.rows - y0) * a.rowlen * 4);
    int out_off = (gx - 2 * H * a.rowlen) * 4;  // byte offset of walk step 0's (virtual) output row inside the window
    const float* tap = buf + halo + lane - H * a.C;
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            const int r = rb + p;
            const int row = y0 - H + r;
            const bool row_ok = row >= 0 && row < a.rows;  // wave-uniform
            const float m = (row_ok && gx_ok) ? qm[p] : 0.0f;
            const float hv = (row_ok && hgx_ok) ? qh[p] : 0.0f;
            prefetch(qm[p], qh[p]);
            buf[halo + lane] = m;
            buf[hs] = hv;
            // Cross-lane hand-off inside ONE wave: DS operations of a wave execute in issue order, so
            // the reads below see the writes above; the scheduling barrier only stops the compiler
            // from reordering them.  (A wavefront-scope fence here would also drain vmcnt and
            // serialise the K rows of loads in flight — measured 3x slower.)
            __builtin_amdgcn_wave_barrier();
            float h1 = 0.0f, h2 = 0.0f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float v = tap[i * a.C];
                if constexpr (MASKED) v = __uint_as_float(__float_as_uint(v) & xm[i]);
                h1 += v * kx.k[i];
                if constexpr (GRAD) h2 += v * ky.k[i];
            }
            __builtin_amdgcn_wave_barrier();  // every lane has read the row before it is overwritten
            ring[p] = h1;
            if constexpr (GRAD) ring2[p] = h2;
            float o = 0.0f, o2 = 0.0f;
#pragma unroll
            for (int i = 0; i < K; ++i) {  // oldest row first: ascending vertical taps
                float rv = ring[(p + 1 + i) % K];
                if constexpr (MASKED) rv = __uint_as_float(__float_as_uint(rv) & ym[i]);
                o += rv * ky.k[i];
                if constexpr (GRAD) o2 += ring2[(p + 1 + i) % K] * kx.k[i];
            }
            if constexpr (GRAD) o = sqrtf(o * o + o2 * o2);
            if (gx_ok && r >= 2 * H && r < nrows) { const uint32_t ow_bits = __float_as_uint(o); row_store<1>(ow, out_off, &ow_bits, a.plain); }
            out_off += a.rowlen * 4;
        }
    }
}

// ---- rolling-column kernel for WIDE kernels, 17..31 taps (round 6) -------------------------------------------------------------------
// The rolling kernels above stop at 15 taps (register ring, 32-float side buffers, 16-tap argument struct); everything wider took the
// LDS-tile kernel, which is 12x (17 x 17) to 65x (31 x 31) slower per image (profiles/r06zh_blur_sizes.txt) — and a gaussian of sigma 3-5
// is 19-31 taps.  The same walk with a K-deep ring, side buffers of up to 64 floats ((K / 2) * C <= 64: two halo loads per lane and
// row) and 32-tap arguments.  Same products in the same order as sep_roll_kernel<K, false>.
struct TapsW { float k[32]; };
template <int K>
__global__ __launch_bounds__(kBlock) void sep_roll_wide_kernel(FilterArgs a, TapsW kx, TapsW ky, PtrList lst) {
    __shared__ float rowbuf[4][64 + 64 + 64 + 4];   // left halo | main | right halo | parking slot
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int H = K / 2;
    const int halo = H * a.C;  // <= 64 (checked on the host)
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int gx0 = tx * kTF + wv * 64;
    if (gx0 >= a.rowlen) return;
    const int y0 = ty * a.th;
    const float* __restrict__ src = list_src(lst, a.listed, a.src, a.src_stride, bz);
    float* __restrict__ dst = list_dst(lst, a.listed, a.dst, a.dst_stride, bz);
    float* buf = rowbuf[wv];
    const int gx = gx0 + lane;
    const bool gx_ok = gx < a.rowlen;
    // halo elements e = lane and lane + 64 of the 2 * halo neighbours: [0, halo) the left ones, [halo, 2 halo) the right ones
    int hcx[2], hslot[2];
    bool h_ok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = lane + 64 * q;
        const bool is_halo = e < 2 * halo;
        const int hgx = e < halo ? gx0 - halo + e : gx0 + 64 + (e - halo);
        h_ok[q] = is_halo && hgx >= 0 && hgx < a.rowlen;
        hcx[q] = is_halo ? min(max(hgx, 0), a.rowlen - 1) : min(gx, a.rowlen - 1);
        hslot[q] = is_halo ? (e < halo ? e : 64 + e) : 192 + (q << 1);   // non-halo lanes park their duplicate where nobody reads
    }
    const int nrows = min(a.th, a.rows - y0) + 2 * H;
    const int cx_m = min(gx, a.rowlen - 1);
    int pf_row = y0 - H;
    float qm[K], qh[K][2];
    auto prefetch = [&](float& m, float (&hv)[2]) {
        const int base = min(max(pf_row, 0), a.rows - 1) * a.rowlen;  // 32-bit: host-checked
        m = src[base + cx_m];
        hv[0] = src[base + hcx[0]];
        hv[1] = src[base + hcx[1]];
        ++pf_row;
    };
#pragma unroll
    for (int p = 0; p < K; ++p) prefetch(qm[p], qh[p]);
    float ring[K];
#pragma unroll
    for (int i = 0; i < K; ++i) ring[i] = 0.0f;
    const __amdgpu_buffer_rsrc_t ow = stream_window(dst + (long long)y0 * a.rowlen, (long long)(a.rows - y0) * a.rowlen * 4);
    int out_off = (gx - 2 * H * a.rowlen) * 4;
    const float* tap = buf + halo + lane - H * a.C;
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            const int r = rb + p;
            const int row = y0 - H + r;
            const bool row_ok = row >= 0 && row < a.rows;  // wave-uniform
            const float m = (row_ok && gx_ok) ? qm[p] : 0.0f;
            const float hv0 = (row_ok && h_ok[0]) ? qh[p][0] : 0.0f, hv1 = (row_ok && h_ok[1]) ? qh[p][1] : 0.0f;
            prefetch(qm[p], qh[p]);
            buf[halo + lane] = m;
            buf[hslot[0]] = hv0;
            buf[hslot[1]] = hv1;
            __builtin_amdgcn_wave_barrier();
            float h1 = 0.0f;
#pragma unroll
            for (int i = 0; i < K; ++i) h1 += tap[i * a.C] * kx.k[i];
            __builtin_amdgcn_wave_barrier();  // every lane has read the row before it is overwritten
            ring[p] = h1;
            float o = 0.0f;
#pragma unroll
            for (int i = 0; i < K; ++i) o += ring[(p + 1 + i) % K] * ky.k[i];   // oldest row first: ascending vertical taps
            if (gx_ok && r >= 2 * H && r < nrows) { const uint32_t ow_bits = __float_as_uint(o); row_store<1>(ow, out_off, &ow_bits, a.plain); }
            out_off += a.rowlen * 4;
        }
    }
}
template <int K>
void launch_roll_wide(hipStream_t st, dim3 grid, const FilterArgs& a, const Taps& kx, const Taps& ky, const PtrList& lst) {
    TapsW wx, wy;
    for (int i = 0; i < 32; ++i) { wx.k[i] = i < kx.n ? kx.k[i] : 0.0f; wy.k[i] = i < ky.n ? ky.k[i] : 0.0f; }
    hipLaunchKernelGGL((sep_roll_wide_kernel<K>), grid, dim3(kBlock), 0, st, a, wx, wy, lst);
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr bool kFourColumnsDefault = true;   // sep_roll4_kernel where it applies: 2-7 % faster than one column per lane on C4, same box, interleaved (profiles/r03k, r03l); test option filter_four_columns = 0 turns it off


// ---- rolling-column kernel, FOUR columns per lane (round 3) -------------------------------------------------------------------
// A lane owns four adjacent flat columns as a float4: ONE 16-byte global load and ONE 16-byte store per row (1 KiB contiguous per
// wave instruction — the store granularity that mattered on the north star — and a quarter of the vector-memory instructions of
// the one-column kernel), the row parked in LDS with one ds_write_b128, the horizontal taps cut out of 2 * ceil(H*C / 4) + 1
// aligned ds_read_b128 at COMPILE-TIME register offsets (C is a template parameter here), both passes in packed f32 math.  Same
// IEEE operations in the same order per element as the one-column kernel.  A wave covers 256 columns, a 256-thread block 1024;
// rows must be a multiple of four floats and 16-byte aligned (host-checked), gradients keep the one-column kernel.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kTF4 = 4 * kTF;  // flat columns per 256-thread block

// GRAD (round 6): the gradient magnitude of sep_roll_kernel<K, true> — gx = V_ky(H_kx), gy = V_kx(H_ky), sqrt(gx^2 + gy^2) — on the same
// four-column walk: a second register ring, the same products in the same order per element.
template <int K, int C, bool GRAD = false>
__global__ __launch_bounds__(kBlock) void sep_roll4_kernel(FilterArgs a, TapsK kx, TapsK ky, PtrList lst) {
    constexpr int H = K / 2, HALO = H * C;       // <= 32 (checked on the host)
    constexpr int HQ = (HALO + 3) / 4;           // halo in float4 chunks
    constexpr int NCH = 2 * HQ + 1;              // chunks a lane reads per row
    __shared__ __attribute__((aligned(16))) float rowbuf[4][32 + 256 + 32 + 4];  // left halo | main | right halo | parking slot
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int gx0 = tx * kTF4 + wv * 256;   // first flat column of this wave
    if (gx0 >= a.rowlen) return;             // whole wave idle (no block barrier below)
    const int y0 = ty * a.th;
    const float* __restrict__ src = list_src(lst, a.listed, a.src, a.src_stride, bz);
    float* __restrict__ dst = list_dst(lst, a.listed, a.dst, a.dst_stride, bz);
    float* buf = rowbuf[wv];

    const int gx = gx0 + 4 * lane;           // this lane's columns gx .. gx + 3 (rowlen % 4 == 0: all four in or all four out)
    const bool is_halo = lane < 2 * HALO;    // lanes [0, HALO) fetch the left neighbours, [HALO, 2 HALO) the right ones
    const int hgx = lane < HALO ? gx0 - HALO + lane : gx0 + 256 + (lane - HALO);
    const int hslot = lane < HALO ? 32 - HALO + lane : 288 + (lane - HALO);
    const bool gx_ok = gx < a.rowlen, hgx_ok = is_halo && hgx >= 0 && hgx < a.rowlen;
    const int nrows = min(a.th, a.rows - y0) + 2 * H;
    const int cx_m = min(gx, a.rowlen - 4);
    const int cx_h = is_halo ? min(max(hgx, 0), a.rowlen - 1) : cx_m;
    int pf_row = y0 - H;

    f32x4_t qm[K];
    float qh[K];
    auto prefetch = [&](f32x4_t& m, float& hv) {
        const int base = min(max(pf_row, 0), a.rows - 1) * a.rowlen;  // 32-bit: host-checked
        m = *reinterpret_cast<const f32x4_t*>(src + base + cx_m);
        hv = src[base + cx_h];
        ++pf_row;
    };
#pragma unroll
    for (int p = 0; p < K; ++p) prefetch(qm[p], qh[p]);

    f32x2_t ring[K][2], ring2[GRAD ? K : 1][2];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        ring[i][0] = f32x2_t{0.0f, 0.0f}; ring[i][1] = f32x2_t{0.0f, 0.0f};
        if constexpr (GRAD) { ring2[i][0] = f32x2_t{0.0f, 0.0f}; ring2[i][1] = f32x2_t{0.0f, 0.0f}; }
    }

    const int hs = is_halo ? hslot : 320;  // non-halo lanes park their duplicate in a slot nobody reads
    const __amdgpu_buffer_rsrc_t ow = stream_window(dst + (long long)y0 * a.rowlen, (long long)(a.rows - y0) * a.rowlen * 4);
    int out_off = (gx - 2 * H * a.rowlen) * 4;
    const f32x4_t* span_p = reinterpret_cast<const f32x4_t*>(buf + 32 + 4 * lane - 4 * HQ);   // 16-byte aligned
    constexpr int kOff = 4 * HQ - HALO;       // float index of tap 0 of column 0 inside the span
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            const int r = rb + p;
            const int row = y0 - H + r;
            const bool row_ok = row >= 0 && row < a.rows;  // wave-uniform
            const f32x4_t m = (row_ok && gx_ok) ? qm[p] : f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            const float hv = (row_ok && hgx_ok) ? qh[p] : 0.0f;
            prefetch(qm[p], qh[p]);
            *reinterpret_cast<f32x4_t*>(buf + 32 + 4 * lane) = m;
            buf[hs] = hv;
            __builtin_amdgcn_wave_barrier();
            float span[4 * NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f32x4_t v = span_p[c];
                span[4 * c] = v.x; span[4 * c + 1] = v.y; span[4 * c + 2] = v.z; span[4 * c + 3] = v.w;
            }
            __builtin_amdgcn_wave_barrier();  // every lane has read the row before it is overwritten
            f32x2_t h01 = {0.0f, 0.0f}, h23 = {0.0f, 0.0f}, g01 = {0.0f, 0.0f}, g23 = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const f32x2_t v01 = {span[kOff + i * C], span[kOff + i * C + 1]}, v23 = {span[kOff + i * C + 2], span[kOff + i * C + 3]};
                h01 += v01 * kx.k[i];   // v_pk_mul_f32 then v_pk_add_f32: two roundings per element, as the reference
                h23 += v23 * kx.k[i];
                if constexpr (GRAD) { g01 += v01 * ky.k[i]; g23 += v23 * ky.k[i]; }
            }
            ring[p][0] = h01; ring[p][1] = h23;
            if constexpr (GRAD) { ring2[p][0] = g01; ring2[p][1] = g23; }
            f32x2_t o01 = {0.0f, 0.0f}, o23 = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < K; ++i) {  // oldest row first: ascending vertical taps
                o01 += ring[(p + 1 + i) % K][0] * ky.k[i];
                o23 += ring[(p + 1 + i) % K][1] * ky.k[i];
            }
            if constexpr (GRAD) {
                f32x2_t p01 = {0.0f, 0.0f}, p23 = {0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    p01 += ring2[(p + 1 + i) % K][0] * kx.k[i];
                    p23 += ring2[(p + 1 + i) % K][1] * kx.k[i];
                }
                const f32x2_t s01 = o01 * o01 + p01 * p01, s23 = o23 * o23 + p23 * p23;   // sqrt(gx^2 + gy^2), P/filter/ops.rs:174-247
                o01 = f32x2_t{sqrtf(s01.x), sqrtf(s01.y)};
                o23 = f32x2_t{sqrtf(s23.x), sqrtf(s23.y)};
            }
            if (gx_ok && r >= 2 * H && r < nrows) {
                const uint32_t bits[4] = {__float_as_uint(o01.x), __float_as_uint(o01.y), __float_as_uint(o23.x), __float_as_uint(o23.y)};
                row_store<4>(ow, out_off, bits, a.plain);
            }
            out_off += a.rowlen * 4;
        }
    }
}

template <int K>
bool launch_roll4_grad(hipStream_t st, dim3 grid, const FilterArgs& a, const TapsK& kx, const TapsK& ky, const PtrList& lst) {
    switch (a.C) {
        case 1: hipLaunchKernelGGL((sep_roll4_kernel<K, 1, true>), grid, dim3(kBlock), 0, st, a, kx, ky, lst); return true;
        case 3: hipLaunchKernelGGL((sep_roll4_kernel<K, 3, true>), grid, dim3(kBlock), 0, st, a, kx, ky, lst); return true;
        case 4: hipLaunchKernelGGL((sep_roll4_kernel<K, 4, true>), grid, dim3(kBlock), 0, st, a, kx, ky, lst); return true;
        default: return false;
    }
}
template <int K>
bool launch_roll4(hipStream_t st, dim3 grid, const FilterArgs& a, const TapsK& kx, const TapsK& ky, const PtrList& lst) {
    switch (a.C) {
        case 1: hipLaunchKernelGGL((sep_roll4_kernel<K, 1>), grid, dim3(kBlock), 0, st, a, kx, ky, lst); return true;
        case 3: hipLaunchKernelGGL((sep_roll4_kernel<K, 3>), grid, dim3(kBlock), 0, st, a, kx, ky, lst); return true;
        case 4: hipLaunchKernelGGL((sep_roll4_kernel<K, 4>), grid, dim3(kBlock), 0, st, a, kx, ky, lst); return true;
        default: return false;
    }
}

template <int K>
void launch_roll_masked(hipStream_t st, dim3 grid, const FilterArgs& a, const TapsK& kx, const TapsK& ky, const PtrList& lst) {
    hipLaunchKernelGGL((sep_roll_kernel<K, false, true>), grid, dim3(kBlock), 0, st, a, kx, ky, lst);
}
template <int K>
void launch_roll(hipStream_t st, dim3 grid, bool grad, const FilterArgs& a, const TapsK& kx, const TapsK& ky, const PtrList& lst) {
    if (grad) hipLaunchKernelGGL((sep_roll_kernel<K, true>), grid, dim3(kBlock), 0, st, a, kx, ky, lst);
    else hipLaunchKernelGGL((sep_roll_kernel<K, false>), grid, dim3(kBlock), 0, st, a, kx, ky, lst);
}

// Centre an n-tap kernel inside K taps.  The zero pad taps contribute (+-0) to the accumulator,
// which never changes it (see the header comment), so the result equals the unpadded filter for
// all finite inputs.
void pad_taps(TapsK& out, const Taps& in, int K) {
    const int off = (K - in.n) / 2;
    for (int i = 0; i < 16; ++i) out.k[i] = (i >= off && i < off + in.n) ? in.k[i - off] : 0.0f;
}

int32_t set_taps(Taps& t, const float* k, int n, const char* what) {
    // an empty kernel is the reference's InvalidKernelLength (P/filter/separable_filter.rs:175-180)
    KH_REQUIRE(k && n >= 1, KH_ERR_INVALID_ARG, "%s: invalid kernel length %d (null or empty kernel)", what, k ? n : 0);
    KH_REQUIRE(n <= kMaxTaps, KH_ERR_UNSUPPORTED, "%s: kernel length %d exceeds %d taps", what, n, kMaxTaps);
    for (int i = 0; i < 64; ++i) t.k[i] = i < n ? k[i] : 0.0f;
    t.n = n;
    return KH_OK;
}

// test option filter_force_tile = 1 routes every call to the LDS-tile kernel (parity tests cover both).
bool force_tile_kernel() { return dev_opt(kOptFilterForceTile) == 1; }

// `whole`: the batch as the caller gave it (strided, or host lists of device pointers); one launch per for_each_launch slice
int32_t launch(kh_stream_t stream, const BatchRef& whole, int cols, int rows, int C, const Taps& kx, const Taps& ky, bool grad,
               const char* what) {
    KH_REQUIRE(cols > 0 && rows > 0 && C > 0, KH_ERR_INVALID_ARG, "%s: zero-sized image %dx%dx%d", what, cols, rows, C);
    KH_REQUIRE(whole.n >= 0 && whole.n <= 65535, KH_ERR_TOO_LARGE, "%s: batch %d outside [0, 65535]", what, whole.n);
    KH_REQUIRE((int64_t)cols * rows * C <= kI32Max, KH_ERR_TOO_LARGE, "%s: image exceeds 32-bit indexing", what);
    if (whole.n == 0) return KH_OK;
    if (whole.listed()) {
        if (int32_t rc = check_list(what, whole.srcs, whole.dsts, whole.n)) return rc;   // (also rejects srcs[k] == dsts[k])
    } else {
        KH_REQUIRE(whole.src && whole.dst, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
        KH_REQUIRE(whole.src != whole.dst, KH_ERR_INVALID_ARG, "%s: in-place filtering is not supported", what);
    }
    const bool all_16B = batch_aligned(whole, 16, sizeof(float));
    return for_each_launch(whole, [&](const BatchRef& b, int, const PtrList& lst) -> int32_t {
    const int batch = b.n;
    FilterArgs a;
    a.src = static_cast<const float*>(b.src); a.dst = static_cast<float*>(b.dst); a.rows = rows; a.rowlen = cols * C; a.C = C;
    a.src_stride = b.ss; a.dst_stride = b.ds;
    a.listed = b.listed() ? 1 : 0;
    a.xlo = a.xhi = a.ylo = a.yhi = 0;
    a.plain = plain_row_stores((int64_t)a.rowlen * 4, b.listed() ? nullptr : b.dst, b.ds * 4, batch);

    // Fast path: rolling-column kernel for odd kernels up to 15 taps whose horizontal halo fits
    // the 32-float side buffers; everything else takes the LDS-tile kernel below.
    // Equal tap counts only: padding the shorter kernel with zero taps would multiply pixels OUTSIDE the reference's n-tap window
    // by 0, and 0 * Inf = NaN — a non-finite pixel would poison a wider neighbourhood than separable_filter.rs does.
    // Unequal odd sizes (rare: e.g. gaussian (3, 7)) take the tile kernel, which walks exactly kx.n / ky.n taps.  3 is the
    // smallest rolling instantiation, so a 1-tap pair goes to the tile kernel as well.
    const int kmax = kx.n > ky.n ? kx.n : ky.n;
    const bool strip_fits_32bit = (int64_t)(kRollStripMax + 16) * a.rowlen * 4 <= kI32Max;  // byte offsets inside a strip's output window
    const bool unequal_ok = !grad && kx.n != ky.n && (kx.n & 1) && (ky.n & 1);   // (round 6: the MASKED rolling kernel)
    if ((kx.n == ky.n || unequal_ok) && (kx.n & 1) && kmax >= 3 && kmax <= 15 && (kmax / 2) * C <= 32 && strip_fits_32bit && !force_tile_kernel()) {
        const int K = kmax < 3 ? 3 : kmax;
        TapsK px, py;
        pad_taps(px, kx, K);
        pad_taps(py, ky, K);
        // four columns per lane (sep_roll4_kernel) for K <= 9, C in {1, 3, 4}, rows that are a multiple of four floats and fill one
        // 1024-column block, 16-byte aligned images; everything else (and test option filter_four_columns = 0) takes one column per lane.
        // (A two-column packed-f32 kernel was measured in round 2 and is not in the library.)
        const int four_opt = dev_opt(kOptFilterFourColumns);
        const bool four_cols = four_opt < 0 ? kFourColumnsDefault : four_opt == 1;
        // (gradients: K = 3 and 5 only — sobel / scharr — since round 6)
        a.xlo = (K - kx.n) / 2; a.xhi = a.xlo + kx.n; a.ylo = (K - ky.n) / 2; a.yhi = a.ylo + ky.n;
        const bool four = !unequal_ok && (!grad || K <= 5) && four_cols && K <= 9 && (C == 1 || C == 3 || C == 4) && (a.rowlen % 4 == 0) && a.rowlen >= kTF4 && all_16B;
        const unsigned tiles_x = cdiv(a.rowlen, four ? kTF4 : kTF);  // 256-thread blocks: 512 measured +1 %, 1024 +9 % (r01q)
        // Strip height: tall strips amortise the ky-1 warm-up rows (4K x 256 images: 360 rows is
        // 5 % faster than 90), short strips keep a small launch wide enough to fill 256 CUs.
        {
            const long long cols_blocks = (long long)tiles_x * batch;
            long long strips = (2048 + cols_blocks - 1) / cols_blocks;          // >= 8 blocks per CU
            const long long min_strips = cdiv(rows, kRollStripMax), max_strips = cdiv(rows, 32);
            strips = strips < min_strips ? min_strips : (strips > max_strips ? max_strips : strips);
            a.th = (int)cdiv(rows, strips);
        }
        a.tiles = xcd_tiles(tiles_x, cdiv(rows, a.th), (unsigned)batch, kXcdEighth);
        KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
        const dim3 grid = xcd_grid(a.tiles);
        hipStream_t st = as_hip(stream);
        if (four && grad) {
            if (K == 3) launch_roll4_grad<3>(st, grid, a, px, py, lst);
            else launch_roll4_grad<5>(st, grid, a, px, py, lst);
            return check_launch(what);
        }
        if (four) {
            switch (K) {
                case 3: launch_roll4<3>(st, grid, a, px, py, lst); break;
                case 5: launch_roll4<5>(st, grid, a, px, py, lst); break;
                case 7: launch_roll4<7>(st, grid, a, px, py, lst); break;
                default: launch_roll4<9>(st, grid, a, px, py, lst); break;
            }
            return check_launch(what);
        }
        if (unequal_ok) {
            switch (K) {
                case 3: launch_roll_masked<3>(st, grid, a, px, py, lst); break;
                case 5: launch_roll_masked<5>(st, grid, a, px, py, lst); break;
                case 7: launch_roll_masked<7>(st, grid, a, px, py, lst); break;
                case 9: launch_roll_masked<9>(st, grid, a, px, py, lst); break;
                case 11: launch_roll_masked<11>(st, grid, a, px, py, lst); break;
                case 13: launch_roll_masked<13>(st, grid, a, px, py, lst); break;
                default: launch_roll_masked<15>(st, grid, a, px, py, lst); break;
            }
            return check_launch(what);
        }
        switch (K) {
            case 3: launch_roll<3>(st, grid, grad, a, px, py, lst); break;
            case 5: launch_roll<5>(st, grid, grad, a, px, py, lst); break;
            case 7: launch_roll<7>(st, grid, grad, a, px, py, lst); break;
            case 9: launch_roll<9>(st, grid, grad, a, px, py, lst); break;
            case 11: launch_roll<11>(st, grid, grad, a, px, py, lst); break;
            case 13: launch_roll<13>(st, grid, grad, a, px, py, lst); break;
            default: launch_roll<15>(st, grid, grad, a, px, py, lst); break;
        }
        return check_launch(what);
    }

    // 17..31 equal odd tap counts: the wide rolling kernel (side buffers of (K / 2) * C <= 64 floats)
    if (!grad && kx.n == ky.n && (kx.n & 1) && kx.n >= 17 && kx.n <= 31 && (kx.n / 2) * C <= 64 && (int64_t)(kRollStripMax + 32) * a.rowlen * 4 <= kI32Max && !force_tile_kernel()) {
        const unsigned tiles_x = cdiv(a.rowlen, kTF);
        {
            const long long cols_blocks = (long long)tiles_x * batch;
            long long strips = (2048 + cols_blocks - 1) / cols_blocks;
            const long long min_strips = cdiv(rows, kRollStripMax), max_strips = cdiv(rows, 4 * kx.n);   // strips of at least four kernel heights: the K - 1 warm-up rows
            strips = strips < min_strips ? min_strips : (strips > max_strips ? max_strips : strips);
            if (strips < 1) strips = 1;
            a.th = (int)cdiv(rows, strips);
        }
        a.tiles = xcd_tiles(tiles_x, cdiv(rows, a.th), (unsigned)batch, kXcdEighth);
        KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
        const dim3 grid = xcd_grid(a.tiles);
        hipStream_t st = as_hip(stream);
        switch (kx.n) {
            case 17: launch_roll_wide<17>(st, grid, a, kx, ky, lst); break;
            case 19: launch_roll_wide<19>(st, grid, a, kx, ky, lst); break;
            case 21: launch_roll_wide<21>(st, grid, a, kx, ky, lst); break;
            case 23: launch_roll_wide<23>(st, grid, a, kx, ky, lst); break;
            case 25: launch_roll_wide<25>(st, grid, a, kx, ky, lst); break;
            case 27: launch_roll_wide<27>(st, grid, a, kx, ky, lst); break;
            case 29: launch_roll_wide<29>(st, grid, a, kx, ky, lst); break;
            default: launch_roll_wide<31>(st, grid, a, kx, ky, lst); break;
        }
        return check_launch(what);
    }
    const int hmax = grad ? (kx.n > ky.n ? kx.n : ky.n) / 2 : 0;
    const int halo = (grad ? hmax : kx.n / 2) * C, vhalo = grad ? hmax : ky.n / 2;
    // Pick the tallest tile (fewest halo re-reads) that still lets two blocks share a CU's LDS.
    int th = 64;
    size_t bytes = 0;
    for (;; th -= kVR) {
        const size_t in_h = th + 2 * vhalo;
        bytes = (in_h * (kTF + 2 * halo) + in_h * kTF * (grad ? 2 : 1)) * sizeof(float);
        if (bytes <= 78 * 1024 || th == kVR) break;
    }
    KH_REQUIRE(bytes <= 160 * 1024, KH_ERR_UNSUPPORTED, "%s: %dx%d taps with %d channels need %zu B of LDS", what, kx.n,
               ky.n, C, bytes);
    if (rows < th) th = ((rows + kVR - 1) / kVR) * kVR;
    a.th = th;
    a.tiles = xcd_tiles(cdiv(a.rowlen, kTF), cdiv(rows, th), (unsigned)batch, cdiv(a.rowlen, kTF) * 2);
    KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
    auto kern = grad ? sep_filter_kernel<true> : sep_filter_kernel<false>;
    const size_t in_h = th + 2 * vhalo;
    bytes = (in_h * (kTF + 2 * halo) + in_h * kTF * (grad ? 2 : 1)) * sizeof(float);
    if (bytes > 48 * 1024)  // opt in to the full 160 KiB LDS of a gfx950 CU (per device, idempotent)
        KH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, xcd_grid(a.tiles), dim3(kBlock), bytes, as_hip(stream), a, kx, ky, lst);
    return check_launch(what);
    });
}

}  // namespace

extern "C" {

// P/filter/kernels.rs:10-13
int32_t kh_box_blur_kernel_1d(int32_t n, float* out) {
    KH_REQUIRE(n >= 1 && out, KH_ERR_INVALID_ARG, "kh_box_blur_kernel_1d: bad arguments");
    for (int i = 0; i < n; ++i) out[i] = 1.0f / (float)n;
    return KH_OK;
}

// P/filter/kernels.rs:25-43 — host expf (the reference evaluates exp on the host too)
int32_t kh_gaussian_kernel_1d(int32_t n, float sigma, float* out) {
    KH_REQUIRE(n >= 1 && out, KH_ERR_INVALID_ARG, "kh_gaussian_kernel_1d: bad arguments");
    const float mean = (float)(n - 1) / 2.0f, sigma_sq = sigma * sigma;
    for (int i = 0; i < n; ++i) {
        const float x = (float)i - mean;
        out[i] = expf(-(x * x) / (2.0f * sigma_sq));
    }
    float norm = 0.0f;
    for (int i = 0; i < n; ++i) norm += out[i];
    for (int i = 0; i < n; ++i) out[i] /= norm;
    return KH_OK;
}

// P/filter/ops.rs:122-155 (SciPy conventions); k = {kx, ky}, s = {sx, sy}, updated in place
int32_t kh_gaussian_resolve(int32_t k[2], float s[2]) {
    KH_REQUIRE(k && s, KH_ERR_INVALID_ARG, "kh_gaussian_resolve: null pointer");
    int kx = k[0], ky = k[1];
    float sx = s[0], sy = s[1];
    if (sy <= 0.0f) sy = sx;
    if (kx == 0 && sx > 0.0f) kx = (int)(2.0f * roundf(4.0f * sx) + 1.0f) | 1;
    if (ky == 0 && sy > 0.0f) ky = (int)(2.0f * roundf(4.0f * sy) + 1.0f) | 1;
    KH_REQUIRE(kx > 0 && kx % 2 == 1 && ky > 0 && ky % 2 == 1, KH_ERR_INVALID_ARG,
               "invalid sigma/kernel-size combination: kernel (%d, %d), sigma (%g, %g)", kx, ky, (double)sx, (double)sy);
    sx = sx > 0.0f ? sx : 0.0f;
    sy = sy > 0.0f ? sy : 0.0f;
    if (sx == 0.0f) sx = ((float)kx - 1.0f) / 8.0f;
    if (sy == 0.0f) sy = ((float)ky - 1.0f) / 8.0f;
    k[0] = kx; k[1] = ky; s[0] = sx; s[1] = sy;
    return KH_OK;
}

}  // extern "C"

namespace {

int32_t separable_impl(const char* what, kh_stream_t stream, const BatchRef& b, int cols, int rows, int channels, const float* kernel_x, int nx,
                       const float* kernel_y, int ny) {
    Taps kx, ky;
    if (int32_t rc = set_taps(kx, kernel_x, nx, what)) return rc;
    if (int32_t rc = set_taps(ky, kernel_y, ny, what)) return rc;
    return launch(stream, b, cols, rows, channels, kx, ky, false, what);
}

int32_t gaussian_impl(const char* what, kh_stream_t stream, const BatchRef& b, int cols, int rows, int channels, int ksize_x, int ksize_y,
                      float sigma_x, float sigma_y) {
    int32_t k[2] = {ksize_x, ksize_y};
    float s[2] = {sigma_x, sigma_y};
    if (int32_t rc = kh_gaussian_resolve(k, s)) return rc;
    KH_REQUIRE(k[0] <= kMaxTaps && k[1] <= kMaxTaps, KH_ERR_UNSUPPORTED, "%s: kernel (%d, %d) exceeds %d taps", what, k[0], k[1], kMaxTaps);
    float tx[64], ty[64];
    kh_gaussian_kernel_1d(k[0], s[0], tx);
    kh_gaussian_kernel_1d(k[1], s[1], ty);
    Taps kx, ky;
    set_taps(kx, tx, k[0], what);
    set_taps(ky, ty, k[1], what);
    return launch(stream, b, cols, rows, channels, kx, ky, false, what);
}

int32_t box_impl(const char* what, kh_stream_t stream, const BatchRef& b, int cols, int rows, int channels, int ksize_x, int ksize_y) {
    KH_REQUIRE(ksize_x >= 1 && ksize_y >= 1, KH_ERR_INVALID_ARG, "%s: invalid kernel length (%d, %d)", what, ksize_x, ksize_y);
    KH_REQUIRE(ksize_x <= kMaxTaps && ksize_y <= kMaxTaps, KH_ERR_UNSUPPORTED, "%s: kernel (%d, %d) exceeds %d taps", what, ksize_x, ksize_y, kMaxTaps);
    float tx[64], ty[64];
    kh_box_blur_kernel_1d(ksize_x, tx);
    kh_box_blur_kernel_1d(ksize_y, ty);
    Taps kx, ky;
    set_taps(kx, tx, ksize_x, what);
    set_taps(ky, ty, ksize_y, what);
    return launch(stream, b, cols, rows, channels, kx, ky, false, what);
}

// kind: KH_GRAD_SOBEL (size 3 | 5) or KH_GRAD_SCHARR (size 3) — P/filter/kernels.rs:55-100
int32_t gradient_impl(const char* what, kh_stream_t stream, const BatchRef& b, int cols, int rows, int channels, int kind, int ksize) {
    static const float s3x[3] = {-1, 0, 1}, s3y[3] = {1, 2, 1}, s5x[5] = {-1, -2, 0, 2, 1}, s5y[5] = {1, 4, 6, 4, 1},
                       c3y[3] = {3, 10, 3};
    const float *tx = nullptr, *ty = nullptr;
    if (kind == KH_GRAD_SOBEL && ksize == 3) { tx = s3x; ty = s3y; }
    else if (kind == KH_GRAD_SOBEL && ksize == 5) { tx = s5x; ty = s5y; }
    else if (kind == KH_GRAD_SCHARR && ksize == 3) { tx = s3x; ty = c3y; }
    KH_REQUIRE(tx, KH_ERR_INVALID_ARG, "invalid kernel length %d for gradient kind %d", ksize, kind);
    Taps kx, ky;
    set_taps(kx, tx, ksize, what);
    set_taps(ky, ty, ksize, what);
    return launch(stream, b, cols, rows, channels, kx, ky, true, what);
}

}  // namespace

extern "C" {

int32_t kh_separable_filter_f32(kh_stream_t stream, const float* src, float* dst, int32_t cols, int32_t rows,
                                int32_t channels, const float* kernel_x, int32_t nx, const float* kernel_y, int32_t ny,
                                int32_t batch, int64_t src_stride, int64_t dst_stride) {
    return separable_impl("kh_separable_filter_f32", stream, strided_batch(src, dst, batch, src_stride, dst_stride), cols, rows, channels,
                          kernel_x, nx, kernel_y, ny);
}
int32_t kh_separable_filter_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t cols,
                                     int32_t rows, int32_t channels, const float* kernel_x, int32_t nx, const float* kernel_y,
                                     int32_t ny) {
    return separable_impl("kh_separable_filter_f32_list", stream, listed_batch(srcs, dsts, n), cols, rows, channels, kernel_x, nx, kernel_y, ny);
}

int32_t kh_gaussian_blur_f32(kh_stream_t stream, const float* src, float* dst, int32_t cols, int32_t rows,
                             int32_t channels, int32_t ksize_x, int32_t ksize_y, float sigma_x, float sigma_y,
                             int32_t batch, int64_t src_stride, int64_t dst_stride) {
    return gaussian_impl("kh_gaussian_blur_f32", stream, strided_batch(src, dst, batch, src_stride, dst_stride), cols, rows, channels, ksize_x,
                         ksize_y, sigma_x, sigma_y);
}
int32_t kh_gaussian_blur_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t cols,
                                  int32_t rows, int32_t channels, int32_t ksize_x, int32_t ksize_y, float sigma_x, float sigma_y) {
    return gaussian_impl("kh_gaussian_blur_f32_list", stream, listed_batch(srcs, dsts, n), cols, rows, channels, ksize_x, ksize_y, sigma_x, sigma_y);
}

int32_t kh_box_blur_f32(kh_stream_t stream, const float* src, float* dst, int32_t cols, int32_t rows, int32_t channels,
                        int32_t ksize_x, int32_t ksize_y, int32_t batch, int64_t src_stride, int64_t dst_stride) {
    return box_impl("kh_box_blur_f32", stream, strided_batch(src, dst, batch, src_stride, dst_stride), cols, rows, channels, ksize_x, ksize_y);
}
int32_t kh_box_blur_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t cols, int32_t rows,
                             int32_t channels, int32_t ksize_x, int32_t ksize_y) {
    return box_impl("kh_box_blur_f32_list", stream, listed_batch(srcs, dsts, n), cols, rows, channels, ksize_x, ksize_y);
}

int32_t kh_gradient_magnitude_f32(kh_stream_t stream, const float* src, float* dst, int32_t cols, int32_t rows,
                                  int32_t channels, int32_t kind, int32_t ksize, int32_t batch, int64_t src_stride,
                                  int64_t dst_stride) {
    return gradient_impl("kh_gradient_magnitude_f32", stream, strided_batch(src, dst, batch, src_stride, dst_stride), cols, rows, channels, kind, ksize);
}
int32_t kh_gradient_magnitude_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t cols,
                                       int32_t rows, int32_t channels, int32_t kind, int32_t ksize) {
    return gradient_impl("kh_gradient_magnitude_f32_list", stream, listed_batch(srcs, dsts, n), cols, rows, channels, kind, ksize);
}

}  // extern "C"
