// u8 fixed-point twins for gfx950: Q8 separable blur (+ the 3x3 binomial), Q10 bilinear remap,
// warp_affine and warp_perspective.
//
// Device twins of P/cuda/filter.rs:116-250 and P/cuda/{remap,warp_affine_u8,warp_perspective_u8}.rs;
// the integer arithmetic is that of the CPU ops gaussian_blur_u8 / box_blur_u8 (P/filter/ops.rs:59,
// 639), remap_u8 (P/interpolation/remap.rs:157), warp_affine_u8 (P/warp/affine.rs:373) and
// warp_perspective_u8 (P/warp/perspective.rs:179): byte-identical results (tests/test_u8_gpu.py).
//
// Blur: the rolling-column structure of the f32 filter (kh_filter.hip) on bytes.  A lane owns 4
// consecutive flat bytes, a wave 256, and walks down a strip of rows; each input row is loaded once
// (one dword per lane + a 32-byte halo each side by 16 lanes), parked in a 320-byte wave-private LDS
// row where the replicate border is patched in, reduced horizontally from LDS (`(acc+128)>>8`, the
// reference's u8 intermediate) into a K-deep register ring, and reduced vertically from the ring.
// 1 read + 1 write of HBM per byte instead of the reference's two passes through a scratch image.
#include <math.h>

#include <algorithm>

#include "kh_common.h"

using namespace kh;

namespace {

// ---- blur --------------------------------------------------------------------------------------------

constexpr int kU8Halo = 32;         // halo bytes staged on each side of a wave's 256
constexpr int kU8Wave = 256;        // flat bytes per wave per row
constexpr int kU8Tile = 4 * kU8Wave;  // per 256-thread block
constexpr int kU8StripMax = 360;

struct U8FilterArgs {
    const uint8_t* src;
    uint8_t* dst;
    int rows, rowlen, cols;  // rowlen = cols * C bytes
    int th;                  // output rows per strip
    long long src_stride, dst_stride;
    XcdTiles tiles;
};
struct TapsQ { uint32_t k[16]; };

typedef uint32_t u32u __attribute__((aligned(1)));  // unaligned dword access (global_load/store_dword)

__device__ __forceinline__ uint32_t rhadd(uint32_t a, uint32_t b) { return (a + b + 1u) >> 1; }

// MODE 0: one u32 accumulator per byte (any taps).  MODE 1: 3x3 binomial of rounding halving adds.
// MODE 2: two bytes per register in 16-bit lanes — valid when the taps of each pass sum to <= 256, which
// quantize_kernel_256 guarantees for every gaussian / box kernel: a lane never exceeds 255*256 + 128 <
// 2^16, so `(b0 | b2 << 16) * k` accumulates both bytes with one 24-bit multiply-add and no carry.  The
// kernel is VALU-bound (r01m: 5.5 ms on 4K x 256 against a 1.6 ms memory floor); MODE 2 halves its
// arithmetic.  v_perm_b32 pulls the even / odd bytes of a tap's unaligned 4-byte window straight into lanes.
template <int K, int C, int MODE>
__global__ __launch_bounds__(kBlock) void blur_u8_roll_kernel(U8FilterArgs a, TapsQ kx, TapsQ ky) {
    constexpr bool BINOMIAL = MODE == 1;
    __shared__ uint32_t rowbuf[4][84];  // 80 dwords of row + a dummy slot
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int H = K / 2;
    constexpr int D = (H * C + 3) / 4;  // halo dwords actually read on each side
    static_assert(H * C <= kU8Halo, "halo does not fit");
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int gx0 = tx * kU8Tile + wv * kU8Wave;
    if (gx0 >= a.rowlen) return;  // whole wave idle (no block barrier below)
    const int y0 = ty * a.th;
    const uint8_t* __restrict__ src = a.src + (long long)bz * a.src_stride;
    uint8_t* __restrict__ dst = a.dst + (long long)bz * a.dst_stride;
    uint32_t* buf = rowbuf[wv];
    uint8_t* bufb = reinterpret_cast<uint8_t*>(buf);

    // Loads are unconditional, from clamped addresses (see kh_filter.hip): a dword that straddles the
    // end of the row is fetched from rowlen-4 and shifted down; bytes outside the row are garbage
    // until the replicate patch below overwrites them.
    const int g = gx0 + 4 * lane;
    const int gm = min(g, a.rowlen - 4);
    const int sm = min(g - gm, 3) * 8;
    const bool is_halo = lane < 16;
    const int gh_raw = lane < 8 ? gx0 - kU8Halo + 4 * lane : gx0 + kU8Wave + 4 * (lane - 8);
    const int gh = is_halo ? min(max(gh_raw, 0), a.rowlen - 4) : gm;
    const int sh = is_halo ? min(max(gh_raw - gh, 0), 3) * 8 : 0;
    const int hslot = is_halo ? (lane < 8 ? lane : 64 + lane) : 80;
    const bool left_edge = gx0 == 0, right_edge = gx0 + kU8Wave + kU8Halo > a.rowlen;
    const int nrows = min(a.th, a.rows - y0) + 2 * H;
    int pf_row = y0 - H;

    uint32_t qm[K], qh[K];
    auto prefetch = [&](uint32_t& m, uint32_t& hv) {
        const int base = min(max(pf_row, 0), a.rows - 1) * a.rowlen;  // replicate rows; 32-bit: host-checked
        m = *reinterpret_cast<const u32u*>(src + base + gm);
        hv = *reinterpret_cast<const u32u*>(src + base + gh);
        ++pf_row;
    };
#pragma unroll
    for (int p = 0; p < K; ++p) prefetch(qm[p], qh[p]);

    uint32_t ring[K][4];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) ring[i][b] = 0;

    const bool full = g + 3 < a.rowlen;
    int out_off = (y0 - 2 * H) * a.rowlen + g;
    const uint32_t* tap = buf + 8 + lane - D;
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            const int r = rb + p;
            const uint32_t m = qm[p] >> sm, hv = qh[p] >> sh;
            prefetch(qm[p], qh[p]);
            buf[8 + lane] = m;
            buf[hslot] = hv;
            __builtin_amdgcn_wave_barrier();
            // replicate border (P/cuda/filter.rs:131-139): positions left of pixel 0 / right of the last
            // pixel take that pixel's byte of the same channel.  Wave-uniform, LDS only.
            if (left_edge) {
                if (lane < kU8Halo) {
                    const int gg = lane - kU8Halo;              // flat index, negative
                    const int ch = ((gg % C) + C) % C;
                    bufb[lane] = bufb[kU8Halo + ch];
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (right_edge) {
                const int last = a.rowlen - C - gx0 + kU8Halo;  // LDS byte position of the last pixel
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int j = lane + 64 * i;
                    const int gg = gx0 - kU8Halo + j;
                    if (gg >= a.rowlen) bufb[j] = bufb[last + (gg % C)];
                }
                __builtin_amdgcn_wave_barrier();
            }
            uint32_t d[2 * D + 1];
#pragma unroll
            for (int i = 0; i < 2 * D + 1; ++i) d[i] = tap[i];
            __builtin_amdgcn_wave_barrier();  // row consumed before the next one overwrites it
            uint32_t packed = 0;
            if constexpr (MODE == 2) {
                uint32_t accE = 0x00800080u, accO = 0x00800080u;  // bytes (0, 2) and (1, 3) of this lane's dword, 16-bit lanes; + the rounding halves
#pragma unroll
                for (int t = 0; t < K; ++t) {
                    constexpr int base = 4 * D - H * C;       // byte offset of tap 0's window in d[]
                    const int rel = base + t * C, i = rel / 4, sft = rel % 4;  // compile-time after unrolling
                    const uint32_t lo = d[i], hi = d[i + 1 < 2 * D + 1 ? i + 1 : i];
                    const uint32_t selE = (uint32_t)sft | 0x0c00u | ((uint32_t)(sft + 2) << 16) | 0x0c000000u;
                    const uint32_t selO = (uint32_t)(sft + 1) | 0x0c00u | ((uint32_t)(sft + 3) << 16) | 0x0c000000u;
                    accE = mad24(__builtin_amdgcn_perm(hi, lo, selE), kx.k[t], accE);
                    accO = mad24(__builtin_amdgcn_perm(hi, lo, selO), kx.k[t], accO);
                }
                ring[p][0] = (accE >> 8) & 0x00ff00ffu;
                ring[p][1] = (accO >> 8) & 0x00ff00ffu;
                // every product goes through mad24 (one v_mad_u32_u24): left as `__umul24(..) + acc` the compiler turned four of the
                // fourteen vertical products into quarter-rate v_mul_lo_u32 and the rest into mul + add3 (r03 ISA review)
                uint32_t oE = 0x00800080u, oO = 0x00800080u;
#pragma unroll
                for (int i = 0; i < K; ++i) {  // oldest row first
                    oE = mad24(ring[(p + 1 + i) % K][0], ky.k[i], oE);
                    oO = mad24(ring[(p + 1 + i) % K][1], ky.k[i], oO);
                }
                oE = (oE >> 8) & 0x00ff00ffu;
                oO = (oO >> 8) & 0x00ff00ffu;
                packed = oE | (oO << 8);
            } else {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                auto byte_at = [&](int t) -> uint32_t {  // tap t of output byte b; offsets are compile-time
                    const int rel = b + (t - H) * C + 4 * D;
                    return (d[rel / 4] >> ((rel % 4) * 8)) & 0xffu;
                };
                if constexpr (BINOMIAL) {
                    ring[p][b] = rhadd(rhadd(byte_at(0), byte_at(1)), rhadd(byte_at(1), byte_at(2)));
                } else {
                    uint32_t acc = 128u;
#pragma unroll
                    for (int t = 0; t < K; ++t) acc = mad24(byte_at(t), kx.k[t], acc);   // byte x u8 tap: 24-bit operands
                    ring[p][b] = (acc >> 8) & 0xffu;  // `as u8`
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                uint32_t o;
                if constexpr (BINOMIAL) {
                    o = rhadd(rhadd(ring[(p + 1) % K][b], ring[(p + 2) % K][b]), rhadd(ring[(p + 2) % K][b], ring[p][b]));
                } else {
                    uint32_t acc = 128u;
#pragma unroll
                    for (int i = 0; i < K; ++i) acc = mad24(ring[(p + 1 + i) % K][b], ky.k[i], acc);  // oldest row first
                    o = (acc >> 8) & 0xffu;
                }
                packed |= o << (8 * b);
            }
            }  // MODE != 2
            if (r >= 2 * H && r < nrows) {
                if (full) {
                    *reinterpret_cast<u32u*>(dst + out_off) = packed;
                } else {
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        if (g + b < a.rowlen) dst[out_off + b] = (uint8_t)(packed >> (8 * b));
                }
            }
            out_off += a.rowlen;
        }
    }
}


// ---- RGB8 blur, planar in registers (round 3) --------------------------------------------------------------------------------
// blur_u8_roll_kernel above is completely VALU-bound (r03h: 72 vector instructions per 4 bytes, the vector ALUs 101 % busy): with
// interleaved RGB the horizontal taps of a byte sit 3 bytes apart, so every (two-byte, tap) product costs a v_perm_b32 to gather the
// bytes and a multiply-add — 7 instructions per byte for a 7-tap row pass.  De-interleaved, the taps of a channel are ADJACENT bytes
// and v_dot4_u32_u8 takes four of them per instruction:
//   * a lane owns FOUR PIXELS (12 bytes = 3 dwords, one 768-byte contiguous wave-load per row), de-interleaves them into one dword
//     per channel (6 v_perm_b32) and gets its left / right neighbours' channel dwords by two wave shifts per channel — no LDS
//     row buffer at all.  ALL 64 lanes store: a wave's row segment is 256 pixels = 768 bytes = whole 128-byte lines (the first
//     version kept lanes 0 / 63 as halo lanes: 744-byte segments, every boundary splitting a line between two waves — a pure
//     copy in that shape is 19 % slower, profiles/r03za).  The quads either side of the wave come from one more load per row
//     (the lower half's lanes all load the quad before the wave's first, the upper's the quad after its last), de-interleaved the
//     same way and handed to the end lanes as the fill value of the DPP wave shifts;
//   * horizontal pass per channel: the (up to 9) taps of output pixel j are bytes j + 4 - H ... of the 12-byte (prev, cur, next)
//     string, taken four at a time: 6 v_alignbyte_b32 + 8 v_dot4_u32_u8 per 4 outputs for K = 7 (3.5 instructions per byte
//     instead of 7), the rounding half in the accumulator operand;
//   * the four sums of a channel go into the 16-bit-lane pair form ((s0 >> 8) | (s2 >> 8) << 16 — one v_perm_b32 each, the sums are
//     < 2^16) that the vertical pass of the old kernel uses: K single-instruction 24-bit multiply-adds per pair;
//   * results are re-interleaved with 9 v_perm_b32 and leave as one 12-byte store.
// Replicate borders: rows by clamping the row index; columns (waves that touch the first / last pixel only, wave-uniform) by
// loading the quad from a clamped position and re-indexing its pixels with ONE per-lane byte selector per channel.
// Same integers as the reference's two u8 passes ((acc + 128) >> 8 after each): byte-identical to the old kernel (tests run both).
// For 3-channel images, 3..9 taps per axis whose quantised taps sum to <= 256 (every gaussian / box kernel), rows of >= 4 pixels.
constexpr int kRgbWavePx = 256;                 // output pixels per wave (64 lanes x 4)
constexpr int kRgbTilePx = 4 * kRgbWavePx;      // per 256-thread block

// per-byte rounding halving add of four packed bytes: (a + b + 1) >> 1 without leaving the byte
__device__ __forceinline__ uint32_t rhadd4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }
// BINOMIAL (round 6, K = 3): the reference's 3 x 3 [1 2 1] / 4 case — what a gaussian of 3 taps and sigma in [0.6, 1.2] is, the default
// sigma included — as rounding halving adds on the planar dwords (rhadd(rhadd(l, c), rhadd(c, r)) per pass, four pixels per
// instruction); it took the interleaved one-accumulator-per-byte kernel before and ran slower than the 5 x 5 gaussian.
// C = 4 (round 6): RGBA images took the interleaved kernel at 0.25-0.50 of peak; here a lane's quad is 16 bytes (one load, one store of whole
// pixels) de-interleaved by a 4 x 4 byte transpose (kh_common.h::deinterleave_quad); `plain`: kh_common.h::plain_row_stores, 2 = a destination off a dword.
template <int K, bool BINOMIAL = false, int C = 3>
__global__ __launch_bounds__(kBlock, C == 4 ? (K <= 3 ? 5 : (K <= 5 ? 4 : (K <= 7 ? 3 : 2))) : (K <= 7 ? 4 : 3)) void blur_u8_rgb_kernel(U8FilterArgs a, TapsQ kx, TapsQ ky, int plain) {   // K = 7: 131 -> 128 VGPRs keeps 4 waves per SIMD
    static_assert(!BINOMIAL || K == 3, "the binomial is 3 x 3");
    constexpr int H = K / 2, G = (K + 3) / 4;   // taps are consumed four at a time
    static_assert(K >= 3 && K <= 9 && (K & 1), "3..9 taps: one neighbour quad on each side covers the window");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int p0 = (int)tx * kRgbTilePx + wv * kRgbWavePx;   // first output pixel of this wave
    if (p0 >= a.cols) return;                               // whole wave idle (no block barrier below)
    const int y0 = ty * a.th;
    const uint8_t* __restrict__ src = a.src + (long long)bz * a.src_stride;
    uint8_t* __restrict__ dst = a.dst + (long long)bz * a.dst_stride;
    // block-uniform: quad offsets are multiples of four bytes and the image fits the V#'s 2 GiB window
    const bool stream_ok = (a.rowlen & 3) == 0 && (long long)a.rows * a.rowlen <= 0x7fffffffLL && plain != 2;
    const __amdgpu_buffer_rsrc_t out_win = stream_window(dst, (long long)a.rows * a.rowlen);
    const int p = p0 + 4 * lane;                            // this lane's quad: pixels p .. p + 3
    const int ph = lane < 32 ? p0 - 4 : p0 + kRgbWavePx;    // the wave's halo quads: left in the lower half's lanes, right in the upper's
    const bool edge = p0 < 4 || p0 + kRgbWavePx + 4 > a.cols;   // wave-uniform: some quad of the wave needs clamping
    const int pc = min(p, a.cols - 4), phc = min(max(ph, 0), a.cols - 4);   // where the quads are loaded from (cols >= 4: host-checked)
    // per-lane byte selector that re-indexes the loaded quad's pixels when the quad was clamped: pixel j <- loaded pixel
    // clamp(p + j, 0, cols - 1) - pc  (0x03020100 = identity)
    uint32_t esel = 0x03020100u, hsel = 0x03020100u;
    if (edge) {
        esel = hsel = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            esel |= (uint32_t)min(max(min(p + j, a.cols - 1) - pc, 0), 3) << (8 * j);
            hsel |= (uint32_t)min(max(min(max(ph + j, 0), a.cols - 1) - phc, 0), 3) << (8 * j);
        }
    }
    const bool writer = p < a.cols;
    const bool full = p + 3 < a.cols;
    const int nrows = min(a.th, a.rows - y0) + 2 * H;
    int pf_row = y0 - H;

    uint32_t wq[3];   // horizontal taps as bytes, four per dword (zero padded): tap t = byte t & 3 of wq[t >> 2]
#pragma unroll
    for (int g = 0; g < 3; ++g) wq[g] = kx.k[4 * g] | (kx.k[4 * g + 1] << 8) | (kx.k[4 * g + 2] << 16) | (kx.k[4 * g + 3] << 24);

    uint32_t q[K][2 * C];  // K rows of raw loads in flight per lane: its quad and its half-wave's halo quad
    auto prefetch = [&](uint32_t (&d)[2 * C]) {
        const uint8_t* rp = src + (long long)min(max(pf_row, 0), a.rows - 1) * a.rowlen;   // replicate rows
        const uint8_t *rq = rp + C * pc, *rh = rp + C * phc;
        if constexpr (C == 4) {
            const u32x4_t v = *reinterpret_cast<const u32x4_unaligned*>(rq), hq = *reinterpret_cast<const u32x4_unaligned*>(rh);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; d[4] = hq.x; d[5] = hq.y; d[6] = hq.z; d[7] = hq.w;
        } else {
            d[0] = *reinterpret_cast<const u32u*>(rq); d[1] = *reinterpret_cast<const u32u*>(rq + 4); d[2] = *reinterpret_cast<const u32u*>(rq + 8);
            d[3] = *reinterpret_cast<const u32u*>(rh); d[4] = *reinterpret_cast<const u32u*>(rh + 4); d[5] = *reinterpret_cast<const u32u*>(rh + 8);
        }
        ++pf_row;
    };
#pragma unroll
    for (int i = 0; i < K; ++i) prefetch(q[i]);

    uint32_t ring[K][C][2];  // [row][channel][even / odd pixel pair], 16-bit lanes
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) { ring[i][c][0] = 0; ring[i][c][1] = 0; }

    long long out_off = (long long)(y0 - 2 * H) * a.rowlen + C * p;
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const int r = rb + s;
            uint32_t dq[2 * C];
#pragma unroll
            for (int k = 0; k < 2 * C; ++k) dq[k] = q[s][k];
            prefetch(q[s]);
            // de-interleave: [R0 G0 B0 R1][G1 B1 R2 G2][B2 R3 G3 B3] -> one dword per channel, pixel j = byte j
            uint32_t cur[C], halo[C];
            deinterleave_quad<C>(dq, cur);
            deinterleave_quad<C>(dq + C, halo);
            if (edge) {   // wave-uniform
#pragma unroll
                for (int c = 0; c < C; ++c) { cur[c] = __builtin_amdgcn_perm(0u, cur[c], esel); halo[c] = __builtin_amdgcn_perm(0u, halo[c], hsel); }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const uint32_t prev = from_lane_below(cur[c], halo[c]), next = from_lane_above(cur[c], halo[c]);
                if constexpr (BINOMIAL) {
                    const uint32_t lft = __builtin_amdgcn_alignbyte(cur[c], prev, 3u), rgt = __builtin_amdgcn_alignbyte(next, cur[c], 1u);   // pixels p - 1 .. p + 2, p + 1 .. p + 4
                    ring[s][c][0] = rhadd4(rhadd4(lft, cur[c]), rhadd4(cur[c], rgt));
                    continue;
                }
                const uint32_t str[4] = {prev, cur[c], next, next};   // bytes 0..11 = pixels p - 4 .. p + 7 of this channel (+ a don't-care dword)
                uint32_t sum[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t acc = 128u;   // the reference's rounding half
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        constexpr int kBase = 4 - H;
                        const int off = kBase + j + 4 * g;   // compile-time after unrolling: first byte of this group of four taps
                        const uint32_t win = (off & 3) == 0 ? str[off >> 2] : __builtin_amdgcn_alignbyte(str[(off >> 2) + 1], str[off >> 2], (uint32_t)(off & 3));
                        acc = __builtin_amdgcn_udot4(win, wq[g], acc, false);
                    }
                    sum[j] = acc;   // < 2^16: the taps sum to <= 256
                }
                // (sum >> 8) of pixels (0, 2) and (1, 3) into 16-bit lanes: byte 1 of each sum
                ring[s][c][0] = __builtin_amdgcn_perm(sum[2], sum[0], 0x0c050c01u);
                ring[s][c][1] = __builtin_amdgcn_perm(sum[3], sum[1], 0x0c050c01u);
            }
            uint32_t pl[C];   // vertical pass, then one dword per channel again (pixel j = byte j)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (BINOMIAL) {   // rows oldest first: s + 1, s + 2, s (mod 3)
                    const uint32_t r0 = ring[(s + 1) % K][c][0], r1 = ring[(s + 2) % K][c][0], r2 = ring[s][c][0];
                    pl[c] = rhadd4(rhadd4(r0, r1), rhadd4(r1, r2));
                    continue;
                }
                uint32_t oe = 0x00800080u, oo = 0x00800080u;
#pragma unroll
                for (int i = 0; i < K; ++i) {   // oldest row first
                    oe = mad24(ring[(s + 1 + i) % K][c][0], ky.k[i], oe);
                    oo = mad24(ring[(s + 1 + i) % K][c][1], ky.k[i], oo);
                }
                pl[c] = __builtin_amdgcn_perm(oo, oe, 0x07030501u);   // (oe.b1, oo.b1, oe.b3, oo.b3) = pixels 0, 1, 2, 3
            }
            if (writer && r >= 2 * H && r < nrows) {
                uint32_t w[C];   // re-interleave: [R0 G0 B0 R1][G1 B1 R2 G2][B2 R3 G3 B3]
                interleave_quad<C>(pl, w);
                uint8_t* o = dst + out_off;
                if (full && stream_ok) {   // write-through non-temporal buffer store (kh_common.h::stream_store), or write-back on rows that are not whole lines
                    row_store<C>(out_win, (int)out_off, w, plain);
                } else if (full) {
#pragma unroll
                    for (int k = 0; k < C; ++k) *reinterpret_cast<u32u*>(o + 4 * k) = w[k];
                } else {
#pragma unroll
                    for (int b = 0; b < 3 * C; ++b)   // at most three pixels of a quad that reaches past the last column
                        if (p + b / C < a.cols) o[b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
                }
            }
            out_off += a.rowlen;
        }
    }
}

// ---- single-channel blur, rolling wave (round 6) -----------------------------------------------------------------------------------
// Gray u8 images (what feature detectors smooth) took blur_u8_roll_kernel at 0.33-0.53 of peak; one plane needs no de-interleaving at
// all.  The walk of the RGB kernel above with a lane owning SIXTEEN pixels of a row (one 16-byte load and store, 1 KiB per wave and
// row): each of its four dwords is the RGB kernel's channel dword — the same v_dot4_u32_u8 row pass on the twelve-byte string
// (previous dword | this dword | next dword), the same 16-bit-lane pair form and 24-bit multiply-add column pass, the binomial's
// rounding halving adds — with the dword before the lane's first and after its last from the neighbouring lanes by wave shifts and,
// at the ends of the wave, from one halo dword per half-wave.  Replicate borders: rows by clamping the row index, columns (edge
// waves) by re-indexing.  RAGGED: any width >= 16 and any alignment (kh_common.h::remap16_*; plain = 2: unaligned global stores).
// Same integers as the other kernels: byte-identical (tests run both).  3..9 taps per axis whose quantised taps sum to <= 256.
constexpr int kGrayWavePx = 1024, kGrayTilePx = 4 * kGrayWavePx;
// K = 11 / 13 / 15: the window reaches seven pixels either side, so the string around a dword is (two before | it | two after) and a half-wave's
// halo is eight bytes; 200-250 registers (two waves per SIMD) — the K rows of loads in flight per lane still cover the latency.
template <int K, bool BINOMIAL, bool RAGGED>
__global__ __launch_bounds__(kBlock, K <= 3 ? 6 : (K <= 5 ? 5 : (K <= 7 ? 4 : (K <= 9 ? 3 : (K == 15 && RAGGED ? 1 : 2))))) void blur_u8_gray_kernel(U8FilterArgs a, TapsQ kx, TapsQ ky, int plain) {
    static_assert(!BINOMIAL || K == 3, "the binomial is 3 x 3");
    constexpr int H = K / 2, G = (K + 3) / 4;   // taps are consumed four at a time
    constexpr bool WIDE = K > 9;                // two neighbour dwords on each side
    constexpr int NH = WIDE ? 2 : 1, HP = 4 * NH;   // halo dwords / pixels per half-wave
    static_assert(K >= 3 && K <= 15 && (K & 1), "3..15 taps");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned tx, ty, bz;
    if (!xcd_tile(a.tiles, tx, ty, bz)) return;
    const int p0 = (int)tx * kGrayTilePx + wv * kGrayWavePx;   // first output pixel of this wave
    if (p0 >= a.cols) return;                                 // whole wave idle (no block barrier below)
    const int y0 = ty * a.th;
    const uint8_t* __restrict__ src = a.src + (long long)bz * a.src_stride;
    uint8_t* __restrict__ dst = a.dst + (long long)bz * a.dst_stride;
    const __amdgpu_buffer_rsrc_t out_win = stream_window(dst, (long long)a.rows * a.cols);   // (rows * cols < 2^31: host-checked)
    const int p = p0 + 16 * lane;                             // this lane's sixteen pixels
    const int nvalid = min(max(a.cols - p, 0), 16);           // RAGGED: 0 .. 16; otherwise all sixteen or none (cols % 16 == 0: host-checked)
    const int ph = lane < 32 ? p0 - HP : p0 + kGrayWavePx;    // the wave's halo dwords: left in the lower half's lanes, right in the upper's
    const bool edge = p0 < HP || p0 + kGrayWavePx + HP > a.cols;   // wave-uniform
    const int pc = min(p, a.cols - 16), phc = min(max(ph, 0), a.cols - HP);   // cols >= 16: host-checked
    uint32_t hsel = 0x03020100u, hsel2 = 0x07060504u;   // halo byte j <- loaded halo byte clamp(ph + j) - phc (two dwords when WIDE)
    Remap16 rm{};
    if (edge) {
        hsel = hsel2 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hsel |= (uint32_t)min(max(min(max(ph + j, 0), a.cols - 1) - phc, 0), HP - 1) << (8 * j);
            hsel2 |= (uint32_t)min(max(min(max(ph + 4 + j, 0), a.cols - 1) - phc, 0), HP - 1) << (8 * j);
        }
        if constexpr (RAGGED) rm = remap16_setup(p, pc, [&](int x) { return min(x, a.cols - 1); });
    }
    const int nrows = min(a.th, a.rows - y0) + 2 * H;
    int pf_row = y0 - H;

    uint32_t wq[4];   // horizontal taps as bytes, four per dword (zero padded): tap t = byte t & 3 of wq[t >> 2]
#pragma unroll
    for (int g = 0; g < 4; ++g) wq[g] = kx.k[4 * g] | (kx.k[4 * g + 1] << 8) | (kx.k[4 * g + 2] << 16) | (kx.k[4 * g + 3] << 24);

    uint32_t q[K][4 + NH];  // K rows of raw loads in flight per lane: its sixteen pixels and its half-wave's halo dword(s)
    auto prefetch = [&](uint32_t (&d)[4 + NH]) {
        const uint8_t* rp = src + (long long)min(max(pf_row, 0), a.rows - 1) * a.cols;   // replicate rows
        const u32x4_t v = *reinterpret_cast<const u32x4_unaligned*>(rp + pc);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        d[4] = *reinterpret_cast<const u32u*>(rp + phc);
        if constexpr (WIDE) d[5] = *reinterpret_cast<const u32u*>(rp + phc + 4);
        ++pf_row;
    };
#pragma unroll
    for (int i = 0; i < K; ++i) prefetch(q[i]);

    uint32_t ring[K][4][2];  // [row][dword][even / odd pixel pair], 16-bit lanes (binomial: [0] = the packed row)
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) { ring[i][c][0] = 0; ring[i][c][1] = 0; }

    int out_off = (y0 - 2 * H) * a.cols + p;
    for (int rb = 0; rb < nrows; rb += K) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const int r = rb + s;
            uint32_t cur[4] = {q[s][0], q[s][1], q[s][2], q[s][3]}, halo = q[s][4], halo2 = WIDE ? q[s][4 + NH - 1] : 0u;
            prefetch(q[s]);
            if (edge) {   // wave-uniform
                if constexpr (RAGGED) remap16_apply(rm, cur, 0u);
                else if (!nvalid) { cur[0] = __builtin_amdgcn_perm(0u, cur[3], 0x03030303u); cur[1] = cur[0]; }   // a lane past the row end: the row's last pixel, replicated
                if constexpr (WIDE) { const uint32_t h0 = halo, h1 = halo2; halo = __builtin_amdgcn_perm(h1, h0, hsel); halo2 = __builtin_amdgcn_perm(h1, h0, hsel2); }
                else halo = __builtin_amdgcn_perm(0u, halo, hsel);
            }
            // (WIDE: lane 0's fills are its halo dwords (p0 - 8 .., p0 - 4 ..), lane 63's (p0 + 1024 .., p0 + 1028 ..))
            const uint32_t prevd = from_lane_below(cur[3], WIDE ? halo2 : halo), nextd = from_lane_above(cur[0], halo);
            const uint32_t prevd2 = WIDE ? from_lane_below(cur[2], halo) : 0u, nextd2 = WIDE ? from_lane_above(cur[1], halo2) : 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t prev = c == 0 ? prevd : cur[c - 1], next = c == 3 ? nextd : cur[c + 1];
                if constexpr (BINOMIAL) {
                    const uint32_t lft = __builtin_amdgcn_alignbyte(cur[c], prev, 3u), rgt = __builtin_amdgcn_alignbyte(next, cur[c], 1u);
                    ring[s][c][0] = rhadd4(rhadd4(lft, cur[c]), rhadd4(cur[c], rgt));
                    continue;
                }
                const uint32_t prev2 = c >= 2 ? cur[c >= 2 ? c - 2 : 0] : (c == 1 ? prevd : prevd2), next2 = c <= 1 ? cur[c <= 1 ? c + 2 : 3] : (c == 2 ? nextd : nextd2);
                // bytes 0.. = pixels -4 (WIDE: -8) .. of this dword (+ a don't-care dword)
                const uint32_t str[6] = {WIDE ? prev2 : prev, WIDE ? prev : cur[c], WIDE ? cur[c] : next, WIDE ? next : next, next2, next2};
                uint32_t sum[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t acc = 128u;   // the reference's rounding half
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        constexpr int kBase = (WIDE ? 8 : 4) - H;
                        const int off = kBase + j + 4 * g;   // compile-time after unrolling: first byte of this group of four taps
                        const uint32_t win = (off & 3) == 0 ? str[off >> 2] : __builtin_amdgcn_alignbyte(str[(off >> 2) + 1], str[off >> 2], (uint32_t)(off & 3));
                        acc = __builtin_amdgcn_udot4(win, wq[g], acc, false);
                    }
                    sum[j] = acc;   // < 2^16: the taps sum to <= 256
                }
                ring[s][c][0] = __builtin_amdgcn_perm(sum[2], sum[0], 0x0c050c01u);   // (sum >> 8) of pixels (0, 2) in 16-bit lanes
                ring[s][c][1] = __builtin_amdgcn_perm(sum[3], sum[1], 0x0c050c01u);   // and of pixels (1, 3)
            }
            uint32_t pl[4];   // vertical pass, four pixels per dword again
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (BINOMIAL) {   // rows oldest first: s + 1, s + 2, s (mod 3)
                    const uint32_t r0 = ring[(s + 1) % K][c][0], r1 = ring[(s + 2) % K][c][0], r2 = ring[s][c][0];
                    pl[c] = rhadd4(rhadd4(r0, r1), rhadd4(r1, r2));
                    continue;
                }
                uint32_t oe = 0x00800080u, oo = 0x00800080u;
#pragma unroll
                for (int i = 0; i < K; ++i) {   // oldest row first
                    oe = mad24(ring[(s + 1 + i) % K][c][0], ky.k[i], oe);
                    oo = mad24(ring[(s + 1 + i) % K][c][1], ky.k[i], oo);
                }
                pl[c] = __builtin_amdgcn_perm(oo, oe, 0x07030501u);   // (oe.b1, oo.b1, oe.b3, oo.b3) = pixels 0, 1, 2, 3
            }
            if (r >= 2 * H && r < nrows) {
                if (nvalid == 16 && (!RAGGED || plain != 2)) row_store<4>(out_win, out_off, pl, plain);
                else if (nvalid == 16) *reinterpret_cast<u32x4_unaligned*>(dst + out_off) = u32x4_t{pl[0], pl[1], pl[2], pl[3]};
                else if (RAGGED && nvalid > 0) store_head_bytes(dst + out_off, pl, nvalid);
            }
            out_off += a.cols;
        }
    }
}

// Fallback for what the rolling kernel does not take (kernels wider than 15 taps, halos beyond 32
// bytes, rows shorter than 4 bytes): one Q8 pass per launch through a scratch image, one thread per
// byte — the reference's own structure (P/cuda/filter.rs:116-165).
struct Taps64 { uint8_t k[64]; int n; };
template <bool HORIZ, bool BINOMIAL>
__global__ __launch_bounds__(kBlock) void blur_u8_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                              int cols, int rows, int C, long long ss, long long ds,
                                                              Taps64 k) {
    const int rowlen = cols * C;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (long long)rows * rowlen) return;
    const int y = (int)(i / rowlen), f = (int)(i % rowlen), x = f / C, ch = f % C;
    const uint8_t* s = src + (long long)blockIdx.y * ss;
    const int half = k.n / 2;
    auto at = [&](int t) -> uint32_t {
        const int xx = HORIZ ? min(max(x + t - half, 0), cols - 1) : x;
        const int yy = HORIZ ? y : min(max(y + t - half, 0), rows - 1);
        return s[((long long)yy * cols + xx) * C + ch];
    };
    uint32_t o;
    if constexpr (BINOMIAL) {
        o = rhadd(rhadd(at(0), at(1)), rhadd(at(1), at(2)));
    } else {
        uint32_t acc = 0;
        for (int t = 0; t < k.n; ++t) acc += at(t) * k.k[t];
        o = (acc + 128u) >> 8;
    }
    dst[(long long)blockIdx.y * ds + i] = (uint8_t)o;
}

// The same pass with FOUR consecutive bytes of a row per thread (round 6; rows of a multiple of four bytes): the four source bytes of
// tap t sit in ONE dword — at byte offset (t - half) * C from the thread's own (horizontal pass; unaligned dword loads) or in row
// y + t - half at the thread's column (vertical pass) — so a tap costs one load and, with the taps of a pass summing to <= 256
// (quantize_kernel_256), two 24-bit multiply-adds on byte pairs held in 16-bit lanes (SWAR, as blur_u8_roll_kernel's MODE 2) instead of
// four byte loads and four multiply-adds; the result leaves as one dword.  Threads whose horizontal taps reach over a row end (the
// replicated border) take the per-byte expression.  Gaussians of 17-31 taps: 12-19 ms -> 1.1-1.6 ms per 32 4K images
// (profiles/r06zi_blur_u8_wide.txt).
template <bool HORIZ, bool SWAR, int C>
__global__ __launch_bounds__(kBlock) void blur_u8_pass4_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                               int cols, int rows, long long ss, long long ds, Taps64 k) {
    const int rowlen = cols * C, nq = rowlen >> 2;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (long long)rows * nq) return;
    const int y = (int)(i / nq), f0 = 4 * (int)(i - (long long)y * nq);
    const uint8_t* s = src + (long long)blockIdx.y * ss;
    const int half = k.n / 2;
    uint32_t out;
    const bool interior = !HORIZ || (f0 >= half * C && f0 + 3 + half * C < rowlen);
    if (interior) {
        uint32_t a02 = 0, a13 = 0, acc[4] = {0, 0, 0, 0};
        for (int t = 0; t < k.n; ++t) {
            const long long off = HORIZ ? (long long)y * rowlen + f0 + (t - half) * C : (long long)min(max(y + t - half, 0), rows - 1) * rowlen + f0;
            const uint32_t d = *reinterpret_cast<const u32u*>(s + off);
            const uint32_t w = k.k[t];
            if constexpr (SWAR) {
                a02 += (d & 0x00ff00ffu) * w;
                a13 += ((d >> 8) & 0x00ff00ffu) * w;
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b] += ((d >> (8 * b)) & 0xffu) * w;
            }
        }
        if constexpr (SWAR) out = (((a02 + 0x00800080u) >> 8) & 0x00ff00ffu) | ((a13 + 0x00800080u) & 0xff00ff00u);
        else out = (((acc[0] + 128u) >> 8) & 0xffu) | ((((acc[1] + 128u) >> 8) & 0xffu) << 8) | ((((acc[2] + 128u) >> 8) & 0xffu) << 16) | ((((acc[3] + 128u) >> 8) & 0xffu) << 24);
    } else {   // a horizontal window over the replicated border: blur_u8_pass_kernel's expression per byte
        out = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int f = f0 + b, x = f / C, ch = f - x * C;
            uint32_t acc = 0;
            for (int t = 0; t < k.n; ++t) acc += (uint32_t)s[((long long)y * cols + min(max(x + t - half, 0), cols - 1)) * C + ch] * k.k[t];
            out |= (((acc + 128u) >> 8) & 0xffu) << (8 * b);
        }
    }
    *reinterpret_cast<uint32_t*>(dst + (long long)blockIdx.y * ds + 4 * i) = out;
}

// The vertical pass with R consecutive rows per thread (a sliding window down one dword column): K + R - 1 row loads for R outputs
// instead of K each — the per-row form above re-read every row K times through the L2.  Byte pairs in 16-bit lanes (tap sums <= 256).
constexpr int kPassRows = 8;
template <int C_UNUSED>
__global__ __launch_bounds__(kBlock) void blur_u8_vpass_rows_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                    int rowlen, int rows, long long ss, long long ds, Taps64 k) {
    constexpr int R = kPassRows;
    const int nq = rowlen >> 2;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int bands = (rows + R - 1) / R;
    if (i >= (long long)bands * nq) return;
    const int yb = (int)(i / nq), f0 = 4 * (int)(i - (long long)yb * nq), y0 = yb * R;
    const uint8_t* s = src + (long long)blockIdx.y * ss + f0;
    const int half = k.n / 2;
    uint32_t a02[R], a13[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { a02[r] = 0; a13[r] = 0; }
    for (int j = 0; j < k.n + R - 1; ++j) {   // source row y0 - half + j feeds output row y0 + r with tap t = j - r
        const uint32_t d = *reinterpret_cast<const u32u*>(s + (long long)min(max(y0 - half + j, 0), rows - 1) * rowlen);
        const uint32_t e = d & 0x00ff00ffu, o = (d >> 8) & 0x00ff00ffu;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = j - r;
            const uint32_t w = (t >= 0 && t < k.n) ? (uint32_t)k.k[t] : 0u;   // wave-uniform
            a02[r] += e * w;
            a13[r] += o * w;
        }
    }
    uint8_t* d0 = dst + (long long)blockIdx.y * ds + f0;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (y0 + r < rows)
            *reinterpret_cast<uint32_t*>(d0 + (long long)(y0 + r) * rowlen) = (((a02[r] + 0x00800080u) >> 8) & 0x00ff00ffu) | ((a13[r] + 0x00800080u) & 0xff00ff00u);
}

template <int K, int C>
void launch_blur_kc(hipStream_t st, bool binomial, const U8FilterArgs& a, const TapsQ& kx, const TapsQ& ky) {
    const dim3 grid = xcd_grid(a.tiles);
    if constexpr (K == 3) {
        if (binomial) {
            hipLaunchKernelGGL((blur_u8_roll_kernel<3, C, 1>), grid, dim3(kBlock), 0, st, a, kx, ky);
            return;
        }
    }
    unsigned sx = 0, sy = 0;
    for (int i = 0; i < 16; ++i) { sx += kx.k[i]; sy += ky.k[i]; }
    const bool no_swar = dev_opt(kOptU8BlurSwar) == 0;  // test option: the plain-integer kernel (the fallback for tap sums above 256)
    if (sx <= 256 && sy <= 256 && !no_swar) hipLaunchKernelGGL((blur_u8_roll_kernel<K, C, 2>), grid, dim3(kBlock), 0, st, a, kx, ky);
    else hipLaunchKernelGGL((blur_u8_roll_kernel<K, C, 0>), grid, dim3(kBlock), 0, st, a, kx, ky);
}
template <int K>
void launch_blur_k(hipStream_t st, int C, bool binomial, const U8FilterArgs& a, const TapsQ& kx, const TapsQ& ky) {
    if (C == 1) launch_blur_kc<K, 1>(st, binomial, a, kx, ky);
    else if (C == 3) launch_blur_kc<K, 3>(st, binomial, a, kx, ky);
    else launch_blur_kc<K, 4>(st, binomial, a, kx, ky);
}

void pad_q(TapsQ& out, const uint8_t* k, int n, int K) {
    const int off = (K - n) / 2;  // zero taps add nothing to the integer accumulator
    for (int i = 0; i < 16; ++i) out.k[i] = (i >= off && i < off + n) ? k[i - off] : 0u;
}

int32_t launch_blur_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int cols, int rows, int C,
                       const uint8_t* qx, int nx, const uint8_t* qy, int ny, bool binomial, int batch, int64_t ss,
                       int64_t ds, const char* what) {
    const int rowlen = cols * C;
    const int kmax = nx > ny ? nx : ny;
    hipStream_t st = as_hip(stream);
    if (kmax <= 15 && (kmax / 2) * C <= kU8Halo && rowlen >= 4) {
        const int K = kmax < 3 ? 3 : kmax;
        TapsQ px, py;
        pad_q(px, qx, nx, K);
        pad_q(py, qy, ny, K);
        U8FilterArgs a;
        a.src = src; a.dst = dst; a.rows = rows; a.rowlen = rowlen; a.cols = cols;
        a.src_stride = ss; a.dst_stride = ds;
        unsigned sxq = 0, syq = 0;
        for (int i = 0; i < 16; ++i) { sxq += px.k[i]; syq += py.k[i]; }
        const bool rgb_off = dev_opt(kOptU8BlurRgb) == 0;   // test option: the interleaved kernel (what the other channel counts take)
        const bool rgb = (C == 3 || C == 4) && K <= 9 && (binomial ? K == 3 : (sxq <= 256 && syq <= 256)) && cols >= 4 && !rgb_off;   // (C = 4: round 6)
        // one channel, 3..9 taps: the rolling gray kernel (round 6; the same test option keeps the interleaved kernel)
        const bool gray = C == 1 && K <= 15 && (binomial ? K == 3 : (sxq <= 256 && syq <= 256)) && cols >= 16 && (int64_t)rows * cols <= kI32Max && !rgb_off;   // (11-15 taps: round 6, later)
        const bool gray_dword_ok = cols % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (batch <= 1 || ds % 4 == 0);   // buffer stores need dword-aligned rows
        const bool gray_ragged = cols % 16 != 0 || !gray_dword_ok;
        const unsigned tiles_x = rgb ? cdiv(cols, kRgbTilePx) : (gray ? cdiv(cols, kGrayTilePx) : cdiv(rowlen, kU8Tile));
        const long long cols_blocks = (long long)tiles_x * batch;
        long long strips = (2048 + cols_blocks - 1) / cols_blocks;  // >= 8 blocks per CU
        const long long min_strips = cdiv(rows, kU8StripMax), max_strips = cdiv(rows, 32);
        strips = strips < min_strips ? min_strips : (strips > max_strips ? max_strips : strips);
        a.th = (int)cdiv(rows, strips);
        a.tiles = xcd_tiles(tiles_x, cdiv(rows, a.th), (unsigned)batch, kXcdEighth);
        KH_REQUIRE(a.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
        if (gray) {
            const dim3 grid = xcd_grid(a.tiles);
            const int plain = gray_dword_ok ? plain_row_stores((int64_t)cols, dst, ds, batch) : 2;
#define KH_GB(KK, BIN) do { if (gray_ragged) hipLaunchKernelGGL((blur_u8_gray_kernel<KK, BIN, true>), grid, dim3(kBlock), 0, st, a, px, py, plain); \
                            else hipLaunchKernelGGL((blur_u8_gray_kernel<KK, BIN, false>), grid, dim3(kBlock), 0, st, a, px, py, plain); } while (0)
            switch (K) {
                case 3: if (binomial) KH_GB(3, true); else KH_GB(3, false); break;
                case 5: KH_GB(5, false); break;
                case 7: KH_GB(7, false); break;
                case 9: KH_GB(9, false); break;
                case 11: KH_GB(11, false); break;
                case 13: KH_GB(13, false); break;
                default: KH_GB(15, false); break;
            }
#undef KH_GB
            return check_launch(what);
        }
        if (rgb) {
            const dim3 grid = xcd_grid(a.tiles);
            // RGB keeps the streaming stores on every width (profiles/r06zr: it does not gain from write-back); RGBA: the row rule, 2 = a destination off a dword
            const bool dword_ok = reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (batch <= 1 || ds % 4 == 0);
            const int plain = C == 3 ? 0 : (dword_ok ? plain_row_stores((int64_t)rowlen, dst, ds, batch) : 2);
#define KH_RB(KK, BIN) do { if (C == 4) hipLaunchKernelGGL((blur_u8_rgb_kernel<KK, BIN, 4>), grid, dim3(kBlock), 0, st, a, px, py, plain); \
                            else hipLaunchKernelGGL((blur_u8_rgb_kernel<KK, BIN, 3>), grid, dim3(kBlock), 0, st, a, px, py, plain); } while (0)
            switch (K) {
                case 3: if (binomial) KH_RB(3, true); else KH_RB(3, false); break;
                case 5: KH_RB(5, false); break;
                case 7: KH_RB(7, false); break;
                default: KH_RB(9, false); break;
            }
#undef KH_RB
            return check_launch(what);
        }
        switch (K) {
            case 3: launch_blur_k<3>(st, C, binomial, a, px, py); break;
            case 5: launch_blur_k<5>(st, C, false, a, px, py); break;
            case 7: launch_blur_k<7>(st, C, false, a, px, py); break;
            case 9: launch_blur_k<9>(st, C, false, a, px, py); break;
            case 11: launch_blur_k<11>(st, C, false, a, px, py); break;
            case 13: launch_blur_k<13>(st, C, false, a, px, py); break;
            default: launch_blur_k<15>(st, C, false, a, px, py); break;
        }
        return check_launch(what);
    }
    // two passes through stream-ordered scratch (P/filter/cuda.rs:119)
    const size_t img = (size_t)rows * rowlen;
    Scratch scratch;
    if (int32_t rc = get_scratch(stream, img * batch, what, scratch)) return rc;
    uint8_t* tmp = scratch.as<uint8_t>();
    Taps64 tx{}, tyv{};
    tx.n = nx; tyv.n = ny;
    for (int i = 0; i < nx; ++i) tx.k[i] = qx[i];
    for (int i = 0; i < ny; ++i) tyv.k[i] = qy[i];
    unsigned sum_x = 0, sum_y = 0;
    for (int i = 0; i < nx; ++i) sum_x += qx[i];
    for (int i = 0; i < ny; ++i) sum_y += qy[i];
    // four bytes per thread (test option u8_blur_swar = 0: multiply-adds per byte; 2: the one-byte-per-thread kernel)
    const int swar_opt = dev_opt(kOptU8BlurSwar);
    if (!binomial && rowlen % 4 == 0 && swar_opt != 2 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 && reinterpret_cast<uintptr_t>(tmp) % 4 == 0 &&
        (batch <= 1 || ds % 4 == 0) && img % 4 == 0) {
        const dim3 g4(cdiv((int64_t)(img / 4), kBlock), (unsigned)batch);
#define KH_P4(H, S, CC, SRC, DST, SS, DS, T) hipLaunchKernelGGL((blur_u8_pass4_kernel<H, S, CC>), g4, dim3(kBlock), 0, st, SRC, DST, cols, rows, (long long)(SS), (long long)(DS), T)
#define KH_P4_C(H, S, SRC, DST, SS, DS, T) do { if (C == 1) KH_P4(H, S, 1, SRC, DST, SS, DS, T); else if (C == 3) KH_P4(H, S, 3, SRC, DST, SS, DS, T); else KH_P4(H, S, 4, SRC, DST, SS, DS, T); } while (0)
        if (sum_x <= 256 && swar_opt != 0) KH_P4_C(true, true, src, tmp, ss, img, tx); else KH_P4_C(true, false, src, tmp, ss, img, tx);
        if (sum_y <= 256 && swar_opt != 0 && swar_opt != 3) {   // R rows per thread (test option u8_blur_swar = 3: one row per thread)
            const dim3 gv(cdiv((int64_t)cdiv(rows, kPassRows) * (rowlen / 4), kBlock), (unsigned)batch);
            hipLaunchKernelGGL((blur_u8_vpass_rows_kernel<0>), gv, dim3(kBlock), 0, st, (const uint8_t*)tmp, dst, rowlen, rows, (long long)img, (long long)ds, tyv);
        } else if (sum_y <= 256 && swar_opt != 0) KH_P4_C(false, true, (const uint8_t*)tmp, dst, img, ds, tyv);
        else KH_P4_C(false, false, (const uint8_t*)tmp, dst, img, ds, tyv);
#undef KH_P4_C
#undef KH_P4
        return check_launch(what);
    }
    const dim3 grid(cdiv((int64_t)img, kBlock), (unsigned)batch);
    if (binomial) {
        hipLaunchKernelGGL((blur_u8_pass_kernel<true, true>), grid, dim3(kBlock), 0, st, src, tmp, cols, rows, C,
                           (long long)ss, (long long)img, tx);
        hipLaunchKernelGGL((blur_u8_pass_kernel<false, true>), grid, dim3(kBlock), 0, st, (const uint8_t*)tmp, dst, cols,
                           rows, C, (long long)img, (long long)ds, tyv);
    } else {
        hipLaunchKernelGGL((blur_u8_pass_kernel<true, false>), grid, dim3(kBlock), 0, st, src, tmp, cols, rows, C,
                           (long long)ss, (long long)img, tx);
        hipLaunchKernelGGL((blur_u8_pass_kernel<false, false>), grid, dim3(kBlock), 0, st, (const uint8_t*)tmp, dst, cols,
                           rows, C, (long long)img, (long long)ds, tyv);
    }
    return check_launch(what);
}

// `two_ok`: the Q10 gathers take 2-channel images too (the reference instantiates them per channel count,
// P/cuda/warp_affine_u8.rs:36-60, and tests C = 2: P/warp/cuda.rs:458-494); the blurs are built for 1, 3, 4.
int32_t check_u8_img(const char* what, const void* src, const void* dst, int sw, int sh, int dw, int dh, int channels,
                     int batch, int64_t ss, int64_t ds, bool two_ok = false) {
    KH_REQUIRE(sw > 0 && sh > 0 && dw > 0 && dh > 0, KH_ERR_INVALID_ARG, "%s: zero-sized image (src %dx%d, dst %dx%d)",
               what, sw, sh, dw, dh);
    KH_REQUIRE(channels == 1 || channels == 3 || channels == 4 || (two_ok && channels == 2), KH_ERR_UNSUPPORTED,
               "%s: no device kernel for %d channels (supported: %s)", what, channels, two_ok ? "1, 2, 3, 4" : "1, 3, 4");
    KH_REQUIRE(batch >= 0 && batch <= 65535, KH_ERR_TOO_LARGE, "%s: batch %d outside [0, 65535]", what, batch);
    KH_REQUIRE((int64_t)sw * sh * channels <= kI32Max && (int64_t)dw * dh * channels <= kI32Max, KH_ERR_TOO_LARGE,
               "%s: image exceeds 32-bit indexing", what);
    KH_REQUIRE(ss >= 0 && ds >= 0, KH_ERR_INVALID_ARG, "%s: negative batch stride", what);
    if (batch > 0) KH_REQUIRE(src && dst, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    return KH_OK;
}

// ---- Q10 bilinear gathers ----------------------------------------------------------------------------

constexpr int kBx = 64, kBy = 4;

struct ImgU8 {
    const uint8_t* src;
    uint8_t* dst;
    int sw, sh, dw, dh;
    long long src_stride, dst_stride;  // bytes between consecutive images
    XcdTiles tiles;
};

// A destination pixel as a packed word (channel c = bits [8c, 8c+8)).  RGB8 rows are written with dword
// stores: in a full wave the 64 pixels are 192 contiguous bytes = 48 dwords, assembled with two
// cross-lane reads per lane instead of three byte-granular store instructions per pixel (the gathers are
// bound by vector-memory instructions).  `wave_full` must be wave-uniform and all 64 lanes must call.
template <int C>
__device__ __forceinline__ void store_px_u8(uint8_t* o, uint32_t px, bool wave_full) {
    if constexpr (C == 4) {
        *reinterpret_cast<uint32_t*>(o) = px;  // pixel-aligned
    } else if constexpr (C == 3) {
        if (wave_full) {
            const int lane = threadIdx.x;  // kBx == 64: one wave per tile row
            const int a = (4 * lane) / 3, off = (4 * lane) % 3;
            const uint32_t pa = (uint32_t)__shfl((int)px, a), pb = (uint32_t)__shfl((int)px, min(a + 1, 63));
            const uint64_t w = (uint64_t)pa | ((uint64_t)pb << 24);
            if (lane < 48) *reinterpret_cast<u32_unaligned*>(o + lane) = (uint32_t)(w >> (8 * off));  // row base + 4*lane
        } else {
            o[0] = (uint8_t)px; o[1] = (uint8_t)(px >> 8); o[2] = (uint8_t)(px >> 16);
        }
    } else if constexpr (C == 2) {
        *reinterpret_cast<u16_unaligned*>(o) = (uint16_t)px;
    } else {
        o[0] = (uint8_t)px;
    }
}

// The Q10 blend of bilinear_sample_u8_valid (P/warp/common.rs:79-165) on four packed taps (channel c of a tap = bits [8c, 8c + 8)):
//     ((p00*fx1 + p01*fx) * fy1 + (p10*fx1 + p11*fx) * fy + 2^19) >> 20,     fx1 = 1024 - fx, fy1 = 1024 - fy
// with exact integer intermediates (< 2^28).  Round 2 spent ~12 VALU instructions per channel on it (four byte extracts, four
// 24-bit multiply-adds, shift, mask, insert) and the staged affine warp was VALU-bound at 67.6 instructions per pixel
// (profiles/r02q).  Now, per channel:
//   * ONE v_perm_b32 per tap row puts the row's two taps into the 16-bit halves of a dword (selector bytes 0x0c = constant zero);
//   * ONE v_dot2_u32_u16 with (fx1, fx) gives the horizontal sum  top = p00*fx1 + p01*fx  (<= 255 * 1024);
//   * two v_mad_u32_u24 do the vertical sum with the weights pre-multiplied by 16, so the result byte lands in bits [31:24]:
//     (X * 16) >> 24 == X >> 20 for X < 2^28;
// and the channels' top bytes are gathered by one or two more v_perm_b32 — six instructions per channel plus two to pack.  The
// integer is the reference's, operand for operand, so every caller stays byte-exact.
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
// fxp_bits = (1024 - fx) | fx << 16, fy16 = 16 * fy: prepared by the caller (the staged gather keeps them per pixel across images)
template <int C>
__device__ __forceinline__ uint32_t blend_q10_w(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, uint32_t fxp_bits, uint32_t fy16) {
    const u16x2_t fxp = __builtin_bit_cast(u16x2_t, fxp_bits);
    const uint32_t fy1_16 = 16384u - fy16;
    uint32_t acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const uint32_t sel = 0x0c000c00u | (uint32_t)c | ((uint32_t)(4 + c) << 16);   // (low tap . c, 0, high tap . c, 0)
        const u16x2_t tp = __builtin_bit_cast(u16x2_t, __builtin_amdgcn_perm(p01, p00, sel));
        const u16x2_t bp = __builtin_bit_cast(u16x2_t, __builtin_amdgcn_perm(p11, p10, sel));
        const uint32_t top = __builtin_amdgcn_udot2(tp, fxp, 0u, false), bot = __builtin_amdgcn_udot2(bp, fxp, 0u, false);
        acc[c] = mad24(top, fy1_16, mad24(bot, fy16, 1u << 23));
    }
    if constexpr (C == 1) return acc[0] >> 24;
    const uint32_t lo = __builtin_amdgcn_perm(acc[C > 1 ? 1 : 0], acc[0], 0x0c0c0703u);         // (acc0.b3, acc1.b3, 0, 0)
    if constexpr (C == 2) return lo;
    if constexpr (C == 3) return __builtin_amdgcn_perm(acc[C > 2 ? 2 : 0], lo, 0x0c070100u);    // (lo.b0, lo.b1, acc2.b3, 0)
    const uint32_t hi = __builtin_amdgcn_perm(acc[C > 3 ? 3 : 0], acc[C > 2 ? 2 : 0], 0x0c0c0703u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
template <int C>
__device__ __forceinline__ uint32_t blend_q10(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, uint32_t fx, uint32_t fy) {
    return blend_q10_w<C>(p00, p01, p10, p11, (1024u - fx) | (fx << 16), fy << 4);
}

// bilinear_sample_u8_valid (P/warp/common.rs:79-165): xi, yi in range; fx, fy in Q10
template <int C>
__device__ __forceinline__ uint32_t sample_q10(const uint8_t* __restrict__ src, int sw, int sh, int xi, int yi, uint32_t fx,
                                               uint32_t fy) {
    const int yi1 = yi + 1 < sh ? yi + 1 : yi;
    // xi1 = xi + 1 < sw ? xi + 1 : xi  ==  load_quad_u8's second pixel; offsets < 2^31 B (host-checked)
    const QuadU8 q = load_quad_u8<C>(src + (unsigned)(yi * sw) * C, src + (unsigned)(yi1 * sw) * C, xi, sw);
    return blend_q10<C>(q.p00, q.p01, q.p10, q.p11, fx, fy);
}

// bilinear_sample_u8 (P/warp/common.rs:16-70): zeros when non-finite or outside
template <int C>
__device__ __forceinline__ uint32_t sample_q10_checked(const uint8_t* __restrict__ src, int sw, int sh, float xf, float yf) {
    bool ok = __builtin_isfinite(xf) && __builtin_isfinite(yf);
    // floor(..) as i32 saturates in the reference; clamping first keeps the range test identical
    const int xi = (int)fminf(fmaxf(floorf(xf), -1.0f), 2147483520.0f);
    const int yi = (int)fminf(fmaxf(floorf(yf), -1.0f), 2147483520.0f);
    ok = ok && xi >= 0 && xi < sw && yi >= 0 && yi < sh;
    if (!ok) return 0u;
    return sample_q10<C>(src, sw, sh, xi, yi, (uint32_t)((xf - (float)xi) * 1024.0f), (uint32_t)((yf - (float)yi) * 1024.0f));
}

#define KH_U8_PROLOGUE                                          \
    unsigned bx_, by_, bz_;                                     \
    if (!xcd_tile(im.tiles, bx_, by_, bz_)) return;             \
    const int x = bx_ * kBx + threadIdx.x;                      \
    const int y = by_ * kBy + threadIdx.y;                      \
    if (x >= im.dw || y >= im.dh) return;                       \
    const bool wave_full = (int)(bx_ * kBx) + kBx <= im.dw; /* uniform: all 64 lanes of this row are inside */

// remap_u8 (P/interpolation/remap.rs:157-300); maps shared by the batch, kU8RemapNB images per thread
constexpr int kU8RemapNB = 4;
template <int C, int MODE>
__global__ __launch_bounds__(kBx* kBy) void remap_u8_kernel(ImgU8 im, const float* __restrict__ map_x,
                                                            const float* __restrict__ map_y, int batch) {
    KH_U8_PROLOGUE
    const long long i = (long long)y * im.dw + x;
    const float xf = map_x[i], yf = map_y[i];
    const int z0 = bz_ * kU8RemapNB;
#pragma unroll
    for (int k = 0; k < kU8RemapNB; ++k) {
        const int z = z0 + k;
        if (z >= batch) break;
        const uint8_t* src = im.src + (long long)z * im.src_stride;
        uint8_t* o = im.dst + (long long)z * im.dst_stride + i * C;
        uint32_t px = 0;
        if constexpr (MODE == KH_INTERP_BILINEAR) {
            px = sample_q10_checked<C>(src, im.sw, im.sh, xf, yf);
        } else {  // nearest, :268-298
            if (xf >= 0.0f && xf < (float)im.sw && yf >= 0.0f && yf < (float)im.sh) {
                const int xi = min(max((int)roundf(xf), 0), im.sw - 1), yi = min(max((int)roundf(yf), 0), im.sh - 1);
                px = load_px_u8<C>(src + ((long long)yi * im.sw + xi) * C);
            }
        }
        store_px_u8<C>(o, px, wave_full);
    }
}

// Rust `f32 as i64` / `as i32`: saturating, NaN -> 0.  Spans only compare against [0, dst_w], so
// clamping to +-4e18 before the conversion is equivalent.
__device__ __forceinline__ long long f2ll_sat(float v) {
    return v != v ? 0ll : (long long)fminf(fmaxf(v, -4.0e18f), 4.0e18f);
}
__host__ __device__ __forceinline__ int f2i_sat(float v) {
    return v != v ? 0 : (v >= 2147483648.0f ? 2147483647 : (v <= -2147483648.0f ? (-2147483647 - 1) : (int)v));
}

// constrain_span (P/warp/span.rs:36-59)
__device__ __forceinline__ void constrain_span(float a, float b, bool ge, float eps, long long& lo, long long& hi) {
    if (fabsf(a) < eps || a == 0.0f) {
        const bool feasible = ge ? (b >= 0.0f) : (b < 0.0f);
        if (!feasible) hi = lo;
        return;
    }
    const float k = -b / a;
    if (ge && a > 0.0f) lo = max(lo, f2ll_sat(ceilf(k)));
    else if (ge) hi = min(hi, f2ll_sat(floorf(k)) + 1);
    else if (a > 0.0f) hi = min(hi, f2ll_sat(ceilf(k)));
    else lo = max(lo, f2ll_sat(floorf(k)) + 1);
}

struct Mat6 { float m[6]; };
struct Mat9 { float m[9]; };

// warp_affine_u8 (P/warp/affine.rs:373-445): per-row valid span (P/warp/span.rs:61-85), Q16
// coordinates stepped from the span start with wrapping adds (P/warp/kernels.rs:386-415) — here
// sx_q_lo + (x - x_lo) * dsx_q in wrapping arithmetic, the same value.
//
// The span algebra (4 IEEE divisions, saturating i64 conversions) depends on the row only; doing it per
// pixel made the gather VALU-bound (r01m: 16.7 ms / 256 4K images, slower than the f32 warp).  A
// dst_h-thread pre-kernel writes one RowSpan per row into stream-ordered scratch shared by the batch.
struct AffineRow { int lo, hi; uint32_t sx_lo, sy_lo; };
__global__ __launch_bounds__(kBlock) void affine_rows_kernel(AffineRow* __restrict__ rows, int dw, int dh, int sw, int sh, Mat6 mi) {
    const int y = blockIdx.x * kBlock + threadIdx.x;
    if (y >= dh) return;
    const float dsx = mi.m[0], dsy = mi.m[3];
    const float sx0 = mi.m[1] * (float)y + mi.m[2], sy0 = mi.m[4] * (float)y + mi.m[5];
    long long lo = 0, hi = dw;
    constrain_span(dsx, sx0, true, 1e-12f, lo, hi);
    constrain_span(dsx, sx0 - (float)sw, false, 1e-12f, lo, hi);
    bool empty = lo >= hi;
    if (!empty) {
        constrain_span(dsy, sy0, true, 1e-12f, lo, hi);
        constrain_span(dsy, sy0 - (float)sh, false, 1e-12f, lo, hi);
        empty = lo >= hi;
    }
    lo = min(max(lo, 0ll), (long long)dw);
    hi = min(max(hi, 0ll), (long long)dw);
    AffineRow r{0, 0, 0u, 0u};
    if (!empty && lo < hi) {
        r.lo = (int)lo;
        r.hi = (int)hi;
        r.sx_lo = (uint32_t)f2i_sat((sx0 + dsx * (float)r.lo) * 65536.0f);
        r.sy_lo = (uint32_t)f2i_sat((sy0 + dsy * (float)r.lo) * 65536.0f);
    }
    rows[y] = r;
}

template <int C>
__global__ __launch_bounds__(kBx* kBy) void warp_affine_u8_kernel(ImgU8 im, const AffineRow* __restrict__ rows, int dsx_q, int dsy_q) {
    KH_U8_PROLOGUE
    const uint8_t* src = im.src + (long long)bz_ * im.src_stride;
    uint8_t* o = im.dst + (long long)bz_ * im.dst_stride + ((long long)y * im.dw + x) * C;
    const AffineRow r = rows[y];
    uint32_t px = 0;
    if (x >= r.lo && x < r.hi) {
        const int sx_q = (int)(r.sx_lo + (uint32_t)(x - r.lo) * (uint32_t)dsx_q);
        const int sy_q = (int)(r.sy_lo + (uint32_t)(x - r.lo) * (uint32_t)dsy_q);
        // The span keeps the indices in range in exact arithmetic; the clamp only matters where Q16
        // rounding drift would take the reference's unchecked sampler outside the image.
        const int xi = min(max(sx_q >> 16, 0), im.sw - 1), yi = min(max(sy_q >> 16, 0), im.sh - 1);
        px = sample_q10<C>(src, im.sw, im.sh, xi, yi, ((uint32_t)(sx_q & 0xFFFF)) >> 6, ((uint32_t)(sy_q & 0xFFFF)) >> 6);
    }
    store_px_u8<C>(o, px, wave_full);
}


// ---- helpers of the staged gather (below): four packed pixels <-> 4*C bytes ------------------------------------------------------
template <int C>
__device__ __forceinline__ void load_quad_px(const uint8_t* __restrict__ p, uint32_t px[4]) {
    if constexpr (C == 4) {
        const uint64_t a = *reinterpret_cast<const u64_unaligned*>(p), b = *reinterpret_cast<const u64_unaligned*>(p + 8);
        px[0] = (uint32_t)a; px[1] = (uint32_t)(a >> 32); px[2] = (uint32_t)b; px[3] = (uint32_t)(b >> 32);
    } else if constexpr (C == 3) {
        const uint64_t a = *reinterpret_cast<const u64_unaligned*>(p);
        const uint32_t b = *reinterpret_cast<const u32_unaligned*>(p + 8);
        px[0] = (uint32_t)a & 0xffffffu; px[1] = (uint32_t)(a >> 24) & 0xffffffu;
        px[2] = ((uint32_t)(a >> 48) | (b << 16)) & 0xffffffu; px[3] = b >> 8;
    } else if constexpr (C == 2) {
        const uint64_t a = *reinterpret_cast<const u64_unaligned*>(p);
        px[0] = (uint32_t)a & 0xffffu; px[1] = (uint32_t)(a >> 16) & 0xffffu; px[2] = (uint32_t)(a >> 32) & 0xffffu; px[3] = (uint32_t)(a >> 48);
    } else {
        const uint32_t a = *reinterpret_cast<const u32_unaligned*>(p);
        px[0] = a & 0xffu; px[1] = (a >> 8) & 0xffu; px[2] = (a >> 16) & 0xffu; px[3] = a >> 24;
    }
}
// four packed pixels -> 4*C bytes at `o` (any alignment)
template <int C>
__device__ __forceinline__ void store_quad_px(uint8_t* o, const uint32_t px[4]) {
    if constexpr (C == 4) {
        *reinterpret_cast<u64_unaligned*>(o) = (uint64_t)px[0] | ((uint64_t)px[1] << 32);
        *reinterpret_cast<u64_unaligned*>(o + 8) = (uint64_t)px[2] | ((uint64_t)px[3] << 32);
    } else if constexpr (C == 3) {
        *reinterpret_cast<u64_unaligned*>(o) = (uint64_t)px[0] | ((uint64_t)px[1] << 24) | ((uint64_t)px[2] << 48);
        *reinterpret_cast<u32_unaligned*>(o + 8) = (px[2] >> 16) | (px[3] << 8);
    } else if constexpr (C == 2) {
        *reinterpret_cast<u64_unaligned*>(o) = (uint64_t)px[0] | ((uint64_t)px[1] << 16) | ((uint64_t)px[2] << 32) | ((uint64_t)px[3] << 48);
    } else {
        *reinterpret_cast<u32_unaligned*>(o) = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
    }
}
// the same 4*C bytes as C dwords (a quad of C-channel pixels is exactly C dwords), for the streaming buffer store
template <int C>
__device__ __forceinline__ void pack_quad_px(const uint32_t px[4], uint32_t w[C]) {
    if constexpr (C == 4) { w[0] = px[0]; w[1] = px[1]; w[2] = px[2]; w[3] = px[3]; }
    else if constexpr (C == 3) { w[0] = px[0] | (px[1] << 24); w[1] = (px[1] >> 8) | (px[2] << 16); w[2] = (px[2] >> 16) | (px[3] << 8); }
    else if constexpr (C == 2) { w[0] = px[0] | (px[1] << 16); w[1] = px[2] | (px[3] << 16); }
    else w[0] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
}
template <int C>
__device__ __forceinline__ void store_one_px(uint8_t* o, uint32_t px) {
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = (uint8_t)(px >> (8 * c));
}

// warp_perspective_u8 (P/warp/perspective.rs:179-322): rows whose denominator keeps one sign get
// the analytic span, other rows the bounds-checked sampler on every column; coordinates are
// evaluated directly per column (perspective_coord_at, P/warp/kernels.rs:107-122).  Row terms come
// from the same kind of pre-kernel as the affine warp.
struct PerspRow { float nx0, ny0, nd0, dnx, dny, dnd; int lo, hi; };
__global__ __launch_bounds__(kBlock) void persp_rows_kernel(PerspRow* __restrict__ rows, int dw, int dh, int sw, int sh, Mat9 inv) {
    const int y = blockIdx.x * kBlock + threadIdx.x;
    if (y >= dh) return;
    const float yf = (float)y, swf = (float)sw, shf = (float)sh;
    PerspRow r;
    r.nx0 = inv.m[1] * yf + inv.m[2]; r.ny0 = inv.m[4] * yf + inv.m[5]; r.nd0 = inv.m[7] * yf + inv.m[8];
    r.dnx = inv.m[0]; r.dny = inv.m[3]; r.dnd = inv.m[6];
    r.lo = 0; r.hi = dw;
    const float nd_end = r.nd0 + r.dnd * ((float)dw - 1.0f);
    const bool pos = r.nd0 > 1e-6f && nd_end > 1e-6f, neg = r.nd0 < -1e-6f && nd_end < -1e-6f;
    if (pos || neg) {
        if (neg) { r.nx0 = -r.nx0; r.ny0 = -r.ny0; r.nd0 = -r.nd0; r.dnx = -r.dnx; r.dny = -r.dny; r.dnd = -r.dnd; }
        long long lo = 0, hi = dw;
        constrain_span(r.dnx, r.nx0, true, 0.0f, lo, hi);
        constrain_span(r.dnx - swf * r.dnd, r.nx0 - swf * r.nd0, false, 0.0f, lo, hi);
        constrain_span(r.dny, r.ny0, true, 0.0f, lo, hi);
        constrain_span(r.dny - shf * r.dnd, r.ny0 - shf * r.nd0, false, 0.0f, lo, hi);
        lo = min(max(lo, 0ll), (long long)dw);
        hi = min(max(hi, 0ll), (long long)dw);
        if (lo >= hi) { lo = 0; hi = 0; }
        r.lo = (int)lo;
        r.hi = (int)hi;
    }
    rows[y] = r;
}

template <int C>
__global__ __launch_bounds__(kBx* kBy) void warp_perspective_u8_kernel(ImgU8 im, const PerspRow* __restrict__ rows) {
    KH_U8_PROLOGUE
    const uint8_t* src = im.src + (long long)bz_ * im.src_stride;
    uint8_t* o = im.dst + (long long)bz_ * im.dst_stride + ((long long)y * im.dw + x) * C;
    const PerspRow r = rows[y];
    uint32_t px = 0;
    if (x >= r.lo && x < r.hi) {
        const float xf_ = (float)x;
        const float nx = r.nx0 + r.dnx * xf_, ny = r.ny0 + r.dny * xf_, nd = r.nd0 + r.dnd * xf_;
        const float inv_nd = 1.0f / nd;
        px = sample_q10_checked<C>(src, im.sw, im.sh, nx * inv_nd, ny * inv_nd);
    }
    store_px_u8<C>(o, px, wave_full);
}


// ---- staged Q10 gather: ONE kernel for warp_affine_u8 / warp_perspective_u8 / remap_u8 (bilinear) -------------------------------
// The per-pixel kernels above cost ~100 VALU instructions and four scattered sub-dword loads per pixel (0.14 of the HBM roofline on
// the 4K rotation, r01); round 2's staged affine kernel got to 0.38 and was VALU-bound at 67.6 instructions per pixel (r02q).
// What an output pixel needs splits into a part that depends on the GEOMETRY only — its coordinates, validity, blend weights, the
// source box of its tile — and a part that depends on the image: four taps and the blend.  A batch shares the geometry (one matrix,
// one pair of maps), so a 512-thread block owns a 64 x 32 destination tile of kStageNB CONSECUTIVE IMAGES and
//   A. evaluates the coordinates of its pixels ONCE (a thread = four consecutive pixels of a row), in the op's own arithmetic —
//      Q16 stepping along the row span (affine), per-column projective division inside the row span (perspective), the two map
//      reads (remap; they are the largest stream of remap_u8, 8 of 14 bytes per pixel, now read once per kStageNB images) — and
//      keeps the first-tap position, the packed horizontal weights (fx1, fx) and 16 * fy in registers;
//   B. reduces the tile's source bounding box in-block (packed 16-bit min / max: v_pk_min_u16 / v_pk_max_u16 through wave
//      shuffles, then eight partials through LDS) — exact by construction, it is the min / max of the very tap indices that will
//      be used, so no pre-pass, no margins, no analytic bound per op;
//   C. derives from the box, once: every pixel's LDS tap address and the (source offset, LDS slot) of the up-to-four 4-pixel
//      groups the thread stages;
//   D. per image: stages the box (one dword per pixel, 4*C-byte contiguous loads, cells past the last image column / row replicate
//      the edge = the reference's `xi + 1 < sw ? xi + 1 : xi` tap rule), barrier, then per pixel two ds_read2_b32 and blend_q10 —
//      about 25 VALU instructions per pixel and image — and one 4*C-byte store per thread.
// A tile whose box does not fit 32 KiB of LDS (strong minification, wild maps) samples from global memory with the same registers
// (a block-uniform branch); images wider or taller than 65535 (16-bit box fields) keep the per-pixel kernels.  Same spans, clamps,
// floors and integer blend as those kernels: byte-identical (tests run both).
constexpr int kStageW = 64, kStageH = 32;   // destination tile (same-box A/B on the 4K rotation, r02p: 8 rows 5.05 ms, 16 rows 4.60, 32 rows 4.46)
constexpr int kStageCap = 8192;             // staged pixels per block: 32 KiB of LDS; four 512-thread blocks per CU (the wave limit)
constexpr int kSpanRows = 128;             // boxes of at most this many rows keep a per-row column span (affine)
constexpr int kStageNB = 8;                 // images per block (4 -> 8: the geometry phase was still ~15 of 43 VALU instructions per pixel and image, r03f)
enum { kOpAffine = 0, kOpPersp = 1, kOpRemap = 2 };
struct GatherOp {
    const void* rows;                         // AffineRow* (affine) / PerspRow* (perspective)
    const float* map_x; const float* map_y;   // remap
    int dsx_q, dsy_q;                         // affine: Q16 column steps
    int batch;
    int nb = kStageNB;                        // images per block (set by launch_staged_gather)
    int tile_rows = kStageH;                  // 32 or 16 destination rows per block (stage_rows below); 8 = a 128 x 8 tile (perspective)
    int spans = 1;                            // affine: stage only the quads inside the per-row spans of the box (test option warp_u8_spans = 0: the whole box)
};

// bilinear_sample_u8's admission rule (P/warp/common.rs:16-70), as in sample_q10_checked: non-finite or outside -> not sampled
__device__ __forceinline__ bool checked_tap(float xf, float yf, int sw, int sh, int& xi, int& yi, uint32_t& fx, uint32_t& fy) {
    bool ok = __builtin_isfinite(xf) && __builtin_isfinite(yf);
    xi = (int)fminf(fmaxf(floorf(ok ? xf : 0.0f), -1.0f), 2147483520.0f);
    yi = (int)fminf(fmaxf(floorf(ok ? yf : 0.0f), -1.0f), 2147483520.0f);
    ok = ok && xi >= 0 && xi < sw && yi >= 0 && yi < sh;
    fx = (uint32_t)(ok ? (xf - (float)xi) * 1024.0f : 0.0f);
    fy = (uint32_t)(ok ? (yf - (float)yi) * 1024.0f : 0.0f);
    return ok;
}

// Raw staging loads: a quad = 4 * C contiguous bytes = kRawDw<C> dwords, kept as loaded (unpacked into one dword per pixel only
// when they are written to LDS), so the quads of image b + 1 can sit in few registers while image b is sampled.
template <int C> struct RawQuad { uint32_t d[C]; };
template <int C>
__device__ __forceinline__ RawQuad<C> load_raw_quad(const uint8_t* __restrict__ p) {
    RawQuad<C> r;
    if constexpr (C == 4) {
        const uint64_t a = *reinterpret_cast<const u64_unaligned*>(p), b = *reinterpret_cast<const u64_unaligned*>(p + 8);
        r.d[0] = (uint32_t)a; r.d[1] = (uint32_t)(a >> 32); r.d[2] = (uint32_t)b; r.d[3] = (uint32_t)(b >> 32);
    } else if constexpr (C == 3) {
        const uint64_t a = *reinterpret_cast<const u64_unaligned*>(p);
        r.d[0] = (uint32_t)a; r.d[1] = (uint32_t)(a >> 32); r.d[2] = *reinterpret_cast<const u32_unaligned*>(p + 8);
    } else if constexpr (C == 2) {
        const uint64_t a = *reinterpret_cast<const u64_unaligned*>(p);
        r.d[0] = (uint32_t)a; r.d[1] = (uint32_t)(a >> 32);
    } else {
        r.d[0] = *reinterpret_cast<const u32_unaligned*>(p);
    }
    return r;
}
// four pixels, one dword each (channel c = bits [8c, 8c + 8); bits above 8C are don't-care: the blend's byte selectors never read them)
template <int C>
__device__ __forceinline__ u32x4_t unpack_raw_quad(const RawQuad<C>& r) {
    if constexpr (C == 4) return u32x4_t{r.d[0], r.d[1], r.d[2], r.d[3]};
    else if constexpr (C == 3)
        return u32x4_t{r.d[0], __builtin_amdgcn_alignbyte(r.d[1], r.d[0], 3), __builtin_amdgcn_alignbyte(r.d[2], r.d[1], 2), r.d[2] >> 8};
    else if constexpr (C == 2) return u32x4_t{r.d[0], r.d[0] >> 16, r.d[1], r.d[1] >> 16};
    else return u32x4_t{r.d[0], r.d[0] >> 8, r.d[0] >> 16, r.d[0] >> 24};
}

// The same for a box that reaches past the last image column (tiles at the right border only; a rolled loop, one quad at a time):
// such a quad was loaded from the row's last four columns and is re-indexed so that cells past the edge replicate it.
template <int C, int NT>
__device__ __forceinline__ void stage_rounds_edge(uint32_t* __restrict__ tile, const uint8_t* __restrict__ src, const uint32_t (&soff)[4], unsigned wmask, int tid, int nq,
                                               int kmax, int xmin, int pitch, int lpitch, int sw) {
    const int qpr = pitch >> 2;
#pragma unroll 1
    for (int k = 0; k < kmax; ++k) {
        const int q = tid + k * NT;
        if (q >= nq) break;
        if (!((wmask >> k) & 1u)) continue;   // a quad outside its row's span: not staged
        uint32_t a[4];
        load_quad_px<C>(src + soff[k], a);
        const int c0 = xmin + 4 * (q % qpr), d = c0 - min(c0, sw - 4);   // 0 for the quads inside the row
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = min(d + j, 3);
            o[j] = t == 0 ? a[0] : (t == 1 ? a[1] : (t == 2 ? a[2] : a[3]));
        }
        *reinterpret_cast<u32x4_t*>(&tile[(q / qpr) * lpitch + 4 * (q % qpr)]) = u32x4_t{o[0], o[1], o[2], o[3]};
    }
}

// The staged images of one block, KM staging rounds per image (block-uniform, a template so that the raw quads are registers).
// Software-pipelined: the loads of image b + 1 are issued right after the barrier that publishes image b's box and complete while
// image b is sampled — a block hides its own load latency instead of relying on its neighbours (three blocks share a CU).
// (Tried, r03ze: boxes of at most half the tile alternating between its two halves — ONE barrier per image instead of two: no change
// in time on any of the three operators; the barriers are not what bounds the kernel.)
template <int C, int KM>
__device__ __forceinline__ void staged_images(uint32_t* __restrict__ tile, const ImgU8& im, int z0, int nimg, const uint32_t (&soff)[4], const int (&sdst)[4], unsigned wmask, int tid,
                                              const int (&la)[4], int pitch, const uint32_t (&fxp)[4], const uint32_t (&fy16)[4], unsigned valid,
                                              long long dst_off, bool mine, bool whole, int x4) {
    // block-uniform: every quad offset of the image is a multiple of four bytes and fits the V#'s 2 GiB window
    const bool stream_ok = ((im.dw * C) & 3) == 0 && (long long)im.dw * im.dh * C <= 0x7fffffffLL;
    const uint8_t* src = im.src + (long long)z0 * im.src_stride;
    RawQuad<C> raw[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) raw[k] = load_raw_quad<C>(src + soff[k]);
    // The store of image b is ISSUED one phase late, after image b + 1 has been staged: loads and stores share vmcnt and may complete out
    // of order with each other, so the wait for image b + 1's quads is a vmcnt(0) — placed right behind the store it also waited for the
    // store's acknowledgement, every image (ablation r04z3: no stores -29 %, no loads -21 %, the two additive).  Issued here the store is
    // older than the loads the next wait is for, and long acknowledged by then.
    uint32_t pend[4] = {0u, 0u, 0u, 0u};
    auto emit = [&](int b_, const uint32_t (&px)[4]) {
        uint8_t* o = im.dst + (long long)(z0 + b_) * im.dst_stride + dst_off;
        if (mine) {
            if (whole && stream_ok) {   // write-through non-temporal buffer store: the image is written once and never read back
                uint32_t w[C];
                pack_quad_px<C>(px, w);
                stream_store<C>(stream_window(im.dst + (long long)(z0 + b_) * im.dst_stride, (long long)im.dw * im.dh * C), (int)dst_off, w);
            } else if (whole) {
                store_quad_px<C>(o, px);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)  // fixed trip count: a run-time bound indexes px[] dynamically and puts it in scratch
                    if (x4 + j < im.dw) store_one_px<C>(o + j * C, px[j]);
            }
        }
    };
#pragma unroll 1
    for (int b = 0; b < nimg; ++b) {
        // One unconditional wait for this image's quads.  Left to the compiler the wait sits inside the `wmask` branches, so a path
        // exists on which the quads were never waited for, and it adds a vmcnt(0) before the next loads overwrite their registers —
        // behind the store below, i.e. the very wait for the store's acknowledgement this order is meant to avoid.
        wait_vmcnt0();
        #pragma unroll
        for (int k = 0; k < KM; ++k) {
            if ((wmask >> k) & 1u) *reinterpret_cast<u32x4_t*>(&tile[sdst[k]]) = unpack_raw_quad<C>(raw[k]);   // r * lpitch + 4 * c4
        }
        if (b > 0) emit(b - 1, pend);
        __syncthreads();
        if (b + 1 < nimg) {   // block-uniform
            src += im.src_stride;
#pragma unroll
            for (int k = 0; k < KM; ++k) raw[k] = load_raw_quad<C>(src + soff[k]);
        }
        uint32_t t[4][4];   // all sixteen taps first: eight LDS reads in flight
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t* t0 = tile + la[j];
            t[j][0] = t0[0]; t[j][1] = t0[1]; t[j][2] = t0[pitch]; t[j][3] = t0[pitch + 1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t px = blend_q10_w<C>(t[j][0], t[j][1], t[j][2], t[j][3], fxp[j], fy16[j]);
            pend[j] = ((valid >> j) & 1u) ? px : 0u;
        }
        __syncthreads();   // every thread has read this image's taps before the next box is written
    }
    emit(nimg - 1, pend);
}

template <int C, int OP, int TH, int TW = kStageW>
__global__ __launch_bounds__(TW / 4 * TH) void gather_u8_staged_kernel(ImgU8 im, GatherOp op) {
    constexpr int QX = TW / 4, NT = QX * TH, CAP = kStageCap * (TW * TH) / (kStageW * kStageH);
    __shared__ __attribute__((aligned(16))) uint32_t tile[CAP];
    __shared__ uint32_t red[16];
    __shared__ uint32_t span_lo[OP == kOpAffine ? kSpanRows : 1], span_hi[OP == kOpAffine ? kSpanRows : 1];
    unsigned bx_, by_, bz_;
    if (!xcd_tile(im.tiles, bx_, by_, bz_)) return;   // block-uniform
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * QX + tx;   // block (QX, TH): QX four-pixel groups x TH rows
    const int x4 = bx_ * TW + 4 * tx, y = by_ * TH + ty;
    const bool row_in = y < im.dh;
    const int yc = min(y, im.dh - 1);

    // A. coordinates of this thread's four pixels (geometry only)
    uint32_t xy[4], fxp[4], fy16[4];
    unsigned valid = 0;
    if constexpr (OP == kOpAffine) {
        const AffineRow r = static_cast<const AffineRow*>(op.rows)[yc];
        uint32_t sx = r.sx_lo + (uint32_t)(x4 - r.lo) * (uint32_t)op.dsx_q, sy = r.sy_lo + (uint32_t)(x4 - r.lo) * (uint32_t)op.dsy_q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x4 + j;
            if (row_in && x >= r.lo && x < r.hi) valid |= 1u << j;   // hi <= dw
            // The span keeps the indices in range in exact arithmetic; the clamp only matters where Q16 rounding drift would take the
            // reference's unchecked sampler outside the image.
            const int xi = min(max((int)sx >> 16, 0), im.sw - 1), yi = min(max((int)sy >> 16, 0), im.sh - 1);
            const uint32_t fx = (sx & 0xFFFFu) >> 6, fy = (sy & 0xFFFFu) >> 6;
            xy[j] = (uint32_t)xi | ((uint32_t)yi << 16);
            fxp[j] = (1024u - fx) | (fx << 16);
            fy16[j] = fy << 4;
            sx += (uint32_t)op.dsx_q; sy += (uint32_t)op.dsy_q;
        }
    } else {
        PerspRow r{};
        if constexpr (OP == kOpPersp) r = static_cast<const PerspRow*>(op.rows)[yc];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x4 + j;
            float xf, yf;
            bool in;
            if constexpr (OP == kOpPersp) {
                in = row_in && x >= r.lo && x < r.hi;
                const float xf_ = (float)x;
                const float nx = r.nx0 + r.dnx * xf_, ny = r.ny0 + r.dny * xf_, nd = r.nd0 + r.dnd * xf_;
                const float inv_nd = 1.0f / nd;
                xf = nx * inv_nd; yf = ny * inv_nd;
            } else {
                in = row_in && x < im.dw;
                const int i = yc * im.dw + min(x, im.dw - 1);   // unconditional, clamped loads (dw * dh < 2^31: host-checked)
                xf = op.map_x[i]; yf = op.map_y[i];
            }
            int xi, yi;
            uint32_t fx, fy;
            const bool ok = checked_tap(xf, yf, im.sw, im.sh, xi, yi, fx, fy) && in;
            if (ok) valid |= 1u << j;
            xy[j] = ok ? ((uint32_t)xi | ((uint32_t)yi << 16)) : 0u;
            fxp[j] = (1024u - fx) | (fx << 16);
            fy16[j] = fy << 4;
        }
    }

    // B. the tile's source box: packed (x, y) minima and maxima of the valid first taps
    u16x2_t lo2 = __builtin_bit_cast(u16x2_t, 0xFFFFFFFFu), hi2 = __builtin_bit_cast(u16x2_t, 0u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool v = (valid >> j) & 1u;
        lo2 = __builtin_elementwise_min(lo2, __builtin_bit_cast(u16x2_t, v ? xy[j] : 0xFFFFFFFFu));
        hi2 = __builtin_elementwise_max(hi2, __builtin_bit_cast(u16x2_t, v ? xy[j] : 0u));
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        lo2 = __builtin_elementwise_min(lo2, __builtin_bit_cast(u16x2_t, (uint32_t)__shfl_xor((int)__builtin_bit_cast(uint32_t, lo2), m)));
        hi2 = __builtin_elementwise_max(hi2, __builtin_bit_cast(u16x2_t, (uint32_t)__shfl_xor((int)__builtin_bit_cast(uint32_t, hi2), m)));
    }
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = __builtin_bit_cast(uint32_t, lo2); red[2 * (tid >> 6) + 1] = __builtin_bit_cast(uint32_t, hi2); }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        lo2 = __builtin_elementwise_min(lo2, __builtin_bit_cast(u16x2_t, red[2 * w]));
        hi2 = __builtin_elementwise_max(hi2, __builtin_bit_cast(u16x2_t, red[2 * w + 1]));
    }
    const int xmin = lo2[0], ymin = lo2[1], xmax = hi2[0], ymax = hi2[1];
    const bool any = xmax >= xmin;                              // block-uniform: some pixel of the tile is sampled
    const bool mine = row_in && x4 < im.dw, whole = x4 + 3 < im.dw;
    const int z0 = bz_ * op.nb, nimg = min(op.nb, op.batch - z0);
    const long long dst_off = ((long long)y * im.dw + x4) * C;

    // staged box: columns [xmin, xmin + pitch), rows [ymin, ymax + 1]: one column / row more than the first taps reach
    const int bh = ymax + 2 - ymin, pitch = (xmax + 2 - xmin + 3) & ~3;
    const bool staged = any && pitch * bh <= CAP;         // block-uniform
    // LDS row pitch (dwords).  A pitch of 0 mod 32 takes a quarter off the tap reads' bank conflicts on the rotation (309 M -> 233 M
    // cycles, as scripts/diag/lds_bank_sim.py predicts) for 1 % of the time (r04x): the LDS is a third busy; the box pitch stays.
    const int lpitch = pitch;

    // C. per-thread plan, from the box: LDS tap index per pixel; source byte offset of the (up to four) quads this thread stages.
    // C0 (affine).  The box of a rotated tile is up to twice its footprint (64 x 32 at 12 degrees: 1.97 staged source pixels per
    // destination pixel, the footprint 1.24): each box row keeps the column span its taps use — per thread the columns of its four
    // pixels, on the rows they touch (min / max through LDS atomics, once per block) — and only the quads inside a row's span
    // are loaded and written (1.39 per pixel).  Exact by construction like the box: a cell outside every span is never read.
    bool spans = false;   // block-uniform
    if constexpr (OP == kOpAffine) {
        spans = staged && op.spans != 0 && bh <= kSpanRows;
        if (spans) {
            if (tid < bh) { span_lo[tid] = 0xFFFFu; span_hi[tid] = 0u; }
            __syncthreads();
            if (valid) {
                uint32_t cmin = 0xFFFFu, cmax = 0u, rmin = 0xFFFFu, rmax = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((valid >> j) & 1u) {
                        const uint32_t c = (xy[j] & 0xFFFFu) - (uint32_t)xmin, r = (xy[j] >> 16) - (uint32_t)ymin;
                        cmin = min(cmin, c); cmax = max(cmax, c); rmin = min(rmin, r); rmax = max(rmax, r);
                    }
                for (uint32_t r = rmin; r <= rmax + 1u; ++r) { atomicMin(&span_lo[r], cmin); atomicMax(&span_hi[r], cmax + 1u); }
            }
            __syncthreads();
        }
    }
    int la[4], sdst[4];
    uint32_t soff[4];
    unsigned wmask = 0;      // bit k: this thread stages the quad of round k
    int kmax = 0, nq = 0;    // block-uniform: staging rounds (512 quads each), quads in the box
    bool at_edge = false;    // block-uniform: the box reaches past the last image column (replicated cells)
    if (staged) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            la[j] = ((valid >> j) & 1u) ? (int)__umul24((xy[j] >> 16) - (uint32_t)ymin, (uint32_t)lpitch) + (int)((xy[j] & 0xFFFFu) - (uint32_t)xmin) : 0;
        const int qpr = pitch >> 2;
        nq = qpr * bh;
        kmax = (nq + NT - 1) / NT;
        at_edge = xmin + pitch > im.sw;
        const float inv_qpr = 1.0f / (float)qpr;   // q / qpr for q < 2048: the float quotient is within one of the integer one
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q0 = tid + k * NT, q = q0 < nq ? q0 : 0;   // idle lanes re-load the box's first quad (and do not write it)
            int r_ = (int)((float)q * inv_qpr);
            r_ -= (r_ * qpr > q);
            r_ += ((r_ + 1) * qpr <= q);
            int c4 = q - r_ * qpr;
            sdst[k] = r_ * lpitch + 4 * c4;
            bool need = q0 < nq;
            if constexpr (OP == kOpAffine) {
                if (spans) {
                    const int lo = (int)span_lo[r_], hi = (int)span_hi[r_];
                    if (lo <= hi) {   // (every row of the box is touched; an untouched one would be staged whole)
                        need = need && 4 * c4 + 3 >= lo && 4 * c4 <= hi;
                        c4 = min(max(c4, lo >> 2), hi >> 2);   // a skipped quad re-loads a quad of its row that is staged anyway
                    }
                }
            }
            wmask |= (need ? 1u : 0u) << k;
            const int c0 = xmin + 4 * c4;
            // an edge quad is loaded from the last four columns of its row and re-indexed afterwards (sw >= 4: host-checked)
            const int cl = at_edge ? min(c0, im.sw - 4) : c0;
            soff[k] = __umul24((uint32_t)min(ymin + r_, im.sh - 1), (uint32_t)(im.sw * C)) + (uint32_t)(cl * C);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) la[j] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { soff[k] = 0u; sdst[k] = 0; }
    }

    // D. the images of this block: stage the box, barrier, sample, store; the second barrier keeps the next image's staging off a box
    // that is still being read.  Latency is hidden by the other blocks of the CU.  Two separate loops (staged / not) so that the
    // fallback's operands are dead in the staged loop (registers decide how many blocks share a CU).
    auto emit = [&](uint8_t* o, const uint32_t (&out)[4]) {
        if (!mine) return;
        if (whole) {
            store_quad_px<C>(o, out);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)  // fixed trip count: a run-time bound indexes out[] dynamically and puts it in scratch
                if (x4 + j < im.dw) store_one_px<C>(o + j * C, out[j]);
        }
    };
    if (staged && !at_edge) {
        switch (kmax) {   // block-uniform
            case 1: staged_images<C, 1>(tile, im, z0, nimg, soff, sdst, wmask, tid, la, lpitch, fxp, fy16, valid, dst_off, mine, whole, x4); break;
            case 2: staged_images<C, 2>(tile, im, z0, nimg, soff, sdst, wmask, tid, la, lpitch, fxp, fy16, valid, dst_off, mine, whole, x4); break;
            case 3: staged_images<C, 3>(tile, im, z0, nimg, soff, sdst, wmask, tid, la, lpitch, fxp, fy16, valid, dst_off, mine, whole, x4); break;
            default: staged_images<C, 4>(tile, im, z0, nimg, soff, sdst, wmask, tid, la, lpitch, fxp, fy16, valid, dst_off, mine, whole, x4); break;
        }
    } else if (staged) {   // tiles whose box reaches past the last image column: rolled staging loop, no pipelining
#pragma unroll 1
        for (int b = 0; b < nimg; ++b) {
            const uint8_t* src = im.src + (long long)(z0 + b) * im.src_stride;
            if (b > 0) __syncthreads();
            stage_rounds_edge<C, NT>(tile, src, soff, wmask, tid, nq, kmax, xmin, pitch, lpitch, im.sw);
            __syncthreads();
            uint32_t out[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t* t0 = tile + la[j];
                const uint32_t px = blend_q10_w<C>(t0[0], t0[1], t0[lpitch], t0[lpitch + 1], fxp[j], fy16[j]);
                out[j] = ((valid >> j) & 1u) ? px : 0u;
            }
            emit(im.dst + (long long)(z0 + b) * im.dst_stride + dst_off, out);
        }
    } else {
#pragma unroll 1
        for (int b = 0; b < nimg; ++b) {
            const uint8_t* src = im.src + (long long)(z0 + b) * im.src_stride;
            uint32_t out[4] = {0u, 0u, 0u, 0u};
            if (any) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((valid >> j) & 1u)
                        out[j] = sample_q10<C>(src, im.sw, im.sh, (int)(xy[j] & 0xFFFFu), (int)(xy[j] >> 16), fxp[j] >> 16, fy16[j] >> 4);
            }
            emit(im.dst + (long long)(z0 + b) * im.dst_stride + dst_off, out);
        }
    }
}

ImgU8 make_img_u8(const uint8_t* src, uint8_t* dst, int sw, int sh, int dw, int dh, int64_t ss, int64_t ds, int groups) {
    return ImgU8{src, dst, sw, sh, dw, dh, ss, ds, xcd_tiles(cdiv(dw, kBx), cdiv(dh, kBy), (unsigned)groups, cdiv(dw, kBx) * 8)};
}

#define KH_DISPATCH_C(KERNEL, channels, grid, stream, ...)                                              \
    do {                                                                                                \
        const dim3 blk(kBx, kBy);                                                                       \
        if ((channels) == 1) hipLaunchKernelGGL((KERNEL<1>), grid, blk, 0, stream, __VA_ARGS__);        \
        else if ((channels) == 2) hipLaunchKernelGGL((KERNEL<2>), grid, blk, 0, stream, __VA_ARGS__);   \
        else if ((channels) == 3) hipLaunchKernelGGL((KERNEL<3>), grid, blk, 0, stream, __VA_ARGS__);   \
        else hipLaunchKernelGGL((KERNEL<4>), grid, blk, 0, stream, __VA_ARGS__);                        \
    } while (0)


// The staged gather serves images up to 65535 x 65535 (16-bit box fields); kh_debug_set_option("warp_u8_direct", 1) (test option) keeps
// the per-pixel kernels for all three operators.
bool use_staged_gather(int sw, int sh) {
    const bool direct = dev_opt(kOptWarpU8Direct) == 1;
    return !direct && sw <= 65535 && sh <= 65535 && sw >= 4;   // 16-bit box fields; a staged quad is four pixels of one row
}
// Rows of the destination tile, from the Jacobian [a b; c d] of the destination -> source map (the matrix of an affine warp, a
// homography's at the image centre): 64 x 16 tiles in 256-thread blocks — six per CU instead of three — when the box of such a tile
// is staged in two rounds (near axis-aligned maps: perspective -5 %, remap -2 %, r04z2); a rotated tile's box needs the 512 threads of
// a 64 x 32 tile (the 12-degree rotation: +12 % with 16 rows).
int stage_rows(float a, float b, float c, float d) {
    const float bw = 64.0f * fabsf(a) + 16.0f * fabsf(b) + 2.0f, bh = 64.0f * fabsf(c) + 16.0f * fabsf(d) + 2.0f;
    if (!(bw < 1e6f && bh < 1e6f)) return kStageH;
    return (ceilf(bw * 0.25f) + 1.0f) * ceilf(bh) <= 512.0f ? 16 : kStageH;
}
template <int OP>
int32_t launch_staged_gather(hipStream_t st, const uint8_t* src, uint8_t* dst, int sw, int sh, int dw, int dh, int channels, int batch,
                             int64_t ss, int64_t ds, const GatherOp& op_, const char* what) {
    // images per block: 16 for batches of 128 and more — the geometry phase (perspective divisions, the remap's map reads) is paid once
    // per block: perspective 3.50-3.60 -> 3.27 ms, remap 3.53-3.64 -> 3.11-3.13 ms, affine unchanged; 32 and 64 are slower (r03ze) —
    // 8 otherwise.
    GatherOp op = op_;
    op.nb = batch >= 128 ? 2 * kStageNB : kStageNB;
    op.spans = dev_opt(kOptWarpU8Spans) != 0;
    const int forced = dev_opt(kOptWarpU8Rows), th = forced == 16 || forced == 32 || forced == 8 ? forced : op.tile_rows;   // test option warp_u8_rows (8 = 128 x 8 tiles)
    const int tw = th == 8 ? 128 : kStageW;
    const unsigned tiles_x = cdiv(dw, tw), tiles_y = cdiv(dh, th), groups = cdiv(batch, op.nb);
    // 64 x 32 tiles, dealt to the XCDs in runs of 256 destination rows like the other gathers.  (128 x 16 tiles — whole 384-byte store rows,
    // WRITE_SIZE 6.50 -> 6.22 GB — r04z1: perspective -2 %, remap +6 %, the 12-degree rotation +34 %: its box needs four staging rounds.
    // Walking each band column-major, so that the blocks in flight form a 2-D patch — r04z5: reads 8.36 -> 7.27 GB on the rotation at
    // the same 3.19 ms, perspective and remap +3 %: the kernel is not bound by its traffic.)
    const ImgU8 im{src, dst, sw, sh, dw, dh, ss, ds, xcd_tiles(tiles_x, tiles_y, groups, tiles_x * (256 / th))};
    KH_REQUIRE(im.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
    const dim3 grid = xcd_grid(im.tiles), blk(tw / 4, th);
#define KH_STAGED(CH)                                                                                                \
    do {                                                                                                             \
        if (th == 8) hipLaunchKernelGGL((gather_u8_staged_kernel<CH, OP, 8, 128>), grid, blk, 0, st, im, op);        \
        else if (th == 16) hipLaunchKernelGGL((gather_u8_staged_kernel<CH, OP, 16>), grid, blk, 0, st, im, op);      \
        else hipLaunchKernelGGL((gather_u8_staged_kernel<CH, OP, kStageH>), grid, blk, 0, st, im, op);               \
    } while (0)
    switch (channels) {
        case 1: KH_STAGED(1); break;
        case 2: KH_STAGED(2); break;
        case 3: KH_STAGED(3); break;
        default: KH_STAGED(4); break;
    }
#undef KH_STAGED
    return check_launch(what);
}

}  // namespace

extern "C" {

// quantize_kernel_256 (P/filter/ops.rs:748-760)
void kh_quantize_kernel_256(const float* k, int32_t n, uint8_t* out) {
    int sum = 0;
    for (int i = 0; i < n; ++i) {
        const float v = k[i] * 256.0f + 0.5f;
        const int q = v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (int)v);  // `as u8` saturates (NaN -> 0)
        out[i] = (uint8_t)q;
        sum += q;
    }
    if (n > 0 && sum != 256) {
        const int c = (int)out[n / 2] + (256 - sum);
        out[n / 2] = (uint8_t)(c < 0 ? 0 : (c > 255 ? 255 : c));
    }
}

int32_t kh_gaussian_blur_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t cols, int32_t rows,
                            int32_t channels, int32_t ksize_x, int32_t ksize_y, float sigma_x, float sigma_y,
                            int32_t batch, int64_t src_stride, int64_t dst_stride) {
    if (int32_t rc = check_u8_img("kh_gaussian_blur_u8", src, dst, cols, rows, cols, rows, channels, batch, src_stride, dst_stride))
        return rc;
    int32_t k[2] = {ksize_x, ksize_y};
    float s[2] = {sigma_x, sigma_y};
    KH_REQUIRE(kh_gaussian_resolve(k, s) == KH_OK, KH_ERR_INVALID_ARG,
               "kh_gaussian_blur_u8: invalid kernel size (%d, %d) / sigma (%g, %g)", ksize_x, ksize_y, sigma_x, sigma_y);
    KH_REQUIRE(k[0] <= 63 && k[1] <= 63, KH_ERR_UNSUPPORTED, "kh_gaussian_blur_u8: kernel (%d, %d) wider than 63 taps", k[0], k[1]);
    KH_REQUIRE(src != dst || batch == 0, KH_ERR_INVALID_ARG, "kh_gaussian_blur_u8: in-place filtering is not supported");
    if (batch == 0) return KH_OK;
    // blur_u8_path (P/filter/ops.rs:21-27): the 3x3 / sigma in [0.6, 1.2] case is the [1,2,1]/4 binomial
    const bool binomial = k[0] == 3 && k[1] == 3 && s[0] >= 0.6f && s[0] <= 1.2f && s[1] >= 0.6f && s[1] <= 1.2f;
    float fx[64], fy[64];
    uint8_t qx[64], qy[64];
    kh_gaussian_kernel_1d(k[0], s[0], fx);
    kh_gaussian_kernel_1d(k[1], s[1], fy);
    kh_quantize_kernel_256(fx, k[0], qx);
    kh_quantize_kernel_256(fy, k[1], qy);
    return launch_blur_u8(stream, src, dst, cols, rows, channels, qx, k[0], qy, k[1], binomial, batch, src_stride,
                          dst_stride, "kh_gaussian_blur_u8");
}

int32_t kh_box_blur_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t cols, int32_t rows,
                       int32_t channels, int32_t ksize_x, int32_t ksize_y, int32_t batch, int64_t src_stride,
                       int64_t dst_stride) {
    if (int32_t rc = check_u8_img("kh_box_blur_u8", src, dst, cols, rows, cols, rows, channels, batch, src_stride, dst_stride))
        return rc;
    // P/filter/ops.rs:66-75: odd, positive sizes only
    KH_REQUIRE(ksize_x > 0 && ksize_y > 0 && (ksize_x & 1) && (ksize_y & 1), KH_ERR_INVALID_ARG,
               "kh_box_blur_u8: kernel size (%d, %d) must be odd and positive", ksize_x, ksize_y);
    KH_REQUIRE(ksize_x <= 63 && ksize_y <= 63, KH_ERR_UNSUPPORTED, "kh_box_blur_u8: kernel (%d, %d) wider than 63 taps", ksize_x, ksize_y);
    KH_REQUIRE(src != dst || batch == 0, KH_ERR_INVALID_ARG, "kh_box_blur_u8: in-place filtering is not supported");
    if (batch == 0) return KH_OK;
    float fx[64], fy[64];
    uint8_t qx[64], qy[64];
    kh_box_blur_kernel_1d(ksize_x, fx);
    kh_box_blur_kernel_1d(ksize_y, fy);
    kh_quantize_kernel_256(fx, ksize_x, qx);
    kh_quantize_kernel_256(fy, ksize_y, qy);
    return launch_blur_u8(stream, src, dst, cols, rows, channels, qx, ksize_x, qy, ksize_y, false, batch, src_stride,
                          dst_stride, "kh_box_blur_u8");
}

int32_t kh_remap_u8(kh_stream_t stream, const uint8_t* src, const float* map_x, const float* map_y, uint8_t* dst,
                    int32_t sw, int32_t sh, int32_t dw, int32_t dh, int32_t channels, int32_t mode, int32_t batch,
                    int64_t src_stride, int64_t dst_stride) {
    if (int32_t rc = check_u8_img("kh_remap_u8", src, dst, sw, sh, dw, dh, channels, batch, src_stride, dst_stride, true)) return rc;
    // P/interpolation/remap.rs:171-178: only nearest and bilinear exist for u8
    KH_REQUIRE(mode == KH_INTERP_NEAREST || mode == KH_INTERP_BILINEAR, KH_ERR_UNSUPPORTED,
               "kh_remap_u8: interpolation mode %d is not supported for u8 (nearest, bilinear)", mode);
    if (batch == 0) return KH_OK;
    KH_REQUIRE(map_x && map_y, KH_ERR_INVALID_ARG, "kh_remap_u8: null map pointer");
    if (mode == KH_INTERP_BILINEAR && use_staged_gather(sw, sh)) {
        GatherOp op{nullptr, map_x, map_y, 0, 0, batch};
        op.tile_rows = 16;   // a correction map is close to the identity
        // one channel (round 6, profiles/r06zu_warp_gray_tiles.txt): a 64-pixel tile row is HALF a cache line; 128 x 8 tiles 0.194 -> 0.146 ms per
        // 16 4K planes (RGB / RGBA: +-3 %, they keep 64 x 16)
        if (channels == 1) op.tile_rows = 8;
        return launch_staged_gather<kOpRemap>(as_hip(stream), src, dst, sw, sh, dw, dh, channels, batch, src_stride, dst_stride, op, "kh_remap_u8");
    }
    const ImgU8 im = make_img_u8(src, dst, sw, sh, dw, dh, src_stride, dst_stride, (batch + kU8RemapNB - 1) / kU8RemapNB);
    KH_REQUIRE(im.tiles.total > 0, KH_ERR_TOO_LARGE, "kh_remap_u8: batch x tiles exceeds one launch");
    const dim3 blk(kBx, kBy), grid = xcd_grid(im.tiles);
    hipStream_t st = as_hip(stream);
    switch (channels * 10 + mode) {
        case 10: hipLaunchKernelGGL((remap_u8_kernel<1, 0>), grid, blk, 0, st, im, map_x, map_y, batch); break;
        case 11: hipLaunchKernelGGL((remap_u8_kernel<1, 1>), grid, blk, 0, st, im, map_x, map_y, batch); break;
        case 20: hipLaunchKernelGGL((remap_u8_kernel<2, 0>), grid, blk, 0, st, im, map_x, map_y, batch); break;
        case 21: hipLaunchKernelGGL((remap_u8_kernel<2, 1>), grid, blk, 0, st, im, map_x, map_y, batch); break;
        case 30: hipLaunchKernelGGL((remap_u8_kernel<3, 0>), grid, blk, 0, st, im, map_x, map_y, batch); break;
        case 31: hipLaunchKernelGGL((remap_u8_kernel<3, 1>), grid, blk, 0, st, im, map_x, map_y, batch); break;
        case 40: hipLaunchKernelGGL((remap_u8_kernel<4, 0>), grid, blk, 0, st, im, map_x, map_y, batch); break;
        default: hipLaunchKernelGGL((remap_u8_kernel<4, 1>), grid, blk, 0, st, im, map_x, map_y, batch); break;
    }
    return check_launch("kh_remap_u8");
}

int32_t kh_warp_affine_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t sw, int32_t sh, int32_t dw,
                          int32_t dh, int32_t channels, const float* m, int32_t batch, int64_t src_stride,
                          int64_t dst_stride) {
    if (int32_t rc = check_u8_img("kh_warp_affine_u8", src, dst, sw, sh, dw, dh, channels, batch, src_stride, dst_stride, true))
        return rc;
    KH_REQUIRE(m, KH_ERR_INVALID_ARG, "kh_warp_affine_u8: null matrix");
    if (batch == 0) return KH_OK;
    Mat6 mi;
    kh_invert_affine_transform(m, mi.m);
    const int dsx_q = f2i_sat(mi.m[0] * 65536.0f), dsy_q = f2i_sat(mi.m[3] * 65536.0f);
    const ImgU8 im = make_img_u8(src, dst, sw, sh, dw, dh, src_stride, dst_stride, batch);
    KH_REQUIRE(im.tiles.total > 0, KH_ERR_TOO_LARGE, "kh_warp_affine_u8: batch x tiles exceeds one launch");
    Scratch scratch;  // per-row spans shared by the batch: caller workspace or stream-ordered pool
    if (int32_t rc = get_scratch(stream, sizeof(AffineRow) * (size_t)dh, "kh_warp_affine_u8", scratch)) return rc;
    AffineRow* rows = scratch.as<AffineRow>();
    hipLaunchKernelGGL(affine_rows_kernel, dim3(cdiv(dh, kBlock)), dim3(kBlock), 0, as_hip(stream), rows, dw, dh, sw, sh, mi);
    if (use_staged_gather(sw, sh)) {
        GatherOp op{rows, nullptr, nullptr, dsx_q, dsy_q, batch};
        op.tile_rows = stage_rows(mi.m[0], mi.m[1], mi.m[3], mi.m[4]);
        if (channels == 1 && op.tile_rows == 16) {   // one channel: whole 128-byte tile rows where the box of a 128 x 8 tile still fits two staging rounds
            const float bw = 128.0f * fabsf(mi.m[0]) + 8.0f * fabsf(mi.m[1]) + 2.0f, bh = 128.0f * fabsf(mi.m[3]) + 8.0f * fabsf(mi.m[4]) + 2.0f;
            if (bw < 1e6f && bh < 1e6f && (ceilf(bw * 0.25f) + 1.0f) * ceilf(bh) <= 512.0f) op.tile_rows = 8;
        }
        return launch_staged_gather<kOpAffine>(as_hip(stream), src, dst, sw, sh, dw, dh, channels, batch, src_stride, dst_stride, op, "kh_warp_affine_u8");
    }
    KH_DISPATCH_C(warp_affine_u8_kernel, channels, xcd_grid(im.tiles), as_hip(stream), im, (const AffineRow*)rows, dsx_q, dsy_q);
    return check_launch("kh_warp_affine_u8");
}

int32_t kh_warp_perspective_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t sw, int32_t sh,
                               int32_t dw, int32_t dh, int32_t channels, const float* m, int32_t batch,
                               int64_t src_stride, int64_t dst_stride) {
    if (int32_t rc = check_u8_img("kh_warp_perspective_u8", src, dst, sw, sh, dw, dh, channels, batch, src_stride, dst_stride, true))
        return rc;
    KH_REQUIRE(m, KH_ERR_INVALID_ARG, "kh_warp_perspective_u8: null matrix");
    Mat9 inv;
    if (int32_t rc = kh_invert_homography(m, inv.m)) return rc;
    if (batch == 0) return KH_OK;
    const ImgU8 im = make_img_u8(src, dst, sw, sh, dw, dh, src_stride, dst_stride, batch);
    KH_REQUIRE(im.tiles.total > 0, KH_ERR_TOO_LARGE, "kh_warp_perspective_u8: batch x tiles exceeds one launch");
    Scratch scratch;
    if (int32_t rc = get_scratch(stream, sizeof(PerspRow) * (size_t)dh, "kh_warp_perspective_u8", scratch)) return rc;
    PerspRow* rows = scratch.as<PerspRow>();
    hipLaunchKernelGGL(persp_rows_kernel, dim3(cdiv(dh, kBlock)), dim3(kBlock), 0, as_hip(stream), rows, dw, dh, sw, sh, inv);
    if (use_staged_gather(sw, sh)) {
        GatherOp op{rows, nullptr, nullptr, 0, 0, batch};
        {   // Jacobian of (x, y) -> (nx / nd, ny / nd) at the centre of the destination
            const float x = 0.5f * (float)dw, y = 0.5f * (float)dh;
            const float nd = inv.m[6] * x + inv.m[7] * y + inv.m[8], xs = (inv.m[0] * x + inv.m[1] * y + inv.m[2]) / nd, ys = (inv.m[3] * x + inv.m[4] * y + inv.m[5]) / nd;
            const float ja = (inv.m[0] - xs * inv.m[6]) / nd, jb = (inv.m[1] - xs * inv.m[7]) / nd, jc = (inv.m[3] - ys * inv.m[6]) / nd, jd = (inv.m[4] - ys * inv.m[7]) / nd;
            op.tile_rows = stage_rows(ja, jb, jc, jd);
            // Round 6 (profiles/r06q_u8_gather_128x8.txt): 128 x 8 tiles — whole 384-byte store rows, the box of a
            // near-axis-aligned map still staged in two rounds of the 256 threads — perspective 2.99 -> 2.87 ms (0.533 -> 0.557); remap
            // +3 % and the 12-degree rotation falls off the box capacity (10.3 ms), so only this operator, only where the box fits.
            const float bw = 128.0f * fabsf(ja) + 8.0f * fabsf(jb) + 2.0f, bh = 128.0f * fabsf(jc) + 8.0f * fabsf(jd) + 2.0f;
            if (op.tile_rows == 16 && bw < 1e6f && bh < 1e6f && (ceilf(bw * 0.25f) + 1.0f) * ceilf(bh) <= 512.0f) op.tile_rows = 8;
        }
        return launch_staged_gather<kOpPersp>(as_hip(stream), src, dst, sw, sh, dw, dh, channels, batch, src_stride, dst_stride, op, "kh_warp_perspective_u8");
    }
    KH_DISPATCH_C(warp_perspective_u8_kernel, channels, xcd_grid(im.tiles), as_hip(stream), im, (const PerspRow*)rows);
    return check_launch("kh_warp_perspective_u8");
}

}  // extern "C"
