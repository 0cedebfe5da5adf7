// The rest of the reference's filter module for gfx950: 3x3 spatial gradients, the repeated-box "fast" blur, median
// blur and the cv2-compatible bilateral filter (crates/kornia-imgproc/src/filter/{ops,median,bilateral}.rs and their
// device twins cuda/{median,bilateral}.rs).  All four are per-pixel maps over a small replicated / reflected window:
//
//   spatial_gradient   4 B in, 8 B out per element: HBM-bound.  Four consecutive elements of the HWC row per thread: nine
//                      aligned dwordx4 window loads, two dwordx4 stores (1 KiB contiguous per wave store); a one-element
//                      form covers unaligned images, C > 4 and row lengths not divisible by 4.
//   fast_hfilter       the running row sum is a serial f32 chain per (row, channel) — that order IS the result — so
//                      the parallelism is rows x channels x batch; lanes are consecutive (row, channel) pairs, which
//                      makes the TRANSPOSED store of the reference's layout the coalesced one; the row reads are staged
//                      through LDS in column chunks so that they are coalesced too.
//   median             exact order statistic, compute-bound (~2 * |network| min/max per pixel): two horizontally adjacent
//                      pixels ride in the 16-bit halves of one register through a proved selection network
//                      (kh_median_net.h), so every v_pk_min_u16 / v_pk_max_u16 serves two pixels; window bytes arrive as
//                      a few unaligned dwords per row.
//   bilateral          per pixel ntaps x (byte load, LDS colour-weight lookup, mul, add, fma) with the host-built
//                      cv2 tables; the 1 KiB colour table lives in LDS, taps stream through the scalar/vector cache;
//                      interior pixels skip the border reflection.
#include <math.h>
#include <string.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include <type_traits>

#include "kh_common.h"
#include "kh_median_net.h"
#include "kh_table_cache.h"

using namespace kh;

namespace {

constexpr int kBx = 64, kBy = 4;

int32_t check_batch(const char* what, int cols, int rows, int C, int batch, int cmax) {
    KH_REQUIRE(cols > 0 && rows > 0, KH_ERR_INVALID_ARG, "%s: zero-sized image %dx%d", what, cols, rows);
    KH_REQUIRE(C >= 1 && C <= cmax, KH_ERR_UNSUPPORTED, "%s: no device kernel for %d channels (supported: 1..%d)", what, C, cmax);
    KH_REQUIRE(batch >= 0 && batch <= 65535, KH_ERR_TOO_LARGE, "%s: batch %d outside [0, 65535]", what, batch);
    KH_REQUIRE((int64_t)cols * rows * C <= kI32Max, KH_ERR_TOO_LARGE, "%s: image exceeds 32-bit indexing", what);
    KH_REQUIRE(cdiv(rows, kBy) <= 65535u, KH_ERR_TOO_LARGE, "%s: %d rows exceed one launch", what, rows);
    return KH_OK;
}

// ---- spatial_gradient_float / scharr_spatial_gradient_float (P/filter/ops.rs:287-590) -----------------------------------
// Nine products added in (dy, dx) row-major order onto 0.0, zero taps included, replicate border
// (row = min(r + dy, rows).max(1) - 1).  Kernels: P/filter/kernels.rs:107-140, a = corner, b = centre weight.
__global__ __launch_bounds__(kBx* kBy) void spatial_gradient_kernel(const float* __restrict__ src, float* __restrict__ gx,
                                                                    float* __restrict__ gy, int rows, int cols, int C, float a,
                                                                    float b, long long ss, long long ds) {
    const int rowlen = cols * C;
    const int i = blockIdx.x * kBx + threadIdx.x, r = blockIdx.y * kBy + threadIdx.y;
    if (i >= rowlen || r >= rows) return;
    const float* s = src + (long long)blockIdx.z * ss;
    const int left = i >= C ? -C : 0, right = i < rowlen - C ? C : 0;
    const float* up = s + (long long)max(r - 1, 0) * rowlen + i;
    const float* mid = s + (long long)r * rowlen + i;
    const float* dn = s + (long long)min(r + 1, rows - 1) * rowlen + i;
    const float v00 = up[left], v01 = up[0], v02 = up[right];
    const float v10 = mid[left], v11 = mid[0], v12 = mid[right];
    const float v20 = dn[left], v21 = dn[0], v22 = dn[right];
    float sx = 0.0f, sy = 0.0f;
    sx += v00 * -a;   sy += v00 * -a;
    sx += v01 * 0.0f; sy += v01 * -b;
    sx += v02 * a;    sy += v02 * -a;
    sx += v10 * -b;   sy += v10 * 0.0f;
    sx += v11 * 0.0f; sy += v11 * 0.0f;
    sx += v12 * b;    sy += v12 * 0.0f;
    sx += v20 * -a;   sy += v20 * a;
    sx += v21 * 0.0f; sy += v21 * b;
    sx += v22 * a;    sy += v22 * a;
    const long long o = (long long)blockIdx.z * ds + (long long)r * rowlen + i;
    gx[o] = sx;
    gy[o] = sy;
}

// Four consecutive elements per thread (C <= 4, row length a multiple of 4, 16-byte-aligned images): the 3 x 12-float window
// of a thread is nine aligned dwordx4 loads and the two results go out as dwordx4 stores — 1 KiB contiguous per wave store
// instead of 256 B (on this chip the store segment size is what separates 5.2 from 6+ TB/s on streaming maps, profiles/r01b),
// and 2.25 load instructions per element instead of 9.  Same nine products in the same order per element.  Threads whose window
// leaves the row (first / last 4 elements of a row) take clamped scalar loads; the values they clamp are never used (the
// replicate rule substitutes the centre column there).
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int C>  // compile-time channel count: every window index below is a constant, so the window stays in registers
__global__ __launch_bounds__(kBx* kBy) void spatial_gradient_x4_kernel(const float* __restrict__ src, float* __restrict__ gx,
                                                                       float* __restrict__ gy, int rows, int cols, float a, float b,
                                                                       long long ss, long long ds) {
    const int rowlen = cols * C;
    const int i = 4 * (blockIdx.x * kBx + threadIdx.x), r = blockIdx.y * kBy + threadIdx.y;
    if (i >= rowlen || r >= rows) return;
    const float* s = src + (long long)blockIdx.z * ss;
    const bool inside = i >= 4 && i + 8 <= rowlen;
    float w[3][12];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int rr = k == 0 ? max(r - 1, 0) : (k == 1 ? r : min(r + 1, rows - 1));
        const float* row = s + (long long)rr * rowlen;
        if (inside) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4v v = *reinterpret_cast<const f32x4v*>(row + i - 4 + 4 * q);
                w[k][4 * q] = v.x; w[k][4 * q + 1] = v.y; w[k][4 * q + 2] = v.z; w[k][4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 12; ++q) w[k][q] = row[min(max(i - 4 + q, 0), rowlen - 1)];
        }
    }
    float ox[4], oy[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool has_left = i + e >= C, has_right = i + e < rowlen - C;  // else replicate: the centre column
        float l[3], m[3], rt[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            m[k] = w[k][4 + e];
            l[k] = has_left ? w[k][4 + e - C] : m[k];
            rt[k] = has_right ? w[k][4 + e + C] : m[k];
        }
        float sx = 0.0f, sy = 0.0f;
        sx += l[0] * -a;    sy += l[0] * -a;
        sx += m[0] * 0.0f;  sy += m[0] * -b;
        sx += rt[0] * a;    sy += rt[0] * -a;
        sx += l[1] * -b;    sy += l[1] * 0.0f;
        sx += m[1] * 0.0f;  sy += m[1] * 0.0f;
        sx += rt[1] * b;    sy += rt[1] * 0.0f;
        sx += l[2] * -a;    sy += l[2] * a;
        sx += m[2] * 0.0f;  sy += m[2] * b;
        sx += rt[2] * a;    sy += rt[2] * a;
        ox[e] = sx;
        oy[e] = sy;
    }
    // both gradients leave through streaming stores (kh_common.h::stream_store); an image beyond the V#'s 2 GiB window keeps plain ones
    const long long img = (long long)rows * rowlen * 4;
    if (img <= 0x7fffffffLL) {   // launch-uniform
        const uint32_t bx_[4] = {__float_as_uint(ox[0]), __float_as_uint(ox[1]), __float_as_uint(ox[2]), __float_as_uint(ox[3])};
        const uint32_t by_[4] = {__float_as_uint(oy[0]), __float_as_uint(oy[1]), __float_as_uint(oy[2]), __float_as_uint(oy[3])};
        const int off = (r * rowlen + i) * 4;
        stream_store<4>(stream_window(gx + (long long)blockIdx.z * ds, img), off, bx_);
        stream_store<4>(stream_window(gy + (long long)blockIdx.z * ds, img), off, by_);
        return;
    }
    const long long o = (long long)blockIdx.z * ds + (long long)r * rowlen + i;
    *reinterpret_cast<f32x4v*>(gx + o) = f32x4v{ox[0], ox[1], ox[2], ox[3]};
    *reinterpret_cast<f32x4v*>(gy + o) = f32x4v{oy[0], oy[1], oy[2], oy[3]};
}

// ---- fast_horizontal_filter (P/filter/separable_filter.rs:202-257) ----------------------------------------------------
// acc = x0 * (half + 1) + sum of the next `half` pixels at column 0, then acc -= leaving, acc += entering (ends replicated);
// out = acc / (2 * half + 1), stored transposed.  One thread per (row, channel); the chain is sequential by definition.
__global__ __launch_bounds__(kBlock) void fast_hfilter_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                              int cols, int C, int half, long long ss, long long ds) {
    const int t = blockIdx.x * kBlock + threadIdx.x;  // = r * C + ch
    if (t >= rows * C) return;
    const int r = t / C, ch = t - r * C;
    const float* row = src + (long long)blockIdx.y * ss + (long long)r * cols * C + ch;
    float* out = dst + (long long)blockIdx.y * ds + t;  // + c * rows * C per column
    const long long ostep = (long long)rows * C;
    const float leftmost = row[0], rightmost = row[(long long)(cols - 1) * C];
    const float norm = (float)(half * 2 + 1);
    float acc = leftmost * (float)(half + 1);
    for (int p = 0; p < half; ++p) acc += row[(long long)(p + 1) * C];
    out[0] = acc / norm;
    for (int c = 1; c < cols; ++c) {
        const float leaving = c >= half + 1 ? row[(long long)(c - half - 1) * C] : leftmost;
        const float entering = c + half < cols ? row[(long long)(c + half) * C] : rightmost;
        acc -= leaving;
        acc += entering;
        out[(long long)c * ostep] = acc / norm;
    }
}

// The same chain with the row reads staged through LDS.  In the direct kernel a wave's 64 lanes read 64 different rows —
// 64 cache lines per load instruction — so it is bound by vector-memory requests, not bytes.  Here a 256-thread block owns
// 256 consecutive (row, channel) chains (~256 / C rows) and walks the image in chunks of T columns: the block loads, for each of
// its rows, the (T + 2 * half + 1) * C contiguous floats the chunk needs (one coalesced 256-byte request per wave instruction)
// into LDS, then every thread runs T steps of its chain out of LDS (two ds_read_b32 per step; row pitch odd in dwords so the
// rows of a wave spread over the banks) and stores transposed (256 contiguous bytes per wave store).  Same operations in the
// same order per chain: bit-identical to the direct kernel.
extern __shared__ __attribute__((aligned(16))) float hfilter_lds[];

__global__ __launch_bounds__(kBlock) void fast_hfilter_lds_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols,
                                                                  int C, int half, int T, int pitch, long long ss, long long ds) {
    const int t0 = blockIdx.x * kBlock, t = t0 + threadIdx.x, total = rows * C;
    const bool live = t < total;
    const int r_first = t0 / C, r_last = min(t0 + kBlock - 1, total - 1) / C, nrows = r_last - r_first + 1;
    const int r = live ? t / C : r_first, ch = live ? t - r * C : 0, rl = r - r_first;
    const float* img = src + (long long)blockIdx.y * ss;
    const float* row = img + (long long)r * cols * C + ch;
    float* out = dst + (long long)blockIdx.y * ds + t;
    const long long ostep = (long long)rows * C;
    const float leftmost = live ? row[0] : 0.0f, rightmost = live ? row[(long long)(cols - 1) * C] : 0.0f;
    const float norm = (float)(half * 2 + 1);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const float* mine = hfilter_lds + rl * pitch + ch;
    float acc = 0.0f;
    for (int c0 = 0; c0 < cols; c0 += T) {
        const int lo = max(c0 - half - 1, 0), hi = min(c0 + T - 1 + half, cols - 1), seg = (hi - lo + 1) * C;
        __syncthreads();  // the previous chunk has been consumed
        // 32 independent loads (eight rows x four 64-float segments) are in flight before the first LDS write: with one load per
        // loop trip every trip cost a full memory round trip, and the ~66 trips per chunk were the kernel's whole time (r02y: 2.6 ms
        // per pass of 64 1080p RGB f32 images; eight in flight: 1.3 ms).  Two blocks of four waves per CU leave 256 VGPRs per lane.
        constexpr int kWaves = kBlock / kWave, kRowsB = 8, kSegB = 4;
        for (int k0 = wave; k0 < nrows; k0 += kRowsB * kWaves) {
            const float* g[kRowsB];
#pragma unroll
            for (int i = 0; i < kRowsB; ++i) g[i] = img + ((long long)(r_first + min(k0 + i * kWaves, nrows - 1)) * cols + lo) * C;
            for (int e0 = lane; e0 < seg; e0 += kSegB * kWave) {
                float v[kRowsB][kSegB];
#pragma unroll
                for (int i = 0; i < kRowsB; ++i)
#pragma unroll
                    for (int j = 0; j < kSegB; ++j) v[i][j] = g[i][min(e0 + j * kWave, seg - 1)];
#pragma unroll
                for (int i = 0; i < kRowsB; ++i)
#pragma unroll
                    for (int j = 0; j < kSegB; ++j)
                        if (k0 + i * kWaves < nrows && e0 + j * kWave < seg) hfilter_lds[(k0 + i * kWaves) * pitch + e0 + j * kWave] = v[i][j];
            }
        }
        __syncthreads();
        if (live) {
            const int cend = min(c0 + T, cols);
            int c = c0;
            if (c0 == 0) {
                acc = leftmost * (float)(half + 1);
                for (int p = 0; p < half; ++p) acc += mine[(p + 1 - lo) * C];
                out[0] = acc / norm;
                c = 1;
            }
            for (; c < cend; ++c) {
                const float leaving = c >= half + 1 ? mine[(c - half - 1 - lo) * C] : leftmost;
                const float entering = c + half < cols ? mine[(c + half - lo) * C] : rightmost;
                acc -= leaving;
                acc += entering;
                out[(long long)c * ostep] = acc / norm;
            }
        }
    }
}

// ---- median_blur (P/filter/median.rs:174-250, cuda/median.rs) ---------------------------------------------------------
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// Window loads: the two pixels' windows span K + 1 columns = (K + 1) * C contiguous bytes per row.  Away from the left / right
// border a thread fetches them as ceil((K + 1) * C / 4) unaligned dwords and picks the bytes out with constant-index bit-field
// extracts — 2..6 vector-memory instructions per row instead of (K + 1) * C byte loads (these kernels are otherwise bound by
// load instructions, like the u8 gathers: DESIGN.md section 5).  The dword run may cover up to 3 bytes more than the window; it
// is taken only when those bytes are still inside the row, so nothing is read outside the image.  WIDE = false (dev knob
// the `false` instantiation, not built) keeps the byte loads everywhere for A/B.
template <int K, int C, bool WIDE>
__global__ __launch_bounds__(kBx* kBy) void median_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int cols,
                                                          long long ss, long long ds) {
    constexpr int R = K / 2, NB = (K + 1) * C, NW = (NB + 3) / 4;
    const int x0 = 2 * (blockIdx.x * kBx + threadIdx.x), y = blockIdx.y * kBy + threadIdx.y;  // this thread: pixels x0 and x0 + 1
    if (x0 >= cols || y >= rows) return;
    const uint8_t* s = src + (long long)blockIdx.z * ss;
    uint8_t* o = dst + (long long)blockIdx.z * ds + ((long long)y * cols + x0) * C;
    int sx[K + 1];
#pragma unroll
    for (int j = 0; j <= K; ++j) sx[j] = min(max(x0 + j - R, 0), cols - 1) * C;  // replicate border (cv2.medianBlur's)
    const bool second = x0 + 1 < cols;
    const bool wide = WIDE && x0 >= R && (x0 - R) * C + 4 * NW <= cols * C;  // no clamped column, dword run inside the row
    uint32_t words[K][NW];
    if (wide) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const uint8_t* row = s + (long long)min(max(y + i - R, 0), rows - 1) * cols * C + (x0 - R) * C;
#pragma unroll
            for (int w = 0; w < NW; ++w) words[i][w] = *reinterpret_cast<const u32_unaligned*>(row + 4 * w);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        us2 v[K * K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            unsigned short col[K + 1];
            if (wide) {
#pragma unroll
                for (int j = 0; j <= K; ++j) col[j] = (unsigned short)((words[i][(j * C + c) >> 2] >> (8 * ((j * C + c) & 3))) & 0xffu);
            } else {
                const uint8_t* row = s + (long long)min(max(y + i - R, 0), rows - 1) * cols * C + c;
#pragma unroll
                for (int j = 0; j <= K; ++j) col[j] = row[sx[j]];
            }
#pragma unroll
            for (int j = 0; j < K; ++j) v[i * K + j] = us2{col[j], col[j + 1]};  // .x: window of x0, .y: window of x0 + 1
        }
#define KH_CE(p, q)                                                                                     \
    {                                                                                                   \
        const us2 lo = __builtin_elementwise_min(v[p], v[q]), hi = __builtin_elementwise_max(v[p], v[q]); \
        v[p] = lo;                                                                                      \
        v[q] = hi;                                                                                      \
    }
        if constexpr (K == 3) { KH_MEDIAN_9(KH_CE) } else { KH_MEDIAN_25(KH_CE) }
#undef KH_CE
        const us2 m = v[K == 3 ? KH_MEDIAN_9_OUT : KH_MEDIAN_25_OUT];
        o[c] = (uint8_t)m.x;
        if (second) o[C + c] = (uint8_t)m.y;
    }
}

// ---- bilateral_filter (P/filter/bilateral.rs, cuda/bilateral.rs:33-62) ------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len) {  // modulo form of P/clahe.rs:36-48 (same indices, cuda/pyramid.rs:36-45)
    if (len == 1) return 0;
    if (p < 0) p = -p;
    const int period = 2 * (len - 1);
    p %= period;
    return p >= len ? period - p : p;
}

struct Tap { int dy, dx; };
struct BilateralTab {           // one device allocation: [colour 256 f32][space n f32][(dy, dx) n x 2 i32][order n i32]
    const float* color;
    const float* space;
    const Tap* taps;
    const int* order;
    int n, radius;
};

__global__ __launch_bounds__(kBx* kBy) void bilateral_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int cols,
                                                             BilateralTab t, int simd_end, long long ss, long long ds) {
    __shared__ float color_w[256];
    color_w[threadIdx.y * kBx + threadIdx.x] = t.color[threadIdx.y * kBx + threadIdx.x];  // kBx * kBy == 256
    __syncthreads();
    const int x = blockIdx.x * kBx + threadIdx.x, y = blockIdx.y * kBy + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const uint8_t* s = src + (long long)blockIdx.z * ss;
    const int val0 = s[(long long)y * cols + x];
    const bool in_simd = x < simd_end;  // cv2's 16-pixel SIMD loop uses a permuted tap order when ntaps == 13
    float wsum = 0.0f, sum = 0.0f;
    // Pixels at least `radius` away from every border need no reflection: two integer modulos per tap (~80 VALU instructions)
    // are skipped for all but a frame of the image.  Same taps, same order, same arithmetic.
    const bool inner = x >= t.radius && x + t.radius < cols && y >= t.radius && y + t.radius < rows;
    const uint8_t* centre = s + (long long)y * cols + x;
    auto accumulate = [&](int k) {
        const Tap tap = t.taps[k];
        int val;
        if (inner) val = centre[tap.dy * cols + tap.dx];
        else val = s[(long long)reflect101(y + tap.dy, rows) * cols + reflect101(x + tap.dx, cols)];
        const float wgt = t.space[k] * color_w[abs(val - val0)];
        wsum += wgt;
        sum = fmaf((float)val, wgt, sum);
    };
    // simd_end is a multiple of 16 and a block spans 64 columns, so in all but one block column every lane uses the same tap
    // order: then the tap index is block-uniform and the order / (dy, dx) / space-weight reads are scalar loads (one per wave)
    // instead of three vector loads per tap per lane (r02y: 8.4 ms per 256 1080p images, bound by load issue).
    const int bx0 = blockIdx.x * kBx, by0 = blockIdx.y * kBy;
    // Blocks that lie wholly inside the reflection frame and use one tap order take the taps eight at a time: the eight table
    // reads (scalar), then the eight pixel bytes, then the eight colour weights are each issued together, and only the
    // accumulation runs in tap order.  One tap per trip was a serial chain of two scalar-load, one vector-load and one LDS
    // latency per tap — 13 taps x ~1000 cycles per thread, which is what the kernel's 6.3 ms were (r02za).
    const bool blk_inner = bx0 >= t.radius && bx0 + kBx - 1 + t.radius < cols && by0 >= t.radius && by0 + kBy - 1 + t.radius < rows;
    auto batched = [&](auto ordered) {
        constexpr int kB = 8;
        for (int k0 = 0; k0 < t.n; k0 += kB) {
            int off[kB], val[kB];
            float sw[kB], cw[kB];
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int kk = min(k0 + j, t.n - 1);
                int k = kk;
                if constexpr (decltype(ordered)::value) k = t.order[kk];
                const Tap tap = t.taps[k];
                off[j] = tap.dy * cols + tap.dx;
                sw[j] = t.space[k];
            }
#pragma unroll
            for (int j = 0; j < kB; ++j) val[j] = centre[off[j]];
#pragma unroll
            for (int j = 0; j < kB; ++j) cw[j] = color_w[abs(val[j] - val0)];
#pragma unroll
            for (int j = 0; j < kB; ++j)
                if (k0 + j < t.n) {
                    const float wgt = sw[j] * cw[j];
                    wsum += wgt;
                    sum = fmaf((float)val[j], wgt, sum);
                }
        }
    };
    if (blk_inner && bx0 + kBx <= simd_end) {
        batched(std::true_type{});
    } else if (blk_inner && bx0 >= simd_end) {
        batched(std::false_type{});
    } else if (bx0 + kBx <= simd_end) {
        for (int kk = 0; kk < t.n; ++kk) accumulate(t.order[kk]);
    } else if (bx0 >= simd_end) {
        for (int kk = 0; kk < t.n; ++kk) accumulate(kk);
    } else {
        for (int kk = 0; kk < t.n; ++kk) accumulate(in_simd ? t.order[kk] : kk);
    }
    dst[(long long)blockIdx.z * ds + (long long)y * cols + x] = (uint8_t)(int)rintf(sum / wsum);
}

// OpenCV's v_exp_default_32f polynomial as the reference transcribes it (bilateral.rs:44-78); every step is an fma there.
float v_exp_f32(float x) {
    const float lo = -88.37626f, hi = 89.0f, log2e = 1.44269504088896340736f, c1 = -6.9335938E-1f, c2 = 2.1219444E-4f;
    const float p0 = 1.9875692E-4f, p1 = 1.3981999E-3f, p2 = 8.333452E-3f, p3 = 4.1665796E-2f, p4 = 1.6666665E-1f, p5 = 5.0000002E-1f;
    x = x < lo ? lo : (x > hi ? hi : x);
    const float mm = floorf(fmaf(x, log2e, 0.5f));
    const uint32_t scale_bits = (uint32_t)((int32_t)mm + 0x7f) << 23;
    float scale;
    memcpy(&scale, &scale_bits, sizeof scale);
    x = fmaf(mm, c1, x);
    x = fmaf(mm, c2, x);
    const float xx = x * x;
    float y = fmaf(x, p0, p1);
    y = fmaf(y, x, p2);
    y = fmaf(y, x, p3);
    y = fmaf(y, x, p4);
    y = fmaf(y, x, p5);
    y = fmaf(y, xx, x);
    return (y + 1.0f) * scale;
}

struct HostTables { int radius = 1; std::vector<int32_t> dy, dx, order; std::vector<float> space, color; };

// radius = d / 2, or round-half-even(1.5 sigma_space) through Rust's saturating cast when d <= 0; at least 1 (bilateral.rs:114-121)
int bilateral_radius(int d, double sigma_space) {
    int radius;
    if (d <= 0) {
        const double r = nearbyint(sigma_space * 1.5);
        radius = r != r ? 0 : (r >= 2147483647.0 ? 2147483647 : (r <= -2147483648.0 ? INT32_MIN : (int)r));
    } else {
        radius = d / 2;
    }
    return radius < 1 ? 1 : radius;
}
constexpr int kMaxBilateralRadius = 512;  // ~824 000 taps per pixel; beyond this the tap tables alone run to gigabytes

// build_tables (bilateral.rs:110-170)
void build_tables(int d, double sigma_color, double sigma_space, HostTables& t) {
    const float color_coeff = (float)(-0.5 / (sigma_color * sigma_color)), space_coeff = (float)(-0.5 / (sigma_space * sigma_space));
    t.radius = bilateral_radius(d, sigma_space);
    t.color.assign(256, 0.0f);
    int i = 0;
    for (; i < 256 - 4; ++i) { const float fi = (float)i; t.color[i] = v_exp_f32(fi * fi * color_coeff); }  // cv2's SIMD polynomial ...
    for (; i < 256; ++i) t.color[i] = expf((float)(i * i) * color_coeff);                                     // ... and its scalar-expf tail (4 NEON lanes)
    for (int y = -t.radius; y <= t.radius; ++y)
        for (int x = -t.radius; x <= t.radius; ++x) {
            const double r = sqrt((double)(y * y + x * x));
            if (r > (double)t.radius) continue;
            t.space.push_back((float)exp((r * r) * (double)space_coeff));
            t.dy.push_back(y);
            t.dx.push_back(x);
        }
    static const int32_t order13[13] = {0, 12, 1, 2, 3, 9, 10, 11, 4, 5, 6, 7, 8};  // cv2's unrolled maxk == 13 block: lines 1,5,2,4,3
    const int n = (int)t.dy.size();
    for (int k = 0; k < n; ++k) t.order.push_back(n == 13 ? order13[k] : k);
}

// Device tables are cached per (device, d, sigma bits) like the resize contribution tables: uploaded with a blocking copy
// into a fresh allocation BEFORE they are published (the reference re-uploads five small arrays on every call); LRU beyond
// 64 parameter sets, never freeing a table a launch may still read (kh_table_cache.h).
// never destroyed: at process exit the HIP runtime may already be gone when static destructors run
TableCache<std::tuple<int, int, uint64_t, uint64_t>>& g_bil = *new TableCache<std::tuple<int, int, uint64_t, uint64_t>>(64);

int32_t get_bilateral_tab(int d, double sigma_color, double sigma_space, hipStream_t stream, const char* what, BilateralTab& out, TableLease& lease) {
    int dev = 0;
    KH_HIP(hipGetDevice(&dev));
    uint64_t cb, sb;
    memcpy(&cb, &sigma_color, 8);
    memcpy(&sb, &sigma_space, 8);
    const auto key = std::make_tuple(dev, d, cb, sb);
    const int32_t rc = g_bil.lookup(key, stream, what, [&](DevTable& e) -> int32_t {
        HostTables t;
        build_tables(d, sigma_color, sigma_space, t);
        const size_t n = t.dy.size();
        std::vector<uint32_t> blob(256 + 4 * n);
        memcpy(&blob[0], t.color.data(), 256 * 4);
        memcpy(&blob[256], t.space.data(), n * 4);
        for (size_t k = 0; k < n; ++k) { blob[256 + n + 2 * k] = (uint32_t)t.dy[k]; blob[256 + n + 2 * k + 1] = (uint32_t)t.dx[k]; }
        memcpy(&blob[256 + 3 * n], t.order.data(), n * 4);
        e.meta[0] = (int)n;
        e.meta[1] = t.radius;
        e.bytes = blob.size() * 4;
        KH_HIP(hipMalloc(&e.dev, e.bytes));
        const hipError_t err = hipMemcpy(e.dev, blob.data(), e.bytes, hipMemcpyHostToDevice);
        if (err != hipSuccess) return fail_hip(err, "hipMemcpy (bilateral tables)");  // ~DevTable frees the allocation
        return KH_OK;
    }, lease);
    if (rc != KH_OK) return rc;
    const uint32_t* base = (const uint32_t*)lease->dev;
    const size_t n = (size_t)lease->meta[0];
    out.color = (const float*)base;
    out.space = (const float*)(base + 256);
    out.taps = (const Tap*)(base + 256 + n);
    out.order = (const int*)(base + 256 + 3 * n);
    out.n = (int)n;
    out.radius = lease->meta[1];
    return KH_OK;
}

}  // namespace

extern "C" {

int32_t kh_spatial_gradient_f32(kh_stream_t stream, const float* src, float* dx, float* dy, int32_t cols, int32_t rows, int32_t channels,
                                int32_t kind, int32_t batch, int64_t src_stride, int64_t dst_stride) {
    const char* what = "kh_spatial_gradient_f32";
    KH_REQUIRE(kind == KH_GRAD_SOBEL || kind == KH_GRAD_SCHARR, KH_ERR_INVALID_ARG, "%s: unknown gradient kind %d", what, kind);
    if (int32_t rc = check_batch(what, cols, rows, channels, batch, 65535)) return rc;
    if (batch == 0) return KH_OK;
    KH_REQUIRE(src && dx && dy, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    KH_REQUIRE(src != dx && src != dy && dx != dy, KH_ERR_INVALID_ARG, "%s: src, dx and dy must be distinct images", what);
    const float a = kind == KH_GRAD_SOBEL ? 0.125f : 0.09375f, b = kind == KH_GRAD_SOBEL ? 0.25f : 0.3125f;
    const int64_t rowlen = (int64_t)cols * channels;
    const bool force_scalar = dev_opt(kOptGradScalar) == 1;  // test option: the unaligned fallback on aligned images
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dy)) % 16) == 0 &&
                         rowlen % 4 == 0 && (batch == 1 || (src_stride % 4 == 0 && dst_stride % 4 == 0));
    if (!force_scalar && aligned && channels <= 4) {
        const dim3 grid(cdiv(rowlen / 4, kBx), cdiv(rows, kBy), batch), blk(kBx, kBy);
        hipStream_t st = as_hip(stream);
        const long long ss = src_stride, ds = dst_stride;
        switch (channels) {
            case 1: hipLaunchKernelGGL(spatial_gradient_x4_kernel<1>, grid, blk, 0, st, src, dx, dy, (int)rows, (int)cols, a, b, ss, ds); break;
            case 2: hipLaunchKernelGGL(spatial_gradient_x4_kernel<2>, grid, blk, 0, st, src, dx, dy, (int)rows, (int)cols, a, b, ss, ds); break;
            case 3: hipLaunchKernelGGL(spatial_gradient_x4_kernel<3>, grid, blk, 0, st, src, dx, dy, (int)rows, (int)cols, a, b, ss, ds); break;
            default: hipLaunchKernelGGL(spatial_gradient_x4_kernel<4>, grid, blk, 0, st, src, dx, dy, (int)rows, (int)cols, a, b, ss, ds); break;
        }
        return check_launch(what);
    }
    hipLaunchKernelGGL(spatial_gradient_kernel, dim3(cdiv(rowlen, kBx), cdiv(rows, kBy), batch), dim3(kBx, kBy), 0, as_hip(stream), src, dx, dy,
                       (int)rows, (int)cols, (int)channels, a, b, (long long)src_stride, (long long)dst_stride);
    return check_launch(what);
}

// box_blur_fast_kernels_1d (P/filter/kernels.rs:151-170), f32 throughout; `as u8` / `as usize` saturate.
int32_t kh_box_blur_fast_kernels_1d(float sigma, int32_t kernels, int32_t* out) {
    KH_REQUIRE(out && kernels >= 0 && kernels <= 255, KH_ERR_INVALID_ARG, "kh_box_blur_fast_kernels_1d: bad argument (kernels %d)", kernels);
    const float n = (float)kernels;
    const float ideal_size = sqrtf(12.0f * sigma * sigma / n + 1.0f);
    float size_l = floorf(ideal_size);
    size_l -= fmodf(size_l, 2.0f) == 0.0f ? 1.0f : 0.0f;
    const float size_u = size_l + 2.0f;
    const float ideal_m = (12.0f * sigma * sigma - n * size_l * size_l - 4.0f * n * size_l - 3.0f * n) / (-4.0f * size_l - 4.0f);
    const float rm = roundf(ideal_m);
    const int m = rm != rm ? 0 : (rm <= 0.0f ? 0 : (rm >= 255.0f ? 255 : (int)rm));
    for (int i = 0; i < kernels; ++i) {
        const float s = i < m ? size_l : size_u;
        out[i] = s != s ? 0 : (s <= 0.0f ? 0 : (s >= 2147483520.0f ? INT32_MAX : (int32_t)s));
    }
    return KH_OK;
}

int32_t kh_fast_horizontal_filter_f32(kh_stream_t stream, const float* src, float* dst_transposed, int32_t cols, int32_t rows,
                                      int32_t channels, int32_t half, int32_t batch, int64_t src_stride, int64_t dst_stride) {
    const char* what = "kh_fast_horizontal_filter_f32";
    if (int32_t rc = check_batch(what, cols, rows, channels, batch, 65535)) return rc;
    // the reference reads `half` pixels past column 0 of every row without a bound: it indexes out of the image (panics) from
    // half >= cols on (separable_filter.rs:229-231)
    KH_REQUIRE(half >= 0 && half < cols, KH_ERR_INVALID_ARG, "%s: half kernel %d does not fit a %d-pixel row", what, half, cols);
    if (batch == 0) return KH_OK;
    KH_REQUIRE(src && dst_transposed && src != dst_transposed, KH_ERR_INVALID_ARG, "%s: null or aliased device pointer", what);
    const dim3 grid(cdiv((int64_t)rows * channels, kBlock), batch);
    // LDS-staged rows when a useful chunk fits: ~(256 / C + 2) rows x (T + 2 * half + 1) columns.  64 KiB keeps two blocks per CU;
    // wide boxes may take up to 150 KiB; beyond that the direct kernel runs.  kh_debug_set_option("hfilter_direct", 1) forces it (test option).
    const bool force_direct = dev_opt(kOptHfilterDirect) == 1;
    const int nrows_max = (kBlock + channels - 1) / channels + 1;
    auto chunk_for = [&](size_t budget) { return (int)(budget / (sizeof(float) * (size_t)nrows_max * channels)) - (2 * half + 1) - 1; };
    int T = chunk_for(64 * 1024);
    if (T < 16) T = chunk_for(150 * 1024);
    if (T > cols) T = cols;
    if (!force_direct && T >= 8 && channels <= kBlock) {
        const int pitch = ((T + 2 * half + 1) * channels) | 1;  // odd dword pitch: consecutive rows start on different banks
        const size_t bytes = sizeof(float) * (size_t)nrows_max * pitch;
        if (bytes > 48 * 1024)
            KH_HIP(hipFuncSetAttribute((const void*)fast_hfilter_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(fast_hfilter_lds_kernel, grid, dim3(kBlock), bytes, as_hip(stream), src, dst_transposed, (int)rows, (int)cols,
                           (int)channels, (int)half, T, pitch, (long long)src_stride, (long long)dst_stride);
        return check_launch(what);
    }
    hipLaunchKernelGGL(fast_hfilter_kernel, grid, dim3(kBlock), 0, as_hip(stream), src, dst_transposed, (int)rows, (int)cols, (int)channels,
                       (int)half, (long long)src_stride, (long long)dst_stride);
    return check_launch(what);
}

// box_blur_fast (P/filter/ops.rs:252-285): three rounds of (rows -> transposed scratch, scratch -> dst); the sizes from
// box_blur_fast_kernels_1d are used as HALF widths, as the reference does.  scratch: batch images of cols*rows*channels floats.
int32_t kh_box_blur_fast_f32(kh_stream_t stream, const float* src, float* dst, float* scratch, int32_t cols, int32_t rows,
                             int32_t channels, float sigma_x, float sigma_y, int32_t batch, int64_t src_stride, int64_t dst_stride) {
    const char* what = "kh_box_blur_fast_f32";
    if (int32_t rc = check_batch(what, cols, rows, channels, batch, 65535)) return rc;
    int32_t hx[3], hy[3];
    kh_box_blur_fast_kernels_1d(sigma_x, 3, hx);
    kh_box_blur_fast_kernels_1d(sigma_y, 3, hy);
    for (int i = 0; i < 3; ++i)
        KH_REQUIRE(hx[i] < cols && hy[i] < rows, KH_ERR_INVALID_ARG, "%s: box half widths (%d, %d) for sigma (%g, %g) do not fit a %dx%d image",
                   what, hx[i], hy[i], (double)sigma_x, (double)sigma_y, cols, rows);
    if (batch == 0) return KH_OK;
    KH_REQUIRE(src && dst && scratch, KH_ERR_INVALID_ARG, "%s: null device pointer", what);
    KH_REQUIRE(src != dst && src != scratch && dst != scratch, KH_ERR_INVALID_ARG, "%s: src, dst and scratch must be distinct", what);
    const int64_t image = (int64_t)cols * rows * channels;
    const float* in = src;
    int64_t in_stride = src_stride;
    for (int i = 0; i < 3; ++i) {
        if (int32_t rc = kh_fast_horizontal_filter_f32(stream, in, scratch, cols, rows, channels, hx[i], batch, in_stride, image)) return rc;
        if (int32_t rc = kh_fast_horizontal_filter_f32(stream, scratch, dst, rows, cols, channels, hy[i], batch, image, dst_stride)) return rc;
        in = dst;
        in_stride = dst_stride;
    }
    return KH_OK;
}

int32_t kh_median_blur_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t cols, int32_t rows, int32_t channels, int32_t ksize,
                          int32_t batch, int64_t src_stride, int64_t dst_stride) {
    const char* what = "kh_median_blur_u8";
    KH_REQUIRE(ksize == 3 || ksize == 5, KH_ERR_INVALID_ARG, "%s: invalid kernel length %d (3 or 5)", what, ksize);  // InvalidKernelLength
    if (int32_t rc = check_batch(what, cols, rows, channels, batch, 4)) return rc;
    if (batch == 0) return KH_OK;
    KH_REQUIRE(src && dst && src != dst, KH_ERR_INVALID_ARG, "%s: null or aliased device pointer", what);
    const dim3 grid(cdiv(cdiv(cols, 2), kBx), cdiv(rows, kBy), batch), blk(kBx, kBy);
    hipStream_t st = as_hip(stream);
    const long long ss = src_stride, ds = dst_stride;
    constexpr bool bytes_only = false;  // (the byte-wise selection network was the round-2 A/B partner of the packed one; not instantiated)
#define KH_MEDIAN_LAUNCH(K, CH)                                                                                                   \
    do {                                                                                                                          \
        if constexpr (bytes_only) hipLaunchKernelGGL((median_kernel<K, CH, false>), grid, blk, 0, st, src, dst, (int)rows, (int)cols, ss, ds); \
        else hipLaunchKernelGGL((median_kernel<K, CH, true>), grid, blk, 0, st, src, dst, (int)rows, (int)cols, ss, ds);          \
    } while (0)
    switch (ksize * 10 + channels) {
        case 31: KH_MEDIAN_LAUNCH(3, 1); break;
        case 32: KH_MEDIAN_LAUNCH(3, 2); break;
        case 33: KH_MEDIAN_LAUNCH(3, 3); break;
        case 34: KH_MEDIAN_LAUNCH(3, 4); break;
        case 51: KH_MEDIAN_LAUNCH(5, 1); break;
        case 52: KH_MEDIAN_LAUNCH(5, 2); break;
        case 53: KH_MEDIAN_LAUNCH(5, 3); break;
        default: KH_MEDIAN_LAUNCH(5, 4); break;
    }
#undef KH_MEDIAN_LAUNCH
    return check_launch(what);
}

// The tables cv2 would build (BilateralTables, bilateral.rs:80-170).  Returns the tap count through *ntaps; the arrays are
// filled only when capacity >= ntaps (call once with capacity 0 to size them).  color_weight has 256 entries.
int32_t kh_bilateral_tables(int32_t d, double sigma_color, double sigma_space, int32_t capacity, int32_t* radius, int32_t* ntaps,
                            int32_t* tap_dy, int32_t* tap_dx, float* space_weight, float* color_weight, int32_t* simd_order) {
    KH_REQUIRE(ntaps, KH_ERR_INVALID_ARG, "kh_bilateral_tables: null ntaps");
    KH_REQUIRE(bilateral_radius(d, sigma_space) <= kMaxBilateralRadius, KH_ERR_TOO_LARGE, "kh_bilateral_tables: window radius %d exceeds %d",
               bilateral_radius(d, sigma_space), kMaxBilateralRadius);
    HostTables t;
    build_tables(d, sigma_color, sigma_space, t);
    const int n = (int)t.dy.size();
    *ntaps = n;
    if (radius) *radius = t.radius;
    if (capacity < n) return KH_OK;
    KH_REQUIRE(tap_dy && tap_dx && space_weight && color_weight && simd_order, KH_ERR_INVALID_ARG, "kh_bilateral_tables: null output");
    memcpy(tap_dy, t.dy.data(), 4 * (size_t)n);
    memcpy(tap_dx, t.dx.data(), 4 * (size_t)n);
    memcpy(space_weight, t.space.data(), 4 * (size_t)n);
    memcpy(color_weight, t.color.data(), 4 * 256);
    memcpy(simd_order, t.order.data(), 4 * (size_t)n);
    return KH_OK;
}

int32_t kh_bilateral_filter_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t cols, int32_t rows, int32_t d, double sigma_color,
                               double sigma_space, int32_t batch, int64_t src_stride, int64_t dst_stride) {
    const char* what = "kh_bilateral_filter_u8";
    if (int32_t rc = check_batch(what, cols, rows, 1, batch, 1)) return rc;
    if (batch == 0) return KH_OK;
    KH_REQUIRE(src && dst && src != dst, KH_ERR_INVALID_ARG, "%s: null or aliased device pointer", what);
    if (sigma_color <= 1e-6 || sigma_space <= 1e-6) {  // cv2: degenerate sigmas copy the source through (bilateral.rs:188-200)
        const size_t image = (size_t)cols * rows;
        if (batch == 1) KH_HIP(hipMemcpyAsync(dst, src, image, hipMemcpyDeviceToDevice, as_hip(stream)));
        else KH_HIP(hipMemcpy2DAsync(dst, (size_t)dst_stride, src, (size_t)src_stride, image, (size_t)batch, hipMemcpyDeviceToDevice, as_hip(stream)));
        return KH_OK;
    }
    KH_REQUIRE(bilateral_radius(d, sigma_space) <= kMaxBilateralRadius, KH_ERR_TOO_LARGE, "%s: window radius %d exceeds %d", what,
               bilateral_radius(d, sigma_space), kMaxBilateralRadius);
    BilateralTab t;
    TableLease lease;  // keeps the table alive until the launch that reads it is enqueued and recorded
    if (int32_t rc = get_bilateral_tab(d, sigma_color, sigma_space, as_hip(stream), what, t, lease)) return rc;
    const int simd_end = cols >= 16 ? ((cols - 16) / 16) * 16 + 16 : 0;  // simd_region_end, bilateral.rs:99-106
    hipLaunchKernelGGL(bilateral_kernel, dim3(cdiv(cols, kBx), cdiv(rows, kBy), batch), dim3(kBx, kBy), 0, as_hip(stream), src, dst, (int)rows,
                       (int)cols, t, simd_end, (long long)src_stride, (long long)dst_stride);
    const int32_t rc = check_launch(what);
    lease->used_on(as_hip(stream));
    return rc;
}

}  // extern "C"
