// Shared host/device helpers for libkornia_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "kornia_hip.h"

namespace kh {

// Thread-local error text behind kh_last_error().
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int32_t fail(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int32_t fail_hip(hipError_t e, const char* what);

inline hipStream_t as_hip(kh_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline hipEvent_t as_hip(kh_event_t e) { return reinterpret_cast<hipEvent_t>(e); }

// Kernel launches report asynchronous launch-time failures through hipGetLastError.
inline int32_t check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? KH_OK : fail_hip(e, what);
}

// Scratch for the few operators that need an intermediate (separable u8 resize, wide u8 blurs, per-row warp spans, Lanczos axis
// tables).  Taken from the workspace the caller registered for this stream (kh_stream_set_workspace) when it is large enough —
// then the call allocates nothing and can be captured into a graph — otherwise from the stream-ordered pool, as the reference's
// adapters do per call (P/filter/cuda.rs:119, P/resize/cuda.rs:262); under stream capture the pool path is refused with the
// number of bytes to register.  Releases pool scratch (stream-ordered) when it goes out of scope.
struct Scratch {
    void* ptr = nullptr;
    kh_stream_t stream = nullptr;
    bool pooled = false;
    Scratch() = default;
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    ~Scratch();
    template <typename T> T* as() const { return static_cast<T*>(ptr); }
};
int32_t get_scratch(kh_stream_t stream, size_t bytes, const char* what, Scratch& out);

constexpr int kWave = 64;       // gfx950 wavefront
constexpr int kBlock = 256;     // 4 waves: one per SIMD of a CU
constexpr int64_t kI32Max = 2147483647LL;

inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by linear id and every XCD has its
// own L2, so a plain (x, y, image) grid puts tiles that share halo rows / columns on different L2s
// and the shared lines are fetched from the fabric once per XCD (r01h PMC: +39 % on the 7x7 filter,
// +47 % on warp_perspective).  XcdTiles cuts the row-major tile list into runs of `run` consecutive
// tiles and deals the runs to the XCDs, so the tiles of a run — neighbours — execute on one L2 at
// about the same time, while the 8 XCDs stay within 8 runs of each other in memory.  run =
// kXcdEighth hands each XCD one contiguous eighth of the launch.  Measured (r01i/r01j, same box A/B):
// the rolling filter gains 7 % from eighths (9.87 -> 9.17 ms on C4) and nothing from short runs; the
// gathers are neutral with short runs and resize 1080p->224 LOSES 13 % with eighths, so gathers use
// runs of 8 tile rows.
constexpr unsigned kXcdEighth = ~0u;
// Division of a block id (< 2^31) by a launch constant without the ~8-instruction-per-quotient float
// reciprocal sequence the compiler emits for a run-time divisor (three of them were ~20 % of the VALU work
// of a gather wave).  Granlund-Montgomery: for n < 2^31 and l = ceil(log2 d), m = ceil(2^(31+l) / d) fits
// 32 bits and floor(n / d) == (n * m) >> (31 + l) == umulhi(n, m) >> (l - 1).  d == 1 is encoded as m == 0.
// The operands are wave-uniform, so this is two scalar instructions.  tests/test_abi.py sweeps it.
struct FastDiv { uint32_t d, m, sh; };
inline FastDiv fast_div(uint32_t d) {
    FastDiv f{d, 0u, 0u};
    if (d <= 1) return f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;  // ceil(log2 d), >= 1
    f.m = (uint32_t)(((1ull << (31 + l)) + d - 1) / d);
    f.sh = l - 1;
    return f;
}
__host__ __device__ __forceinline__ uint32_t fast_quot(uint32_t n, const FastDiv& f) {
    return f.m ? (uint32_t)(((uint64_t)n * f.m) >> 32) >> f.sh : n;
}

// ---- batches whose images are NOT equally spaced: pointer lists --------------------------------------------------------------------
// The reference's operators take ONE image per call (`resize(&Image, &mut Image)`, P/resize/mod.rs:114-132) and its batch entry a
// slice of separately allocated frame buffers (`run_raw_batch(frames: &[&CudaSlice<u8>], ..)`, P/preprocess.rs:1258-1282) that it
// walks with one launch per frame.  Here the `*_list` entry points take HOST arrays of n device pointers and launch once per kListMax
// images: the (source, destination) bases of a launch travel BY VALUE in the kernel arguments (2 KiB), where a block picks its image's
// pair with one scalar load — no device-side table to allocate, upload or keep alive, nothing a stream capture cannot record.
// Measured on the north star (profiles/r06a_ubench_nv12_one_store.txt): 1024 frame bases through such a table, 448 or 128 per launch,
// 4.374 / 4.373 ms against 4.373 ms for base + k * stride.  The short-lived gather / preprocess kernels have a separate LIST
// instantiation (selecting at run time inside one kernel cost their equally spaced launches up to 8 %, kh_geom.hip); the strip-walking
// filters, whose waves live for hundreds of rows, take the list as their last argument and select at run time (`listed` is
// launch-uniform: a scalar branch; equally spaced launches pass a zeroed list) — same-box A/B within noise (profiles/r06d, r06i).
constexpr int kListMax = 128;
struct PtrList { const void* src[kListMax]; void* dst[kListMax]; };
// What a batched launcher is handed: `n` images at src + k * ss / dst + k * ds (ELEMENTS of the operator's type, or bytes where the
// entry says so), or — `list` — at srcs[k] / dsts[k].
struct BatchRef {
    const void* src; void* dst; int64_t ss, ds; int n;
    const void* const* srcs; void* const* dsts;
    bool list;                               // a `_list` entry made this batch (the arrays themselves may still be NULL: check_list)
    bool listed() const { return list; }
};
inline BatchRef strided_batch(const void* src, void* dst, int n, int64_t ss, int64_t ds) { return BatchRef{src, dst, ss, ds, n, nullptr, nullptr, false}; }
template <typename T>
inline BatchRef listed_batch(const T* const* srcs, T* const* dsts, int n) {
    return BatchRef{nullptr, nullptr, 0, 0, n, reinterpret_cast<const void* const*>(srcs), reinterpret_cast<void* const*>(dsts), true};
}
// every pointer of a list present (n > 0); `what` names the entry in the message
int32_t check_list(const char* what, const void* const* srcs, void* const* dsts, int n);
// do all the images of the batch satisfy `align` bytes (sources and destinations; strides given in units of `elem` bytes)?
bool batch_aligned(const BatchRef& b, size_t align, size_t elem);
// Runs `launch(chunk, first, lst)` once for a strided batch (chunk == b, lst zeroed) or once per kListMax images of a list (chunk =
// that slice re-based at its first image, `first` = index of its first image in the whole batch); stops at the first failure.
template <typename F>
int32_t for_each_launch(const BatchRef& b, F&& launch) {
    static const PtrList kNoList{};
    if (!b.listed()) return launch(b, 0, kNoList);
    for (int first = 0; first < b.n; first += kListMax) {
        PtrList lst{};
        const int n = b.n - first < kListMax ? b.n - first : kListMax;
        for (int k = 0; k < n; ++k) { lst.src[k] = b.srcs[first + k]; lst.dst[k] = b.dsts[first + k]; }
        const BatchRef chunk{lst.src[0], lst.dst[0], 0, 0, n, b.srcs + first, b.dsts + first, true};
        if (int32_t rc = launch(chunk, first, lst)) return rc;
    }
    return KH_OK;
}
// device side: the bases of image `k` of the launch
template <typename T>
__device__ __forceinline__ const T* list_src(const PtrList& l, bool listed, const T* base, long long stride, unsigned k) {
    return listed ? static_cast<const T*>(l.src[k]) : base + (long long)k * stride;
}
template <typename T>
__device__ __forceinline__ T* list_dst(const PtrList& l, bool listed, T* base, long long stride, unsigned k) {
    return listed ? static_cast<T*>(l.dst[k]) : base + (long long)k * stride;
}

// ---- development / test options ---------------------------------------------------------------------------------------------------
// Nothing in this library reads the environment (round 3 had 26 getenv knobs, several on launch paths).  The ALTERNATE kernels a
// launcher can route to — the fallbacks other geometries, alignments or channel counts take anyway: IEEE division, the four-tap
// sampler, the LDS-tile filter, the per-pixel warps ... — can be FORCED by a test through kh_debug_set_option(name, value)
// (include/kornia_hip_testing.h — not part of the drop-in boundary), so that the parity tests reach them on convenient inputs.  The
// options are per THREAD (a launcher reads those of the thread calling it): one thread-local load where a launcher decides, nothing
// process-wide to reroute other callers; -1 = unset (the production choice).  Variants that were measured and rejected are not in
// the library at all.
enum DevOpt : int {
    kOptPreIeeeDiv, kOptPreGrid, kOptPreQuads, kOptFilterForceTile, kOptFilterFourColumns, kOptGradScalar, kOptHfilterDirect,
    kOptResizeU8Gather, kOptPyrDirect, kOptPyrRoll, kOptMorphDirect, kOptMorphRoll, kOptU8BlurRgb, kOptU8BlurSwar, kOptWarpU8Direct, kOptWarpU8Spans, kOptWarpU8Rows, kOptResizeRows, kOptWarpF32Px, kOptResizeU8Px, kOptRowStores, kOptPreF16Lut,
    kOptCount
};
int dev_opt(DevOpt o);               // kh_runtime.hip: the calling thread's value
// kh_pyramid_morph.hip: resize_fast_u8's exact 2x bilinear upscale on the rolling pyrup kernels (false = not taken; `rc` = the launch's result)
bool resize_up2_u8_rolling(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int sw, int sh, int channels, int batch, int64_t ss, int64_t ds,
                           const char* what, int32_t& rc, bool nearest, bool opencv = false);   // opencv: resize_opencv_u8's INTER_LINEAR arithmetic
void set_dev_opt(int o, int value);  // kh_debug_set_option only

struct XcdTiles { unsigned tiles_x, tiles_y, total, run; FastDiv by_run, by_img, by_row; };
constexpr int kXcds = 8;
inline XcdTiles xcd_tiles(unsigned tiles_x, unsigned tiles_y, unsigned images, unsigned run) {
    const uint64_t total = (uint64_t)tiles_x * tiles_y * images;
    XcdTiles t{tiles_x, tiles_y, (unsigned)total, 0, FastDiv{1, 0, 0}, fast_div(tiles_x * tiles_y), fast_div(tiles_x)};
    if (total > 0x7ff00000ull) t.total = 0;  // caller rejects (KH_ERR_TOO_LARGE)
    else {
        if (run == kXcdEighth) run = (unsigned)((total + kXcds - 1) / kXcds);
        if (run > 1 && total > run) { t.run = run; t.by_run = fast_div(run); }
    }
    return t;
}
inline dim3 xcd_grid(const XcdTiles& t) {
    if (!t.run) return dim3(t.total);
    const unsigned group = kXcds * t.run;
    return dim3((t.total + group - 1) / group * group);
}
__device__ __forceinline__ bool xcd_tile(const XcdTiles& t, unsigned& bx, unsigned& by, unsigned& bz) {
    const unsigned b = blockIdx.x;
    unsigned id = b;
    if (t.run) {
        const unsigned xcd = b % kXcds, slot = b / kXcds, grp = fast_quot(slot, t.by_run);
        id = (grp * kXcds + xcd) * t.run + (slot - grp * t.run);
    }
    if (id >= t.total) return false;
    bz = fast_quot(id, t.by_img);
    const unsigned r = id - bz * (t.tiles_x * t.tiles_y);
    by = fast_quot(r, t.by_row);
    bx = r - by * t.tiles_x;
    return true;
}

// Streaming stores through the buffer path.  The gfx940-family cache-policy bits travel in the `aux` operand of the raw buffer
// builtins (1 = sc0, 2 = nt, 16 = sc1), so the COMPILER sees the store and handles hazards / waitcnts — unlike inline asm, where a
// missing wait state between a 16-byte store and the next VALU write of its data registers corrupted 15 % of the output
// (profiles/r02c_ubench_nv12.txt).  `sc0 sc1 nt` = write-through at system scope + non-temporal: the lines leave the L2 for memory
// in arrival order instead of waiting for an LRU eviction.  Measured on the north-star store pattern: 4.44-4.51 ms against 4.70-4.80
// with the plain non-temporal global store, +4 % on a flat fill (profiles/r02f_ubench_nv12.txt).
// Offsets are 32-bit and range-checked by the hardware against `bytes` (out-of-range lanes are dropped, not faulted); keep the
// per-plane offset in the VECTOR offset: with an SGPR soffset ROCm 7.2 schedules a packed VALU write of the data registers into
// the slot right after the store — a gfx9 hazard — and 2 % of one plane came out wrong (profiles/r02d_ubench_nv12.txt).
constexpr int kAuxSc0 = 1, kAuxNt = 2, kAuxSc1 = 16;
constexpr int kAuxStream = kAuxSc0 | kAuxSc1 | kAuxNt;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);  // raw buffer, 32-bit data format
}

// A block-local (or image-local) streaming output window: a V# whose base is a WAVE-UNIFORM pointer and whose range covers what is
// left of the destination from there (clamped to 2 GiB; offsets are 32-bit).  Stores past the range are dropped by the hardware, so
// a tail needs no branch.  `stream_store` writes 4 / 8 / 12 / 16 bytes with the write-through non-temporal policy (kAuxStream).
// Measured: +8.6 % on a 12-byte-in / 12-byte-out f32 map, +-1 % on read-dominated maps (profiles/r02m_ubench_maps.txt).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stream_window(const void* uniform_base, long long bytes_left) {
    return buffer_rsrc(uniform_base, (uint32_t)(bytes_left < 0 ? 0 : (bytes_left > 0x7fffffffLL ? 0x7fffffffLL : bytes_left)));
}
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3_t __attribute__((ext_vector_type(3)));
template <int NDW>
__device__ __forceinline__ void stream_store(__amdgpu_buffer_rsrc_t rs, int byte_off, const uint32_t* w) {
    static_assert(NDW >= 1 && NDW <= 4, "1..4 dwords");
    if constexpr (NDW == 1) __builtin_amdgcn_raw_buffer_store_b32(w[0], rs, byte_off, 0, kAuxStream);
    else if constexpr (NDW == 2) __builtin_amdgcn_raw_buffer_store_b64((u32x2_t{w[0], w[1]}), rs, byte_off, 0, kAuxStream);
    else if constexpr (NDW == 3) __builtin_amdgcn_raw_buffer_store_b96((u32x3_t{w[0], w[1], w[2]}), rs, byte_off, 0, kAuxStream);
    else __builtin_amdgcn_raw_buffer_store_b128((u32x4_t{w[0], w[1], w[2], w[3]}), rs, byte_off, 0, kAuxStream);
}

// ROW-structured kernels (a wave owns a fixed column segment and walks down the rows) and the streaming policy (round 6,
// profiles/r06zq_row_stores.txt): a write-through store pays for every 128-byte line it writes PARTIALLY, and when the rows of an
// image are not whole lines (row bytes % 128 != 0, or an image base off a line) every wave segment of every row starts and ends inside
// a line.  The f32 rolling filters then ran 1.5x slower than with ordinary write-back stores (3839-pixel rows: 0.41 vs 0.26 ms per 16
// 4K planes; 1001 x 3: 0.62 vs 0.43), while on line-aligned rows the streaming policy wins by 3-30 %.  Launchers of such kernels pass
// `plain_row_stores(...)` to the kernel, which stores through `row_store` (a wave-uniform branch around the two policies).
template <int NDW>
__device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t rs, int byte_off, const uint32_t* w, int plain) {
    if (!plain) { stream_store<NDW>(rs, byte_off, w); return; }
    static_assert(NDW >= 1 && NDW <= 4, "1..4 dwords");
    if constexpr (NDW == 1) __builtin_amdgcn_raw_buffer_store_b32(w[0], rs, byte_off, 0, 0);
    else if constexpr (NDW == 2) __builtin_amdgcn_raw_buffer_store_b64((u32x2_t{w[0], w[1]}), rs, byte_off, 0, 0);
    else if constexpr (NDW == 3) __builtin_amdgcn_raw_buffer_store_b96((u32x3_t{w[0], w[1], w[2]}), rs, byte_off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128((u32x4_t{w[0], w[1], w[2], w[3]}), rs, byte_off, 0, 0);
}
// 1 = write-back stores: the destination's rows are not whole 128-byte lines (test option row_stores: 0 = streaming, 1 = write-back always)
inline int plain_row_stores(int64_t row_bytes, const void* dst, int64_t image_stride_bytes, int batch) {
    const int opt = dev_opt(kOptRowStores);
    if (opt >= 0) return opt ? 1 : 0;
    return (row_bytes % 128 != 0 || reinterpret_cast<uintptr_t>(dst) % 128 != 0 || (batch > 1 && image_stride_bytes % 128 != 0)) ? 1 : 0;
}

// Unaligned 2/4/8-byte global accesses (fine on gfx950; the compiler emits single dword/dwordx2 ops).
typedef uint16_t u16_unaligned __attribute__((aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint64_t u64_unaligned __attribute__((aligned(1)));
// s_waitcnt vmcnt(0) (expcnt / lgkmcnt untouched): every vector-memory load and store of this wave has completed
__device__ __forceinline__ void wait_vmcnt0() { __builtin_amdgcn_s_waitcnt(0x0F70); }
typedef u32x4_t u32x4_unaligned __attribute__((aligned(1)));

// The four u8 taps of a bilinear sample — pixels x0 and x1 = min(x0 + 1, w - 1) of two rows, C channels
// each — with one or two wide loads per row instead of 2*C byte loads.  Gathers on this chip are bound by
// the number of vector-memory instructions per pixel, and the two taps of a row are adjacent in memory.
// The pair loads are branch-free (base clamped so both pixels are inside the row, halves selected
// afterwards), and BOTH rows are fetched under ONE uniform width test: a load inside its own branch, even
// a uniform one, makes the compiler wait for it before issuing the next (measured 3x on the fused
// pipeline: 2.2 vs 6.4 ms).
// Pixels come back PACKED (channel c of a pixel = bits [8c, 8c+8) of its word): passing small arrays
// through the helper made the compiler park them in LDS (promote-alloca) in some kernels — 2.3x slower
// than the byte loads it replaced (profiles/r01s_ab.log).
// a * b + c on 24-bit operands in ONE v_mad_u32_u24.  Written as `__umul24(a, b) + c` the compiler merges the two multiply-adds of a
// channel into mul + mul + add3 (three instructions instead of two): the blend is the inner loop of a VALU-bound kernel.
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
#ifdef KH_HOSTSIM
    return __umul24(a, b) + c;
#else
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#endif
}

// prev / next lane's value by DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1): lane i takes lane i -/+ 1 and the END lane,
// which has no source, keeps `end` — where a wave's halo value can be waiting.  A vector-ALU move, not an LDS-crossbar
// instruction like ds_bpermute.  Checked on gfx950 by scripts/ubench/dpp_wave_shift.hip.
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v, uint32_t end) { return (uint32_t)__builtin_amdgcn_update_dpp((int)end, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t from_lane_above(uint32_t v, uint32_t end) { return (uint32_t)__builtin_amdgcn_update_dpp((int)end, (int)v, 0x130, 0xf, 0xf, false); }
// Ragged rows in the single-channel u8 rolling kernels (round 6; kh_pyramid_morph.hip, kh_u8.hip).  Those kernels give a lane SIXTEEN (pyrup: eight) pixels of a row with one
// wide load and took only widths that are whole lanes; every other width fell to the tile kernels, 2.4-4x slower (1000-pixel rows,
// profiles/r06zr_misaligned_rows.txt).  On a wave that reaches the row end every lane now loads the sixteen bytes at pc = min(p, w - 16)
// and re-indexes them: lane byte j (pixel p + j) <- loaded byte map(p + j) - pc, or the border constant — which covers the lane the row
// ends in (its own pixels shifted, then border pixels), the lane after it (border pixels only: what its neighbour's windows reach) and,
// with identity selectors, every lane before.  Per output dword the four source indices lie within four consecutive bytes (ascending
// pixels, a reflection, or one replicated pixel), i.e. in ONE of the dword pairs (L1:L0), (L2:L1), (L3:L2): three v_perm_b32 and two
// selects with per-lane selectors computed once.
struct Remap16 { uint32_t sel[4], cmask[4]; int k[4]; };
template <class MapFn>
__device__ __forceinline__ Remap16 remap16_setup(int p, int pc, MapFn map) {   // map(x): source pixel index, or < 0 for the constant
    Remap16 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int id[4], lo = 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = map(p + 4 * c + j);
            id[j] = m < 0 ? -1 : min(max(m - pc, 0), 15);
            lo = id[j] < 0 ? lo : min(lo, id[j]);
        }
        const int k = min(lo >> 2, 2);
        uint32_t sel = 0, cm = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sel |= (uint32_t)(id[j] < 0 ? 0 : min(max(id[j] - 4 * k, 0), 7)) << (8 * j);
            cm |= id[j] < 0 ? 0xffu << (8 * j) : 0u;
        }
        r.sel[c] = sel; r.cmask[c] = cm; r.k[c] = k;
    }
    return r;
}
__device__ __forceinline__ void remap16_apply(const Remap16& r, uint32_t (&L)[4], uint32_t cv) {
    uint32_t o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t r0 = __builtin_amdgcn_perm(L[1], L[0], r.sel[c]), r1 = __builtin_amdgcn_perm(L[2], L[1], r.sel[c]), r2 = __builtin_amdgcn_perm(L[3], L[2], r.sel[c]);
        const uint32_t v = r.k[c] == 0 ? r0 : (r.k[c] == 1 ? r1 : r2);
        o[c] = (v & ~r.cmask[c]) | (cv & r.cmask[c]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) L[c] = o[c];
}
// the first `n` (1..15) bytes of four dwords to a byte-aligned address (the lane a ragged row ends in)
__device__ __forceinline__ void store_head_bytes(uint8_t* o, const uint32_t (&w)[4], int n) {
#pragma unroll
    for (int b = 0; b < 15; ++b)
        if (b < n) o[b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
}

// Four pixels of C = 3 / 4 bytes <-> one dword per channel (pixel j = byte j): the planar u8 rolling kernels' (kh_u8.hip, kh_pyramid_morph.hip)
// view of a lane's quad.  RGB: six v_perm_b32 in, five out; RGBA: a 4 x 4 byte transpose, eight each way.
template <int C>
__device__ __forceinline__ void deinterleave_quad(const uint32_t* d, uint32_t (&ch)[C]) {   // four pixels of C bytes -> one dword per channel (pixel j = byte j)
    if constexpr (C == 3) {
        constexpr uint32_t in1[3] = {0x0c060300u, 0x0c070401u, 0x0c0c0502u}, in2[3] = {0x05020100u, 0x06020100u, 0x07040100u};
#pragma unroll
        for (int c = 0; c < 3; ++c) ch[c] = __builtin_amdgcn_perm(d[2], __builtin_amdgcn_perm(d[1], d[0], in1[c]), in2[c]);
    } else {
        const uint32_t a = __builtin_amdgcn_perm(d[1], d[0], 0x05010400u), b = __builtin_amdgcn_perm(d[1], d[0], 0x07030602u);
        const uint32_t c_ = __builtin_amdgcn_perm(d[3], d[2], 0x05010400u), e = __builtin_amdgcn_perm(d[3], d[2], 0x07030602u);
        ch[0] = __builtin_amdgcn_perm(c_, a, 0x05040100u); ch[1] = __builtin_amdgcn_perm(c_, a, 0x07060302u);
        ch[2] = __builtin_amdgcn_perm(e, b, 0x05040100u); ch[3] = __builtin_amdgcn_perm(e, b, 0x07060302u);
    }
}
template <int C>
__device__ __forceinline__ void interleave_quad(const uint32_t (&pl)[C], uint32_t (&w)[C]) {   // the inverse: C dwords of four whole pixels
    if constexpr (C == 3) {
        const uint32_t rg = __builtin_amdgcn_perm(pl[1], pl[0], 0x05010400u), rg2 = __builtin_amdgcn_perm(pl[1], pl[0], 0x07030602u);
        w[0] = __builtin_amdgcn_perm(pl[2], rg, 0x02040100u);
        w[1] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(pl[2], rg, 0x0c0c0503u), rg2, 0x01000504u);
        w[2] = __builtin_amdgcn_perm(pl[2], rg2, 0x07030206u);
    } else {
        uint32_t t[4];
        deinterleave_quad<4>(pl, t);   // a 4 x 4 byte transpose is its own inverse
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = t[c];
    }
}
struct QuadU8 { uint32_t p00, p01, p10, p11; };
__device__ __forceinline__ uint32_t chan_u8(uint32_t px, int c) { return (px >> (8 * c)) & 0xffu; }

template <int C>
__device__ __forceinline__ void load_pair_u8_wide(const uint8_t* __restrict__ row, int x0, int w, uint32_t& p0, uint32_t& p1) {
    const int xb = min(x0, w - 2);   // w >= 2
    const uint8_t* p = row + (unsigned)(xb * C);
    uint32_t a, b;
    if constexpr (C == 1) {
        const uint32_t v = *reinterpret_cast<const u16_unaligned*>(p);
        a = v & 0xffu; b = v >> 8;
    } else if constexpr (C == 2) {
        const uint32_t v = *reinterpret_cast<const u32_unaligned*>(p);
        a = v & 0xffffu; b = v >> 16;
    } else if constexpr (C == 3) {
        const uint32_t lo = *reinterpret_cast<const u32_unaligned*>(p);
        const uint32_t hi = *reinterpret_cast<const u16_unaligned*>(p + 4);
        a = lo & 0xffffffu; b = (lo >> 24) | (hi << 8);
    } else {
        const uint64_t v = *reinterpret_cast<const u64_unaligned*>(p);
        a = (uint32_t)v; b = (uint32_t)(v >> 32);
    }
    p0 = x0 != xb ? b : a;  // x0 is the last column: both taps are the pair's second pixel
    p1 = b;
}
template <int C>
__device__ __forceinline__ uint32_t load_px_u8(const uint8_t* __restrict__ p) {
    uint32_t v = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) v |= (uint32_t)p[c] << (8 * c);
    return v;
}
template <int C>
__device__ __forceinline__ QuadU8 load_quad_u8(const uint8_t* __restrict__ row0, const uint8_t* __restrict__ row1, int x0, int w) {
    QuadU8 q;
    if (w >= 2) {  // uniform
        load_pair_u8_wide<C>(row0, x0, w, q.p00, q.p01);
        load_pair_u8_wide<C>(row1, x0, w, q.p10, q.p11);
    } else {       // a 1-pixel row replicates its pixel
        q.p00 = q.p01 = load_px_u8<C>(row0);
        q.p10 = q.p11 = load_px_u8<C>(row1);
    }
    return q;
}

}  // namespace kh

#define KH_HIP(call)                                           \
    do {                                                       \
        hipError_t kh_e_ = (call);                             \
        if (kh_e_ != hipSuccess) return kh::fail_hip(kh_e_, #call); \
    } while (0)

#define KH_REQUIRE(cond, code, ...)                            \
    do {                                                       \
        if (!(cond)) return kh::fail((code), __VA_ARGS__);     \
    } while (0)
