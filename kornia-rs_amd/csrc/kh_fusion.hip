// Fused per-pixel pipelines (source -> maps -> sink) for gfx950.
//
// The reference's fusion engine (P/cuda/fusion.rs:196-520) generates CUDA source per pipeline shape
// and compiles it with NVRTC; values flow through registers between stages and every stage's
// parameters sit in one constant blob.  Here the same contract is met without a run-time compiler:
// the kernel is specialised at build time on (source kind, sink kind) and walks the map stages from
// a small stage program in the kernel arguments — the program lives in SGPRs, every branch on a
// stage kind is wave-uniform, so the value still flows through VGPRs from the source read to the
// sink write, with no intermediate memory.  Stage arithmetic is that of the reference's snippets
// (ReadU8RgbBilinear :520-590, Normalize :592-622, RgbToGray :624-643, WriteChwF32 :645-667,
// WriteC1F32 :669-690), uncontracted like its fmad=false build.
#include <math.h>
#include <string.h>

#include <string>

#include "kh_common.h"

using namespace kh;

namespace {

constexpr int kBx = 64, kBy = 4;
constexpr int kMaxMaps = 12;
constexpr int kMaxBatchPerLaunch = 32;  // per-image source pointers travel in the kernel arguments

struct MapStage { int kind; float f[6]; };

struct FusedProgram {
    const uint8_t* src[kMaxBatchPerLaunch];
    float* dst;
    long long dst_stride;  // elements per image
    int dw, dh, sw, sh;
    float ax, bx, ay, by;
    int nmaps;
    MapStage maps[kMaxMaps];
    XcdTiles tiles;
};

template <int SINK>
__global__ __launch_bounds__(kBx* kBy) void fused_pipeline_kernel(FusedProgram P) {
    unsigned bx_, by_, bz_;
    if (!xcd_tile(P.tiles, bx_, by_, bz_)) return;
    const int x = bx_ * kBx + threadIdx.x, y = by_ * kBy + threadIdx.y;
    if (x >= P.dw || y >= P.dh) return;
    const uint8_t* __restrict__ src = P.src[bz_];
    float* __restrict__ dst = P.dst + (long long)bz_ * P.dst_stride;

    // source: ReadU8RgbBilinear (fusion.rs:545-585)
    const float sxf = fmaxf(P.ax * (float)x + P.bx, 0.0f);
    const float syf = fmaxf(P.ay * (float)y + P.by, 0.0f);
    const unsigned sx0 = min((unsigned)sxf, (unsigned)P.sw - 1u), sy0 = min((unsigned)syf, (unsigned)P.sh - 1u);
    const unsigned sy1 = min(sy0 + 1u, (unsigned)P.sh - 1u);  // sx1 = min(sx0 + 1, sw - 1) is load_quad_u8's second pixel
    const float wx = sxf - (float)sx0, wy = syf - (float)sy0;
    const uint8_t* r0 = src + (size_t)sy0 * P.sw * 3u;
    const uint8_t* r1 = src + (size_t)sy1 * P.sw * 3u;
    const float w00 = (1.0f - wy) * (1.0f - wx), w01 = (1.0f - wy) * wx, w10 = wy * (1.0f - wx), w11 = wy * wx;
    // the two taps of a row are adjacent: one dword + one ushort load instead of six byte loads
    // sx1 = min(sx0 + 1, sw - 1) is load_quad_u8's second pixel (12 byte loads: 3.9 ms, 4 paired loads: 2.2 ms on
    // the 1080p -> 640 x 1024 probe, profiles/r01s_ab.log)
    const QuadU8 q = load_quad_u8<3>(r0, r1, (int)sx0, P.sw);
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        v[c] = w00 * (float)chan_u8(q.p00, c) + w01 * (float)chan_u8(q.p01, c) + w10 * (float)chan_u8(q.p10, c) +
               w11 * (float)chan_u8(q.p11, c);

    // maps (wave-uniform program walk)
    for (int i = 0; i < P.nmaps; ++i) {
        const MapStage& m = P.maps[i];
        if (m.kind == KH_FUSE_NORMALIZE) {  // :610-616
            v[0] = v[0] * m.f[0] + m.f[3];
            v[1] = v[1] * m.f[1] + m.f[4];
            v[2] = v[2] * m.f[2] + m.f[5];
        } else {  // KH_FUSE_RGB_TO_GRAY, :636-640
            const float g = 0.299f * v[0] + 0.587f * v[1] + 0.114f * v[2];
            v[0] = g; v[1] = g; v[2] = g;
        }
    }

    // sink
    const size_t di = (size_t)y * P.dw + x;
    // (streaming stores, kh_common.h::stream_store: the tensor is written once; a frame beyond the V#'s 2 GiB window keeps plain stores)
    const size_t plane = (size_t)P.dw * P.dh;
    constexpr int kPlanes = SINK == KH_FUSE_WRITE_CHW_F32 ? 3 : 1;
    if (plane * kPlanes * 4u <= 0x7fffffffu) {   // launch-uniform
        const __amdgpu_buffer_rsrc_t rs = stream_window(dst, (long long)(plane * kPlanes * 4u));
#pragma unroll
        for (int c = 0; c < kPlanes; ++c) {
            const uint32_t bits = __float_as_uint(v[c]);
            stream_store<1>(rs, (int)((di + c * plane) * 4u), &bits);
        }
    } else if constexpr (SINK == KH_FUSE_WRITE_CHW_F32) {  // :657-661
        dst[di] = v[0];
        dst[di + plane] = v[1];
        dst[di + 2u * plane] = v[2];
    } else {  // KH_FUSE_WRITE_C1_F32, :681
        dst[di] = v[0];
    }
}

}  // namespace

struct kh_fused_pipeline_s {
    FusedProgram prog;
    int sink;
    int batch;
    size_t src_bytes;   // bytes the source stage reads per image
    size_t out_elems;   // elements the sink writes per image
    std::string text;   // the stage program, for introspection
};

extern "C" {

int32_t kh_fused_pipeline_build(const kh_fused_stage* stages, int32_t nstages, int32_t dst_w, int32_t dst_h, int32_t batch,
                                int64_t out_elems_per_image, kh_fused_pipeline_t* out) {
    const char* what = "kh_fused_pipeline_build";
    KH_REQUIRE(out, KH_ERR_INVALID_ARG, "%s: null out pointer", what);
    *out = nullptr;
    KH_REQUIRE(stages && nstages >= 2, KH_ERR_INVALID_ARG, "invalid pipeline: need at least a source and a sink stage");
    KH_REQUIRE(batch >= 1, KH_ERR_INVALID_ARG, "invalid pipeline: batch must be >= 1");
    KH_REQUIRE(dst_w > 0 && dst_h > 0, KH_ERR_INVALID_ARG, "invalid pipeline: empty destination grid %dx%d", dst_w, dst_h);
    KH_REQUIRE(nstages - 2 <= kMaxMaps, KH_ERR_TOO_LARGE, "fused parameter blob exceeds %d map stages (needs %d)", kMaxMaps,
               nstages - 2);
    const kh_fused_stage& s0 = stages[0];
    const kh_fused_stage& sn = stages[nstages - 1];
    KH_REQUIRE(s0.kind == KH_FUSE_READ_U8RGB_BILINEAR, KH_ERR_INVALID_ARG, "invalid pipeline: stage 0 (kind %d) is not a source",
               s0.kind);
    KH_REQUIRE(sn.kind == KH_FUSE_WRITE_CHW_F32 || sn.kind == KH_FUSE_WRITE_C1_F32, KH_ERR_INVALID_ARG,
               "invalid pipeline: last stage (kind %d) is not a sink", sn.kind);
    const int sw = s0.u[0], sh = s0.u[1], rdw = s0.u[2], rdh = s0.u[3];
    KH_REQUIRE(sw > 0 && sh > 0 && rdw > 0 && rdh > 0, KH_ERR_INVALID_ARG, "invalid pipeline: source stage has an empty geometry");
    KH_REQUIRE((int64_t)sw * sh * 3 <= kI32Max && (int64_t)dst_w * dst_h * 3 <= kI32Max, KH_ERR_TOO_LARGE,
               "%s: image exceeds 32-bit indexing", what);

    auto* p = new kh_fused_pipeline_s();
    FusedProgram& g = p->prog;
    memset(&g, 0, sizeof(g));
    g.dw = dst_w; g.dh = dst_h; g.sw = sw; g.sh = sh;
    // half-pixel mapping s = a*d + b built on the host from the stage's own dst size (:538-543)
    const float axv = (float)sw / (float)rdw, ayv = (float)sh / (float)rdh;
    g.ax = axv; g.bx = 0.5f * axv - 0.5f; g.ay = ayv; g.by = 0.5f * ayv - 0.5f;
    char line[256];
    snprintf(line, sizeof line, "stage 0: read_u8rgb_bilinear sw=%d sh=%d ax=%.9g bx=%.9g ay=%.9g by=%.9g\n", sw, sh, g.ax, g.bx,
             g.ay, g.by);
    p->text = line;
    for (int i = 1; i + 1 < nstages; ++i) {
        const kh_fused_stage& s = stages[i];
        if (s.kind != KH_FUSE_NORMALIZE && s.kind != KH_FUSE_RGB_TO_GRAY) {
            delete p;
            return fail(KH_ERR_INVALID_ARG, "invalid pipeline: stage %d (kind %d) is not a map", i, s.kind);
        }
        MapStage& m = g.maps[g.nmaps++];
        m.kind = s.kind;
        for (int k = 0; k < 6; ++k) m.f[k] = s.f[k];
        if (s.kind == KH_FUSE_NORMALIZE)
            snprintf(line, sizeof line, "stage %d: normalize scale=(%.9g, %.9g, %.9g) bias=(%.9g, %.9g, %.9g)\n", i, s.f[0], s.f[1],
                     s.f[2], s.f[3], s.f[4], s.f[5]);
        else
            snprintf(line, sizeof line, "stage %d: rgb_to_gray\n", i);
        p->text += line;
    }
    p->sink = sn.kind;
    p->batch = batch;
    p->src_bytes = (size_t)sw * sh * 3;
    p->out_elems = (size_t)dst_w * dst_h * (sn.kind == KH_FUSE_WRITE_CHW_F32 ? 3 : 1);
    snprintf(line, sizeof line, "stage %d: %s  [grid %dx%d, batch %d]\n", nstages - 1,
             sn.kind == KH_FUSE_WRITE_CHW_F32 ? "write_chw_f32" : "write_c1_f32", dst_w, dst_h, batch);
    p->text += line;
    if (batch > 1 && (uint64_t)out_elems_per_image < p->out_elems) {  // :388-394
        const size_t need = p->out_elems;
        delete p;
        return fail(KH_ERR_INVALID_ARG,
                    "invalid pipeline: out_elems_per_image %lld is smaller than the sink's per-image output (%zu elements)",
                    (long long)out_elems_per_image, need);
    }
    g.dst_stride = batch > 1 ? out_elems_per_image : (long long)p->out_elems;
    *out = p;
    return KH_OK;
}

int32_t kh_fused_pipeline_launch(kh_fused_pipeline_t p, kh_stream_t stream, const uint8_t* const* srcs, int32_t nsrcs,
                                 int64_t src_bytes_each, float* dst, int64_t dst_elems) {
    const char* what = "kh_fused_pipeline_launch";
    KH_REQUIRE(p && srcs && dst, KH_ERR_INVALID_ARG, "%s: null argument", what);
    KH_REQUIRE(nsrcs == p->batch, KH_ERR_INVALID_ARG, "invalid pipeline: pipeline built for batch %d, got %d sources", p->batch,
               nsrcs);
    KH_REQUIRE(src_bytes_each <= 0 || (uint64_t)src_bytes_each >= p->src_bytes, KH_ERR_SLICE_TOO_SMALL,
               "%s: source holds %lld bytes, the source stage reads %zu", what, (long long)src_bytes_each, p->src_bytes);
    const uint64_t need = (uint64_t)(p->batch - 1) * (uint64_t)p->prog.dst_stride + p->out_elems;
    KH_REQUIRE(dst_elems <= 0 || (uint64_t)dst_elems >= need, KH_ERR_SLICE_TOO_SMALL,
               "%s: destination holds %lld elements, the pipeline writes %llu", what, (long long)dst_elems,
               (unsigned long long)need);
    for (int i = 0; i < nsrcs; ++i) KH_REQUIRE(srcs[i], KH_ERR_INVALID_ARG, "%s: null source pointer %d", what, i);
    hipStream_t st = as_hip(stream);
    const dim3 blk(kBx, kBy);
    for (int base = 0; base < nsrcs; base += kMaxBatchPerLaunch) {
        FusedProgram g = p->prog;
        const int n = nsrcs - base < kMaxBatchPerLaunch ? nsrcs - base : kMaxBatchPerLaunch;
        for (int i = 0; i < n; ++i) g.src[i] = srcs[base + i];
        g.dst = dst + (long long)base * g.dst_stride;
        g.tiles = xcd_tiles(cdiv(g.dw, kBx), cdiv(g.dh, kBy), (unsigned)n, cdiv(g.dw, kBx) * 8);
        KH_REQUIRE(g.tiles.total > 0, KH_ERR_TOO_LARGE, "%s: batch x tiles exceeds one launch", what);
        if (p->sink == KH_FUSE_WRITE_CHW_F32)
            hipLaunchKernelGGL(fused_pipeline_kernel<KH_FUSE_WRITE_CHW_F32>, xcd_grid(g.tiles), blk, 0, st, g);
        else
            hipLaunchKernelGGL(fused_pipeline_kernel<KH_FUSE_WRITE_C1_F32>, xcd_grid(g.tiles), blk, 0, st, g);
    }
    return check_launch(what);
}

int32_t kh_fused_pipeline_describe(kh_fused_pipeline_t p, char* buf, size_t n) {
    KH_REQUIRE(p && buf && n > 0, KH_ERR_INVALID_ARG, "kh_fused_pipeline_describe: null argument");
    snprintf(buf, n, "%s", p->text.c_str());
    return (int32_t)p->text.size();
}

void kh_fused_pipeline_destroy(kh_fused_pipeline_t p) { delete p; }

}  // extern "C"
