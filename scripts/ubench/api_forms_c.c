/* Round 6: BASELINE configs[1] (bilinear 1920x1080x3 f32 -> 224x224, N = 256 separately allocated images) driven from plain C through
 * the C ABI only — what a Rust host pays without an interpreter in the way:
 *   eager   256 x kh_resize_f32(.., batch = 1)            one launch per image (the reference's per-image operator, resize/mod.rs:114-132)
 *   graph   the same 256 calls captured once, replayed    (kornia-py's cuda.Graph, cuda_ext/mod.rs:1684-1790)
 *   list    kh_resize_f32_list(srcs, dsts, 256)           two launches of 128 (src, dst) pairs
 *   strided kh_resize_f32(.., batch = 256, stride)        equally spaced (one allocation), one launch
 * gcc -std=c99 -O2 -Iinclude scripts/ubench/api_forms_c.c -Lkornia-rs_amd/lib -lkornia_hip -Wl,-rpath,$PWD/kornia-rs_amd/lib -o scripts/ubench/bin/api_forms_c */
#include <stdio.h>
#include <stdlib.h>

#include "kornia_hip.h"

#define CK(x) do { int32_t rc_ = (x); if (rc_ != KH_OK) { char m_[256]; kh_last_error(m_, sizeof m_); printf("FAILED %s:%d rc %d: %s\n", __FILE__, __LINE__, rc_, m_); return 1; } } while (0)
enum { N = 256, SW = 1920, SH = 1080, DW = 224, DH = 224, C = 3, STEPS = 30 };

static kh_stream_t st;
static const float* srcs[N];
static float* dsts[N];
static float *big_src, *big_dst;

static int eager(void) { for (int k = 0; k < N; ++k) CK(kh_resize_f32(st, srcs[k], dsts[k], SW, SH, DW, DH, C, KH_INTERP_BILINEAR, 1, 0, 0)); return 0; }
static int list(void) { CK(kh_resize_f32_list(st, srcs, dsts, N, SW, SH, DW, DH, C, KH_INTERP_BILINEAR, KH_MAP_HALF_PIXEL)); return 0; }
static int strided(void) { CK(kh_resize_f32(st, big_src, big_dst, SW, SH, DW, DH, C, KH_INTERP_BILINEAR, N, (int64_t)SW * SH * C, (int64_t)DW * DH * C)); return 0; }
static kh_graph_t graph;
static int replay(void) { CK(kh_graph_launch(graph, st)); return 0; }

static int timed(const char* name, int (*fn)(void)) {
    kh_event_t e0, e1;
    float ms = 0.0f, best = 1e9f, sum = 0.0f;
    CK(kh_event_create(&e0, 1)); CK(kh_event_create(&e1, 1));
    for (int w = 0; w < 5; ++w) if (fn()) return 1;
    CK(kh_stream_synchronize(st));
    for (int r = 0; r < STEPS; ++r) {
        CK(kh_event_record(e0, st));
        if (fn()) return 1;
        CK(kh_event_record(e1, st));
        CK(kh_event_synchronize(e1));
        CK(kh_event_elapsed_ms(e0, e1, &ms));
        sum += ms; if (ms < best) best = ms;
    }
    printf("%-8s mean %.4f ms  min %.4f ms per %d images (%.2f us per image)\n", name, sum / STEPS, best, N, 1e3f * sum / STEPS / N);
    CK(kh_event_destroy(e0)); CK(kh_event_destroy(e1));
    return 0;
}

int main(void) {
    CK(kh_set_device(0));
    CK(kh_stream_create(&st));
    const size_t sb = (size_t)SW * SH * C * 4, db = (size_t)DW * DH * C * 4;
    void* spacer;
    for (int k = 0; k < N; ++k) {   /* separately allocated operands, a spacer of varying size between them */
        void *s, *d;
        if ((5 * k) % 3) CK(kh_malloc_async(&spacer, (size_t)((5 * k) % 3) << 21, 0, st));
        CK(kh_malloc_async(&s, sb, 1, st)); CK(kh_malloc_async(&d, db, 0, st));
        srcs[k] = (const float*)s; dsts[k] = (float*)d;
    }
    CK(kh_malloc_async((void**)&big_src, sb * N, 1, st)); CK(kh_malloc_async((void**)&big_dst, db * N, 0, st));
    CK(kh_stream_synchronize(st));
    CK(kh_graph_capture_begin(st));
    if (eager()) return 1;
    CK(kh_graph_capture_end(st, &graph));
    printf("# configs[1] through the C ABI from plain C, %d timed steps each\n", STEPS);
    for (int round = 0; round < 2; ++round)
        if (timed("strided", strided) || timed("list", list) || timed("graph", replay) || timed("eager", eager)) return 1;
    return 0;
}
