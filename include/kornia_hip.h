/*
 * kornia_hip.h — C ABI of libkornia_hip.so, the MI355X (gfx950) device backend for the
 * kornia-rs image-processing hot path.
 *
 * This is the drop-in boundary: every entry point replaces one Rust-internal launcher (or
 * allocator call) of the reference's CUDA backend.  The reference has no C ABI of its own — its
 * device seam is `cudarc` + NVRTC inside the crates — so each declaration cites the reference
 * interface it stands in for (paths relative to the reference checkout):
 *
 *   T/ = crates/kornia-tensor/src/     I/ = crates/kornia-image/src/
 *   P/ = crates/kornia-imgproc/src/    PY/ = kornia-py/src/
 *
 * Conventions
 *   - plain C: pointers, fixed-width ints, floats; no C++ / torch types.
 *   - `kh_stream_t` is an opaque `hipStream_t` (NULL = the legacy default stream).  A stream
 *     created by another runtime in the same process (e.g. torch.cuda.Stream.cuda_stream) may
 *     be passed as-is.
 *   - every function returns `int32_t` status: 0 (KH_OK) on success, a negative KH_ERR_* code
 *     otherwise; a human-readable message is kept per thread and read with kh_last_error().
 *   - operators are asynchronous on `stream`, never allocate or free their operands, never
 *     synchronise, and fully overwrite `dst` (it may be uninitialised) — the ownership rules of
 *     the reference launchers (I/cuda.rs:110-121, P/cuda/filter.rs:361).
 *   - there is NO CPU fallback behind any entry point: without a HIP device they fail with
 *     KH_ERR_HIP.
 */
#ifndef KORNIA_HIP_H
#define KORNIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KH_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* status codes (mirror the reference's typed errors: I/error.rs, P/cuda/mod.rs:182-196,
 * P/preprocess.rs:253-340)                                                                     */
enum {
    KH_OK = 0,
    KH_ERR_INVALID_ARG = -1,   /* null pointer, zero/odd dimension, bad enum value              */
    KH_ERR_HIP = -2,           /* a HIP runtime call failed (message carries hipGetErrorString) */
    KH_ERR_UNSUPPORTED = -3,   /* no device kernel for this dtype/channel count — never a fallback
                                  (P/cuda/dispatch.rs:203-211 no_gpu_kernel_err)                */
    KH_ERR_TOO_LARGE = -4,     /* dims exceed 32-bit kernel indexing (P/preprocess.rs:1336-1339) */
    KH_ERR_SINGULAR = -5,      /* non-invertible transform (P/warp/perspective.rs:41-60)        */
    KH_ERR_SLICE_TOO_SMALL = -6 /* buffer shorter than the geometry needs (P/cuda/mod.rs:182)   */
};

typedef struct kh_stream_opaque* kh_stream_t; /* == hipStream_t */
typedef struct kh_event_opaque* kh_event_t;   /* == hipEvent_t  */

/* Copy this thread's last error message into `buf` (NUL-terminated, truncated to `cap`).
 * Returns the full message length.  Replaces `CudaError`'s Display (T/cuda.rs:60-86).          */
KH_API size_t kh_last_error(char* buf, size_t cap);

/* Library version string, e.g. "kornia-hip 0.1.0 (gfx950)".                                    */
KH_API const char* kh_version(void);

/* ------------------------------------------------------------------------------------------ */
/* Device runtime: replaces cudarc's CudaContext/CudaStream/CudaEvent use in T/cuda.rs and
 * P/cuda/dispatch.rs:50-82 (DeviceExec::for_streams — same-device check + event fence).       */

KH_API int32_t kh_device_count(int32_t* count);
KH_API int32_t kh_set_device(int32_t device);
KH_API int32_t kh_get_device(int32_t* device);
/* name: >= 256 bytes; cu_count / total_mem_bytes may be NULL.                                  */
KH_API int32_t kh_device_info(int32_t device, char* name, size_t name_cap, int32_t* cu_count,
                              uint64_t* total_mem_bytes);

KH_API int32_t kh_stream_create(kh_stream_t* out);            /* non-blocking stream on the current device */
KH_API int32_t kh_stream_destroy(kh_stream_t stream);
KH_API int32_t kh_stream_synchronize(kh_stream_t stream);     /* T/cuda.rs:1258 to_host sync */
KH_API int32_t kh_stream_wait_event(kh_stream_t stream, kh_event_t event);

KH_API int32_t kh_event_create(kh_event_t* out, int32_t enable_timing);
KH_API int32_t kh_event_destroy(kh_event_t event);
KH_API int32_t kh_event_record(kh_event_t event, kh_stream_t stream);
KH_API int32_t kh_event_synchronize(kh_event_t event);
KH_API int32_t kh_event_elapsed_ms(kh_event_t start, kh_event_t stop, float* ms);
/* Cross-stream fence used by the residency dispatch when src and dst carry different streams:
 * `consumer` waits for everything queued so far on `producer` (P/cuda/dispatch.rs:56-66).     */
KH_API int32_t kh_stream_fence(kh_stream_t producer, kh_stream_t consumer);

/* ------------------------------------------------------------------------------------------ */
/* HIP allocator: replaces CudaAllocator / PinnedAllocator / CudaUnifiedAllocator
 * (T/cuda.rs:214-298, 355-380, 440-511) and zeros_cuda / uninit_cuda (T/cuda.rs:860,891).     */

/* Stream-ordered device allocation from the device's default mem-pool.  `zeroed != 0` also
 * queues a memset (CudaAllocator::allocate is always zeroed; uninit_cuda skips it).            */
KH_API int32_t kh_malloc_async(void** out, size_t bytes, int32_t zeroed, kh_stream_t stream);
KH_API int32_t kh_free_async(void* ptr, kh_stream_t stream);
/* Keep up to `bytes` cached in the pool instead of returning it to the driver at sync points
 * (T/cuda.rs:238-262 release-threshold tuning).                                                */
KH_API int32_t kh_mempool_set_release_threshold(int32_t device, uint64_t bytes);
/* Page-locked host memory, zero-filled (PinnedAllocator, T/cuda.rs:355-380).                   */
KH_API int32_t kh_host_alloc(void** out, size_t bytes);
KH_API int32_t kh_host_free(void* ptr);
/* Managed (unified) memory attached globally, zero-filled (T/cuda.rs:440-511).                 */
KH_API int32_t kh_malloc_managed(void** out, size_t bytes);
KH_API int32_t kh_free(void* ptr);

KH_API int32_t kh_memcpy_h2d_async(void* dst, const void* src, size_t bytes, kh_stream_t stream);
KH_API int32_t kh_memcpy_d2h_async(void* dst, const void* src, size_t bytes, kh_stream_t stream);
KH_API int32_t kh_memcpy_d2d_async(void* dst, const void* src, size_t bytes, kh_stream_t stream);
KH_API int32_t kh_memset_async(void* dst, int32_t value, size_t bytes, kh_stream_t stream);

/* Residency of an arbitrary pointer — what MemoryResource::domain() answers for owned storage
 * (T/resource.rs:19-60).  domain: 0 = Host (pageable or unknown), 1 = Device, 2 = Unified,
 * 3 = Host pinned.  device = ordinal for 1/2/3, -1 for 0.                                     */
enum { KH_DOMAIN_HOST = 0, KH_DOMAIN_DEVICE = 1, KH_DOMAIN_UNIFIED = 2, KH_DOMAIN_HOST_PINNED = 3 };
KH_API int32_t kh_pointer_domain(const void* ptr, int32_t* domain, int32_t* device);

/* ------------------------------------------------------------------------------------------ */
/* Fused camera preprocess: raw frame -> resized, normalised, channel-planar tensor.
 * Replaces the six NVRTC entries `resize_normalize_to_chw_{bilinear,nearest,lanczos}[_f16]`
 * and their launcher `Preprocessor::launch_view` (P/preprocess.rs:430-647, 1324-1375).        */

enum { /* P/preprocess.rs:157-167 SourceFormat::fmt_code */
    KH_FMT_RGB = 0,  /* interleaved R,G,B[,A]; bpp 3 or 4 (alpha skipped) */
    KH_FMT_BGR = 1,  /* interleaved B,G,R[,A]; bpp 3 or 4                 */
    KH_FMT_GRAY = 2, /* 1 byte/px broadcast to 3 channels                 */
    KH_FMT_NV12 = 3, /* w*h luma plane, then interleaved half-res UV rows  */
    KH_FMT_YUYV = 4  /* packed 4:2:2  Y0 U Y1 V                            */
};
enum { KH_SAMPLE_NEAREST = 0, KH_SAMPLE_BILINEAR = 1, KH_SAMPLE_LANCZOS = 2 };
enum { KH_OUT_F32 = 0, KH_OUT_F16 = 1 };

typedef struct kh_preprocess_params {
    /* geometry: src = (dst - pad) / scale per axis (P/preprocess.rs:350-369 Affine) */
    float scale_x, scale_y, pad_x, pad_y;
    int32_t src_w, src_h;
    int32_t src_pitch;  /* bytes per primary-plane row (NV12 ignores it: pitch == src_w) */
    int32_t src_bpp;    /* interleaved bytes/px: 3|4 (RGB/BGR), 1 (gray, NV12), 2 (YUYV)  */
    int32_t fmt;        /* KH_FMT_*    */
    int32_t dst_w, dst_h;
    float mean[3];      /* [0,1] domain */
    float inv_std[3];
    float pad_value;    /* 0..255 scale, used where the sample falls outside the source   */
    int32_t sampling;   /* KH_SAMPLE_* */
    int32_t out_dtype;  /* KH_OUT_*    */
    /* batch: frame k reads src + k*src_frame_stride (bytes) and writes
     * dst + k*dst_frame_stride (elements).  One launch covers the whole batch — the
     * reference loops one launch per frame (P/preprocess.rs:1277-1280).                     */
    int32_t nframes;
    int32_t flags;      /* KH_PRE_* bit set */
    int64_t src_frame_stride;
    int64_t dst_frame_stride;
} kh_preprocess_params;

/* flags: force the one-thread-per-pixel kernel even where a specialised variant applies (used
 * by the parity tests to prove the variants agree bit for bit).                              */
enum { KH_PRE_FORCE_GENERIC = 1 };

/* dst: nframes x [3, dst_h, dst_w] of f32 (or IEEE binary16 bits when out_dtype == KH_OUT_F16). */
KH_API int32_t kh_preprocess_to_chw(kh_stream_t stream, const uint8_t* src, void* dst,
                                    const kh_preprocess_params* p);

/* Name of the kernel variant kh_preprocess_to_chw would launch for `p` (for profiles/benches):
 * "generic" or "nv12_identity".  Returns NULL and sets the error on invalid params.           */
KH_API const char* kh_preprocess_variant(const kh_preprocess_params* p);

#ifdef __cplusplus
}
#endif
#endif /* KORNIA_HIP_H */
