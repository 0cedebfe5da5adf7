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
/* A no-op `void (*deleter)(DLManagedTensor*)`: hosts swap it into tensors still exported through DLPack when their own
 * deleter callback is about to become uncallable (interpreter shutdown).  T/dlpack.rs:72-170 relies on Rust's Box for this. */
KH_API void kh_dlpack_noop_deleter(void* managed_tensor);
/* (The library's two test hooks — kh_debug_fast_quot, kh_debug_set_option — are declared in kornia_hip_testing.h: they are not part
 * of the boundary a host binds.)                                                                          */

/* ------------------------------------------------------------------------------------------ */
/* Device runtime: replaces cudarc's CudaContext/CudaStream/CudaEvent use in T/cuda.rs and
 * P/cuda/dispatch.rs:50-82 (DeviceExec::for_streams — same-device check + event fence).       */

/* Number of HIP runtime images (libamdhip64*) mapped into this process; their paths, newline-separated, go to `buf`
 * (may be NULL).  The reference never faces this (cudarc dlopens the one driver library, T/cuda.rs:170-212); on
 * ROCm a Python wheel may bundle its own runtime, and two runtimes in one process do not order copies or stream
 * waits against each other.  A result above 1 means device work through this library is unsafe — make every HIP
 * user of the process resolve to one image (INTEGRATION.md "One HIP runtime per process").                        */
KH_API int32_t kh_hip_runtime_images(char* buf, size_t cap);
/* HIP version the library was compiled against / the bound runtime reports (major * 10^7 + minor * 10^5 + patch; 0 = unknown):
 * a host that shares another library's bundled runtime (torch wheels) can see a version skew before it bites.              */
KH_API int32_t kh_hip_versions(int32_t* build_version, int32_t* runtime_version);

KH_API int32_t kh_device_count(int32_t* count);
KH_API int32_t kh_set_device(int32_t device);
KH_API int32_t kh_get_device(int32_t* device);
/* name: >= 256 bytes; cu_count / total_mem_bytes may be NULL.                                  */
KH_API int32_t kh_device_info(int32_t device, char* name, size_t name_cap, int32_t* cu_count,
                              uint64_t* total_mem_bytes);

/* free / total bytes of the current device (cuMemGetInfo; kornia_rs.cuda.mem_get_info).            */
KH_API int32_t kh_mem_get_info(uint64_t* free_bytes, uint64_t* total_bytes);

KH_API int32_t kh_stream_create(kh_stream_t* out);            /* non-blocking stream on the current device */
KH_API int32_t kh_stream_destroy(kh_stream_t stream);
KH_API int32_t kh_stream_synchronize(kh_stream_t stream);     /* T/cuda.rs:1258 to_host sync */
KH_API int32_t kh_stream_wait_event(kh_stream_t stream, kh_event_t event);

/* Stream capture -> executable graph: replaces kornia_rs.cuda.Graph.{capture, replay}
 * (PY/cuda_ext/mod.rs:1684-1790: cudarc begin_capture(THREAD_LOCAL) / end_capture / launch).  Everything
 * enqueued on `stream` between begin and end is recorded instead of run; kh_graph_launch replays it.  The
 * recorded work must be allocation-free (preallocated outputs), as the reference requires; an empty
 * capture is an error.  kh_graph_capture_end always ends the capture, also when it fails.           */
typedef struct kh_graph_s* kh_graph_t;
KH_API int32_t kh_graph_capture_begin(kh_stream_t stream);
KH_API int32_t kh_graph_capture_end(kh_stream_t stream, kh_graph_t* out);
KH_API int32_t kh_graph_launch(kh_graph_t graph, kh_stream_t stream);
KH_API int32_t kh_graph_destroy(kh_graph_t graph);

/* Caller scratch for the operators that need an intermediate (kh_resize_fast_u8 / kh_resize_normalize_to_chw_u8_f32 in their
 * separable modes, kh_gaussian_blur_u8 / kh_box_blur_u8 beyond 15 taps, kh_warp_affine_u8, kh_warp_perspective_u8,
 * kh_resize_f32 Lanczos): the reference's low-level launchers take it from the caller (P/cuda/filter.rs:361) while its adapters
 * allocate per call (P/filter/cuda.rs:119).  Here an operator uses the workspace registered for its stream when it is large
 * enough — then it allocates nothing and may be captured — and the stream-ordered pool otherwise (refused under capture,
 * with the byte count in the message).  One workspace serves ONE host thread's calls on that stream.  kh_last_workspace_bytes:
 * scratch the last compute call on this thread asked for (0 = none) — run once eagerly, read it, register, capture.
 * Registrations are filed under (current device, stream): handle 0 is the default stream of every device, so select the stream's
 * device before registering, as before any launch; a buffer that lives on another device is KH_ERR_INVALID_ARG.  kh_stream_destroy
 * drops the stream's registration; the caller must unregister (pointer and size 0) before freeing a registered buffer.
 * kh_stream_workspace_bytes: the size registered for `stream` on the current device (0 = none).                                   */
KH_API int32_t kh_stream_set_workspace(kh_stream_t stream, void* device_ptr, size_t bytes);
KH_API int32_t kh_last_workspace_bytes(size_t* bytes);
KH_API int32_t kh_stream_workspace_bytes(kh_stream_t stream, size_t* bytes);

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

/* The same for frames that are NOT equally spaced — the reference's own batch signature, `Preprocessor::run_raw_batch(frames:
 * &[&CudaSlice<u8>], ..)` (P/preprocess.rs:1234-1282), which it walks with one launch per frame (:1277-1280).  `frames`: HOST
 * array of p->nframes device pointers, read during the call (it may be reused on return); p->src_frame_stride is ignored.  The
 * frame bases travel in the kernel arguments, 256 per launch — nothing is allocated or uploaded, and the call can be captured
 * into a graph.  dst is one [nframes, 3, dst_h, dst_w] tensor, as in the reference (:1243-1256).                                 */
KH_API int32_t kh_preprocess_to_chw_list(kh_stream_t stream, const uint8_t* const* frames, void* dst,
                                         const kh_preprocess_params* p);

/* Name of the kernel variant kh_preprocess_to_chw would launch for `p` (for profiles/benches):
 * "generic", "generic_bilinear_on_grid" (bilinear whose source coordinates all fall on whole pixels: one tap per
 * pixel, same bits) or "nv12_identity".  Returns NULL and sets the error on invalid params.                       */
KH_API const char* kh_preprocess_variant(const kh_preprocess_params* p);

/* ------------------------------------------------------------------------------------------ */
/* Colour conversions: one entry per reference launcher of P/cuda/color/{gray,swizzle,yuv,
 * hsv_hls,misc,video}.rs (adapters P/color/cuda_dispatch.rs:32-47).  Interleaved HWC pixels,
 * `npixels` = rows*cols; src and dst must not alias.                                           */

/* gray: u8 Q14 `(4899R+9617G+1868B+8192)>>14`, f32 `0.299r+0.587g+0.114b`
 * (P/cuda/color/gray.rs:24,56,82,97; CPU P/color/gray/kernels.rs:229-244,405-412)             */
KH_API int32_t kh_gray_from_rgb_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int64_t npixels);
KH_API int32_t kh_gray_from_rgb_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels);
KH_API int32_t kh_rgb_from_gray_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int64_t npixels);
KH_API int32_t kh_rgb_from_gray_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels);

/* channel swizzles (P/cuda/color/swizzle.rs:20-175; CPU P/color/rgb/mod.rs:128-316).
 * swap_rb: 0 = rgba_from_rgb / rgb_from_rgba, 1 = bgra_from_rgb / rgb_from_bgra.  Alpha is 255 /
 * 1.0.  `background` is a HOST pointer to 3 bytes or NULL (NULL = drop alpha; else
 * `round(c*a/255 + bg*(1-a/255))`, P/color/rgb/mod.rs:311-316).                                */
KH_API int32_t kh_bgr_from_rgb_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int64_t npixels);
KH_API int32_t kh_bgr_from_rgb_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels);
KH_API int32_t kh_rgba_from_rgb_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int64_t npixels, int32_t swap_rb);
KH_API int32_t kh_rgba_from_rgb_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels, int32_t swap_rb);
KH_API int32_t kh_rgb_from_rgba_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int64_t npixels,
                                   int32_t swap_rb, const uint8_t* background);

/* full-range YCbCr / YUV, "Family A" (P/cuda/color/yuv.rs:110,189,236; CPU
 * P/color/yuv/kernels.rs:23-126,541-690).  order: KH_YCC_YCRCB stores [Y,Cr,Cb] (OpenCV YCrCb),
 * KH_YCC_YUV stores [Y,U,V] with the cv2 RGB2YUV constants.                                    */
enum { KH_YCC_YCRCB = 0, KH_YCC_YUV = 1 };
KH_API int32_t kh_ycc_from_rgb_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int64_t npixels, int32_t order);
KH_API int32_t kh_rgb_from_ycc_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int64_t npixels, int32_t order);
KH_API int32_t kh_ycc_from_rgb_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels, int32_t order);
KH_API int32_t kh_rgb_from_ycc_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels, int32_t order);

/* HSV / HLS, f32 in the 0..255 domain (P/cuda/color/hsv_hls.rs:49-266; CPU
 * P/color/hsv/kernels.rs:150-177,310-334, P/color/hls/kernels.rs:159-187,357-391)             */
KH_API int32_t kh_hsv_from_rgb_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels);
KH_API int32_t kh_rgb_from_hsv_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels);
KH_API int32_t kh_hls_from_rgb_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels);
KH_API int32_t kh_rgb_from_hls_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels);

/* sepia (Q8 u8 / f32 matrix) and 256-entry colour LUT (P/cuda/color/misc.rs:32,64,109; CPU
 * P/color/sepia.rs:17-104, P/color/colormap.rs:115-122).  `lut_device`: r[256] g[256] b[256]
 * in device memory.                                                                            */
KH_API int32_t kh_sepia_from_rgb_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int64_t npixels);
KH_API int32_t kh_sepia_from_rgb_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels);
KH_API int32_t kh_apply_colormap_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int64_t npixels,
                                    const uint8_t* lut_device);

/* video formats, BT.601 limited range: Q20 decode, Q8 encode (P/cuda/color/video.rs:67-240; CPU
 * P/color/yuv/kernels.rs:696-1216 and 1223-1573).  Buffers are tightly packed.
 * planar420 layout: 0 NV12, 1 NV21, 2 I420, 3 YV12;  packed422 layout: 0 YUYV, 1 UYVY, 2 YVYU. */
KH_API int32_t kh_rgb_from_planar420_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t width,
                                        int32_t height, int32_t layout);
KH_API int32_t kh_rgb_from_packed422_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t width,
                                        int32_t height, int32_t layout);
KH_API int32_t kh_nv12_from_rgb_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t width, int32_t height);
KH_API int32_t kh_yuyv_from_rgb_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t width, int32_t height);

/* ------------------------------------------------------------------------------------------ */
/* f32 geometric resampling.  One entry per reference launcher family of
 * P/cuda/{resize,warp_affine,warp_perspective,remap}.rs (adapters P/resize/cuda.rs:34,
 * P/warp/cuda.rs:29-130, P/interpolation/remap.rs:384); results are bit-identical to the
 * reference CPU ops `resize`, `warp_affine`, `warp_perspective`, `remap`
 * (P/resize/mod.rs:114, P/warp/affine.rs:123, P/warp/perspective.rs:115, P/interpolation/remap.rs:43).
 * Images are HWC f32, channels in {1, 3, 4} (the reference device path has C == 3 only).
 * `batch` same-sized images `src_stride` / `dst_stride` ELEMENTS apart go out as one launch.
 * Matrices are the FORWARD (src -> dst) transform in host memory, inverted on the host like the
 * reference adapters do (P/warp/cuda.rs:65).                                                  */
enum { KH_INTERP_NEAREST = 0, KH_INTERP_BILINEAR = 1, KH_INTERP_BICUBIC = 2, KH_INTERP_LANCZOS = 3 };

KH_API int32_t kh_resize_f32(kh_stream_t stream, const float* src, float* dst, int32_t src_w, int32_t src_h,
                             int32_t dst_w, int32_t dst_h, int32_t channels, int32_t mode, int32_t batch,
                             int64_t src_stride, int64_t dst_stride);
/* The reference's resize launchers take a PixelMapping (P/cuda/resize.rs:433-473): HalfPixel — the public
 * `resize` and the default everywhere — or AlignCorners (`src = dst * (src_len-1)/(dst_len-1)`, a 1-wide
 * destination axis pins to 0).  kh_resize_mapped_f32 == launch_resize_{bilinear_downscale,nearest_downscale,
 * bicubic,lanczos}_cuda(.., mapping) (:490-930); kh_resize_f32 is the HalfPixel case.
 * kh_resize_bilinear_normalize_f32 == launch_resize_bilinear_normalize_cuda (:580-650, kernel :184-236):
 * 3-channel bilinear resize fused with `(px - mean[c]) * (1 / std[c])`, HWC f32 out; mean / std are HOST
 * pointers to 3 floats; a zero std is an error.                                                      */
enum { KH_MAP_HALF_PIXEL = 0, KH_MAP_ALIGN_CORNERS = 1 };
KH_API int32_t kh_pixel_mapping_coeffs(int32_t mapping, int32_t src_len, int32_t dst_len, float out_a_b[2]);
KH_API int32_t kh_resize_mapped_f32(kh_stream_t stream, const float* src, float* dst, int32_t src_w, int32_t src_h,
                                    int32_t dst_w, int32_t dst_h, int32_t channels, int32_t mode, int32_t mapping,
                                    int32_t batch, int64_t src_stride, int64_t dst_stride);
KH_API int32_t kh_resize_bilinear_normalize_f32(kh_stream_t stream, const float* src, float* dst, int32_t src_w,
                                                int32_t src_h, int32_t dst_w, int32_t dst_h, const float* mean,
                                                const float* std_dev, int32_t mapping, int32_t batch,
                                                int64_t src_stride, int64_t dst_stride);
KH_API int32_t kh_warp_affine_f32(kh_stream_t stream, const float* src, float* dst, int32_t src_w, int32_t src_h,
                                  int32_t dst_w, int32_t dst_h, int32_t channels, const float* m2x3, int32_t mode,
                                  int32_t batch, int64_t src_stride, int64_t dst_stride);
/* KH_ERR_SINGULAR if the homography cannot be inverted (checked before any launch).           */
KH_API int32_t kh_warp_perspective_f32(kh_stream_t stream, const float* src, float* dst, int32_t src_w, int32_t src_h,
                                       int32_t dst_w, int32_t dst_h, int32_t channels, const float* m3x3, int32_t mode,
                                       int32_t batch, int64_t src_stride, int64_t dst_stride);
/* map_x / map_y: dst_h x dst_w f32 in device memory, shared by every image of the batch;
 * out-of-range or NaN coordinates write 0 (P/cuda/remap.rs:80-85).                            */
KH_API int32_t kh_remap_f32(kh_stream_t stream, const float* src, const float* map_x, const float* map_y, float* dst,
                            int32_t src_w, int32_t src_h, int32_t dst_w, int32_t dst_h, int32_t channels, int32_t mode,
                            int32_t batch, int64_t src_stride, int64_t dst_stride);
/* `_list` forms: n same-sized images that are NOT equally spaced — separately allocated `Image`s, which is what the reference's
 * per-image operators are handed (`resize(&Image, &mut Image, ..)` P/resize/mod.rs:114-132, `warp_affine` P/warp/affine.rs:123,
 * `warp_perspective` P/warp/perspective.rs:115, `remap` P/interpolation/remap.rs:43; a host that loops them pays one launch per
 * image and amortises that with a captured graph, kornia-py/src/cuda_ext/mod.rs:1684-1790).  `srcs` / `dsts`: HOST arrays of n
 * device pointers, read during the call; the (source, destination) bases travel in the kernel arguments, 128 images per launch;
 * nothing is allocated or uploaded.  Results are those of n single-image calls, bit for bit.                                    */
KH_API int32_t kh_resize_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t src_w,
                                  int32_t src_h, int32_t dst_w, int32_t dst_h, int32_t channels, int32_t mode, int32_t mapping);
KH_API int32_t kh_resize_bilinear_normalize_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n,
                                                     int32_t src_w, int32_t src_h, int32_t dst_w, int32_t dst_h, const float* mean,
                                                     const float* std_dev, int32_t mapping);
KH_API int32_t kh_warp_affine_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t src_w,
                                       int32_t src_h, int32_t dst_w, int32_t dst_h, int32_t channels, const float* m2x3,
                                       int32_t mode);
KH_API int32_t kh_warp_perspective_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n,
                                            int32_t src_w, int32_t src_h, int32_t dst_w, int32_t dst_h, int32_t channels,
                                            const float* m3x3, int32_t mode);
KH_API int32_t kh_remap_f32_list(kh_stream_t stream, const float* const* srcs, const float* map_x, const float* map_y,
                                 float* const* dsts, int32_t n, int32_t src_w, int32_t src_h, int32_t dst_w, int32_t dst_h,
                                 int32_t channels, int32_t mode);
/* Brown-Conrady undistortion maps written directly in device memory (the reference builds them
 * on the host: P/calibration/distortion.rs:135-152).  intrinsic = {fx, fy, cx, cy},
 * distortion = {k1, k2, k3, k4, k5, k6, p1, p2}, host pointers, all-f64 arithmetic.          */
KH_API int32_t kh_correction_map_polynomial_f32(kh_stream_t stream, float* map_x, float* map_y, int32_t width,
                                                int32_t height, const double* intrinsic, const double* distortion);
/* host helpers (no device work): P/warp/affine.rs:18-38, :70-79; P/warp/perspective.rs:41-60  */
KH_API void kh_invert_affine_transform(const float m2x3[6], float out[6]);
KH_API void kh_get_rotation_matrix2d(float center_x, float center_y, float angle_deg, float scale, float out[6]);
KH_API int32_t kh_invert_homography(const float m3x3[9], float out[9]);

/* ------------------------------------------------------------------------------------------ */
/* Separable f32 filters — ONE fused LDS-tiled kernel per call (the reference launches H and V
 * passes through a scratch image, P/cuda/filter.rs:361-385; sobel/scharr five launches,
 * P/filter/cuda.rs:185-237).  Semantics of P/filter/separable_filter.rs:87-164 and
 * P/filter/ops.rs:39-247: zero border (out-of-image taps skipped), f32 intermediate,
 * ascending-tap accumulation.  Any channel count; kernels up to 63 taps; src != dst.         */
enum { KH_GRAD_SOBEL = 0, KH_GRAD_SCHARR = 1 };

KH_API int32_t kh_separable_filter_f32(kh_stream_t stream, const float* src, float* dst, int32_t cols, int32_t rows,
                                       int32_t channels, const float* kernel_x, int32_t nx, const float* kernel_y,
                                       int32_t ny, int32_t batch, int64_t src_stride, int64_t dst_stride);
/* kernel size 0 = derive from sigma, sigma 0 = derive from kernel size (SciPy conventions,
 * P/filter/ops.rs:122-155); taps built on the host with expf (P/filter/kernels.rs:25-43).     */
KH_API int32_t kh_gaussian_blur_f32(kh_stream_t stream, const float* src, float* dst, int32_t cols, int32_t rows,
                                    int32_t channels, int32_t ksize_x, int32_t ksize_y, float sigma_x, float sigma_y,
                                    int32_t batch, int64_t src_stride, int64_t dst_stride);
KH_API int32_t kh_box_blur_f32(kh_stream_t stream, const float* src, float* dst, int32_t cols, int32_t rows,
                               int32_t channels, int32_t ksize_x, int32_t ksize_y, int32_t batch, int64_t src_stride,
                               int64_t dst_stride);
/* sobel (ksize 3|5) / scharr (ksize 3): sqrt(gx^2 + gy^2), P/filter/ops.rs:174-247            */
KH_API int32_t kh_gradient_magnitude_f32(kh_stream_t stream, const float* src, float* dst, int32_t cols, int32_t rows,
                                         int32_t channels, int32_t kind, int32_t ksize, int32_t batch,
                                         int64_t src_stride, int64_t dst_stride);
/* `_list` forms (see kh_resize_f32_list): n separately allocated images per call, 128 per launch; the reference's filters take one
 * `&Image` per call (P/filter/ops.rs:39,116,174,214; P/filter/separable_filter.rs:87).                                        */
KH_API int32_t kh_separable_filter_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n,
                                            int32_t cols, int32_t rows, int32_t channels, const float* kernel_x, int32_t nx,
                                            const float* kernel_y, int32_t ny);
KH_API int32_t kh_gaussian_blur_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t cols,
                                         int32_t rows, int32_t channels, int32_t ksize_x, int32_t ksize_y, float sigma_x,
                                         float sigma_y);
KH_API int32_t kh_box_blur_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n, int32_t cols,
                                    int32_t rows, int32_t channels, int32_t ksize_x, int32_t ksize_y);
KH_API int32_t kh_gradient_magnitude_f32_list(kh_stream_t stream, const float* const* srcs, float* const* dsts, int32_t n,
                                              int32_t cols, int32_t rows, int32_t channels, int32_t kind, int32_t ksize);
/* host tap builders (P/filter/kernels.rs:10-43) and parameter resolution                      */
KH_API int32_t kh_box_blur_kernel_1d(int32_t n, float* out);
KH_API int32_t kh_gaussian_kernel_1d(int32_t n, float sigma, float* out);
KH_API int32_t kh_gaussian_resolve(int32_t ksize_xy[2], float sigma_xy[2]);

/* ------------------------------------------------------------------------------------------ */
/* The rest of the reference's filter module.  Strides in ELEMENTS between batch images.
 *
 * spatial_gradient_float / scharr_spatial_gradient_float (P/filter/ops.rs:287-590; the _parallel
 * variants share the arithmetic): normalised 3x3 Sobel / Scharr cross-correlation
 * (P/filter/kernels.rs:107-140), replicate border, nine products added in row-major tap order.
 * kind = KH_GRAD_SOBEL | KH_GRAD_SCHARR; src, dx, dy distinct images of the same shape.        */
KH_API int32_t kh_spatial_gradient_f32(kh_stream_t stream, const float* src, float* dx, float* dy, int32_t cols,
                                       int32_t rows, int32_t channels, int32_t kind, int32_t batch, int64_t src_stride,
                                       int64_t dst_stride);
/* box_blur_fast (P/filter/ops.rs:252-285): three rounds of a running-sum box per axis with the
 * sizes of box_blur_fast_kernels_1d (P/filter/kernels.rs:151-170) used as half widths, exactly as
 * the reference does.  `scratch` = batch images of cols*rows*channels floats (the transposed
 * intermediate, caller-provided like the reference's low-level filter launcher's scratch,
 * P/cuda/filter.rs:361).  A half width that does not fit the image is KH_ERR_INVALID_ARG (the
 * reference indexes out of bounds there).  kh_fast_horizontal_filter_f32 is one pass
 * (fast_horizontal_filter, P/filter/separable_filter.rs:202-257): dst is rows-wide, cols-tall.   */
KH_API int32_t kh_box_blur_fast_kernels_1d(float sigma, int32_t kernels, int32_t* out);
KH_API int32_t kh_fast_horizontal_filter_f32(kh_stream_t stream, const float* src, float* dst_transposed, int32_t cols,
                                             int32_t rows, int32_t channels, int32_t half, int32_t batch,
                                             int64_t src_stride, int64_t dst_stride);
KH_API int32_t kh_box_blur_fast_f32(kh_stream_t stream, const float* src, float* dst, float* scratch, int32_t cols,
                                    int32_t rows, int32_t channels, float sigma_x, float sigma_y, int32_t batch,
                                    int64_t src_stride, int64_t dst_stride);
/* median_blur == launch_median_u8 (P/filter/median.rs:174-250, P/cuda/median.rs:95-138): exact
 * median of the replicate-bordered ksize x ksize window per channel; ksize 3 | 5 (else
 * KH_ERR_INVALID_ARG, the reference's InvalidKernelLength), 1..4 channels.                        */
KH_API int32_t kh_median_blur_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t cols, int32_t rows,
                                 int32_t channels, int32_t ksize, int32_t batch, int64_t src_stride,
                                 int64_t dst_stride);
/* bilateral_filter == launch_bilateral_u8 (P/filter/bilateral.rs:172-300, P/cuda/bilateral.rs:33-135):
 * single-channel u8, byte-for-byte cv2.bilateralFilter semantics — circular window of radius d/2
 * (or round(1.5 sigma_space) for d <= 0), reflect-101 border, cv2's colour table (its SIMD exp
 * polynomial + scalar-expf tail) and its position-dependent tap order; sigma <= 1e-6 copies the
 * source through.  Tables are built on the host and cached on the device per (d, sigmas): the
 * FIRST call with a new parameter set uploads them with a blocking copy, so warm it up once
 * before capturing the call into a graph (kh_graph_capture_begin).
 * A window radius above 512 (d > 1025 or sigma_space > ~341 with d <= 0) is KH_ERR_TOO_LARGE.
 * kh_bilateral_tables returns them (build_tables, bilateral.rs:110-170): *ntaps always; the
 * arrays (color_weight: 256 entries) only when capacity >= *ntaps.                                */
KH_API int32_t kh_bilateral_filter_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t cols, int32_t rows,
                                      int32_t d, double sigma_color, double sigma_space, int32_t batch,
                                      int64_t src_stride, int64_t dst_stride);
KH_API int32_t kh_bilateral_tables(int32_t d, double sigma_color, double sigma_space, int32_t capacity, int32_t* radius,
                                   int32_t* ntaps, int32_t* tap_dy, int32_t* tap_dx, float* space_weight,
                                   float* color_weight, int32_t* simd_order);

/* CIE colour spaces — replaces the 16 NVRTC kernels of P/cuda/color/cie.rs (adapters
 * P/color/cuda_dispatch.rs) == linear_rgb_from_rgb, rgb_from_linear_rgb, xyz_from_rgb, rgb_from_xyz,
 * lab_from_rgb, rgb_from_lab, luv_from_rgb, rgb_from_luv (P/color/cie/mod.rs:58-130): f32 RGB in [0, 1],
 * D65, OpenCV coefficients; XYZ is the bare 3x3 matrix (no gamma), Lab / Luv linearise first.  The
 * matrix conversions are bit-identical to the reference's scalar path; the transfer / cube-root stages
 * are held to its own tolerances against the f64 formulas (powf / cbrtf differ in the last bits between
 * math libraries, as they do between the reference's NEON and scalar paths).                    */
enum { KH_CIE_LINEAR_RGB_FROM_RGB = 0, KH_CIE_RGB_FROM_LINEAR_RGB = 1, KH_CIE_XYZ_FROM_RGB = 2, KH_CIE_RGB_FROM_XYZ = 3,
       KH_CIE_LAB_FROM_RGB = 4, KH_CIE_RGB_FROM_LAB = 5, KH_CIE_LUV_FROM_RGB = 6, KH_CIE_RGB_FROM_LUV = 7 };
KH_API int32_t kh_cie_convert_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels, int32_t conversion);

/* f64 colour conversions — replace the 18 f64 launchers of P/color/cuda_dispatch.rs:48-61,111-135
 * (gray::launch_{gray_from_rgb,rgb_from_gray}_f64, hsv_hls::launch_*_f64, cie::launch_*_f64,
 * yuv::launch_{ycc_from_rgb,rgb_from_ycc}_f64) == the f64 arms of gray_from_rgb (P/color/gray/mod.rs:41),
 * hsv/hls (P/color/hsv/mod.rs:64-113, P/color/hls/mod.rs:64-130; [0,255] domain), YCbCr / YUV
 * (P/color/yuv/mod.rs:95-145) and the CIE `*_scalar64` formulas (P/color/cie/kernels.rs:64-215), CPU
 * arithmetic operation for operation.  Interleaved f64 pixels, 3 -> 3 channels except GRAY_FROM_RGB
 * (3 -> 1) and RGB_FROM_GRAY (1 -> 3); codes 0..7 equal KH_CIE_*.                                  */
enum { KH_F64_GRAY_FROM_RGB = 8, KH_F64_RGB_FROM_GRAY = 9, KH_F64_HSV_FROM_RGB = 10, KH_F64_RGB_FROM_HSV = 11,
       KH_F64_HLS_FROM_RGB = 12, KH_F64_RGB_FROM_HLS = 13, KH_F64_YCBCR_FROM_RGB = 14, KH_F64_RGB_FROM_YCBCR = 15,
       KH_F64_YUV_FROM_RGB = 16, KH_F64_RGB_FROM_YUV = 17 };
KH_API int32_t kh_color_convert_f64(kh_stream_t stream, const double* src, double* dst, int64_t npixels, int32_t conversion);

/* Bayer mosaic -> RGB8, bilinear, cv2-compatible — replaces launch_rgb_from_bayer_u8 (P/cuda/color/bayer.rs) ==
 * rgb_from_bayer (P/color/bayer/mod.rs:37-70, kernels.rs:30-200): rounded integer averages over
 * replicate-clamped neighbours; the 1-pixel frame takes its interior neighbour's result.            */
enum { KH_BAYER_RGGB = 0, KH_BAYER_BGGR = 1, KH_BAYER_GRBG = 2, KH_BAYER_GBRG = 3 };
KH_API int32_t kh_rgb_from_bayer_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t width, int32_t height,
                                    int32_t pattern);

/* YUYV -> RGB8 with a selectable matrix — replaces the yuyv_to_rgb_{bt601_full,bt709_full,bt601_limited}_u8
 * launchers (P/cuda/color/video.rs:128-190) == convert_yuyv_to_rgb_u8 (P/color/yuv/mod.rs:342-410; Q10
 * integer, one (U,V) per pixel pair).  src: width*height*2 bytes `Y0 U Y1 V`; an odd width leaves the
 * last pixel of each row untouched, as the reference does.                                         */
enum { KH_YUV_BT601_FULL = 0, KH_YUV_BT709_FULL = 1, KH_YUV_BT601_LIMITED = 2 };
KH_API int32_t kh_yuyv_to_rgb_mode_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t width, int32_t height,
                                      int32_t mode);

/* ------------------------------------------------------------------------------------------ */
/* u8 fixed-point twins.  Byte-identical to the reference CPU ops they replace the
 * device launchers of; HWC u8, channels in {1, 3, 4}, `batch` images `*_stride` BYTES apart.
 *
 * Blur — replaces launch_gaussian_blur_u8 / launch_box_blur_u8 and the binomial special case
 * (P/cuda/filter.rs:116-250, adapters P/filter/cuda.rs:282) == gaussian_blur_u8 / box_blur_u8
 * (P/filter/ops.rs:639, 59): taps quantised to Q8 with the centre absorbing the rounding error
 * (quantize_kernel_256, :748-760), replicate border, `(acc + 128) >> 8` after EACH pass; a 3x3
 * kernel with both sigmas in [0.6, 1.2] is the [1,2,1]/4 binomial of rounding halving adds
 * (blur_u8_path, :21-27).  One fused launch, no scratch image (kernels wider than 15 taps fall back
 * to the reference's two-pass structure through stream-ordered scratch).  src != dst.            */
KH_API void kh_quantize_kernel_256(const float* kernel, int32_t n, uint8_t* out);
KH_API int32_t kh_gaussian_blur_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t cols, int32_t rows,
                                   int32_t channels, int32_t ksize_x, int32_t ksize_y, float sigma_x, float sigma_y,
                                   int32_t batch, int64_t src_stride, int64_t dst_stride);
/* odd kernel sizes only (P/filter/ops.rs:66-75)                                                 */
KH_API int32_t kh_box_blur_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t cols, int32_t rows,
                              int32_t channels, int32_t ksize_x, int32_t ksize_y, int32_t batch, int64_t src_stride,
                              int64_t dst_stride);
/* Q10 bilinear gathers — replace launch_remap_u8 (P/cuda/remap.rs), launch_warp_affine_u8
 * (P/cuda/warp_affine_u8.rs) and launch_warp_perspective_u8 (P/cuda/warp_perspective_u8.rs) ==
 * remap_u8 (P/interpolation/remap.rs:157; nearest | bilinear only), warp_affine_u8
 * (P/warp/affine.rs:373: per-row valid span, Q16 stepped coordinates) and warp_perspective_u8
 * (P/warp/perspective.rs:179: analytic span on constant-sign rows, direct per-column coordinates).
 * Sampler: P/warp/common.rs:16-165, `(top*fy1 + bot*fy + 2^19) >> 20`, zeros outside.  Matrices
 * are FORWARD (src -> dst), host memory.  These three also take 2-channel images (the reference
 * instantiates them per channel count and tests C = 2, P/warp/cuda.rs:458-494).                  */
KH_API int32_t kh_remap_u8(kh_stream_t stream, const uint8_t* src, const float* map_x, const float* map_y,
                           uint8_t* dst, int32_t src_w, int32_t src_h, int32_t dst_w, int32_t dst_h, int32_t channels,
                           int32_t mode, int32_t batch, int64_t src_stride, int64_t dst_stride);
KH_API int32_t kh_warp_affine_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t src_w, int32_t src_h,
                                 int32_t dst_w, int32_t dst_h, int32_t channels, const float* m2x3, int32_t batch,
                                 int64_t src_stride, int64_t dst_stride);
/* KH_ERR_SINGULAR if the homography cannot be inverted                                          */
KH_API int32_t kh_warp_perspective_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t src_w,
                                      int32_t src_h, int32_t dst_w, int32_t dst_h, int32_t channels, const float* m3x3,
                                      int32_t batch, int64_t src_stride, int64_t dst_stride);

/* u8 resize cascade — replaces resize_fast_u8_cuda (P/resize/cuda.rs:207-330, kernels
 * P/cuda/resize_u8.rs) == resize_fast_u8_aa (P/resize/mod.rs:348): routed by the reference's single
 * selector resize_u8_path (:283-340) — exact-2x RGB bilinear -> box / 75-25 fast paths, nearest (any
 * channel count 1..4), Q14 bilinear (C in {1,3,4}, source >= 2x2 else KH_ERR_INVALID_ARG), bicubic /
 * lanczos -> two-pass Q14 separable (antialias != 0 widens the kernel by the downscale factor, PIL
 * semantics; 0 = fixed 4 / 6 taps, OpenCV semantics).  Strides in BYTES.                        */
KH_API int32_t kh_resize_fast_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t src_w, int32_t src_h,
                                 int32_t dst_w, int32_t dst_h, int32_t channels, int32_t mode, int32_t antialias,
                                 int32_t batch, int64_t src_stride, int64_t dst_stride);
/* Fused resize + normalise + HWC->CHW for RGB8 == resize_normalize_to_tensor_u8_to_f32{,_bilinear,
 * _nearest,_separable} (P/resize/fused.rs:57,147,885,938; CPU-only in the reference — "the CPU timing
 * twin" of the camera preprocess).  out[c] = sample * scale[c] + bias[c] with the pre-combined
 * NormalizeParams (scale = 1/(std*255), bias = -mean/std; HOST pointers to 3 floats).  bilinear
 * dispatches exact-2x downscales to the f32 box average like the reference (:172-174); bicubic /
 * lanczos = Q14 horizontal pass, i32 vertical accumulate, no u8 requantisation.  dst = [3, dst_h,
 * dst_w] f32 planes; src stride in BYTES, dst stride in ELEMENTS.                                */
KH_API int32_t kh_resize_normalize_to_chw_u8_f32(kh_stream_t stream, const uint8_t* src, float* dst, int32_t src_w,
                                                 int32_t src_h, int32_t dst_w, int32_t dst_h, const float* scale,
                                                 const float* bias, int32_t mode, int32_t antialias, int32_t batch,
                                                 int64_t src_stride, int64_t dst_stride);
/* cv2.resize-compatible INTER_NEAREST / INTER_LINEAR == resize_opencv_{u8,f32}
 * (P/resize/opencv_compat.rs:76-250; CPU-only in the reference).  Strides in ELEMENTS.           */
KH_API int32_t kh_resize_opencv_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t src_w, int32_t src_h,
                                   int32_t dst_w, int32_t dst_h, int32_t channels, int32_t mode, int32_t batch,
                                   int64_t src_stride, int64_t dst_stride);
KH_API int32_t kh_resize_opencv_f32(kh_stream_t stream, const float* src, float* dst, int32_t src_w, int32_t src_h,
                                    int32_t dst_w, int32_t dst_h, int32_t channels, int32_t mode, int32_t batch,
                                    int64_t src_stride, int64_t dst_stride);

/* ------------------------------------------------------------------------------------------ */
/* Gaussian pyramid + morphology.  Replace launch_pyrdown / launch_pyrup {f32,u8}
 * (P/cuda/pyramid.rs:84-362) and launch_morphology (P/cuda/morphology.rs) == pyrdown_f32 / pyrup_f32
 * / pyrdown_u8 / pyrup_u8 (P/pyramid.rs:312,210,469,804) and dilate / erode (P/morphology/ops.rs:22,
 * 125).  pyrdown: dst = ceil(src/2) per axis, 5x5 [1 4 6 4 1]^2/256, reflect-101; pyrup: dst = 2*src,
 * [1 6 1]/8 even / [1 1]/2 odd taps (f32: the reference's special border rows and columns; u8:
 * reflect-101 with a u8 intermediate).  One launch, no intermediate image.  HWC, channels in
 * {1,3,4}; strides in ELEMENTS.                                                                  */
KH_API int32_t kh_pyrdown_f32(kh_stream_t stream, const float* src, float* dst, int32_t src_w, int32_t src_h,
                              int32_t channels, int32_t batch, int64_t src_stride, int64_t dst_stride);
KH_API int32_t kh_pyrup_f32(kh_stream_t stream, const float* src, float* dst, int32_t src_w, int32_t src_h,
                            int32_t channels, int32_t batch, int64_t src_stride, int64_t dst_stride);
KH_API int32_t kh_pyrdown_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t src_w, int32_t src_h,
                             int32_t channels, int32_t batch, int64_t src_stride, int64_t dst_stride);
KH_API int32_t kh_pyrup_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t src_w, int32_t src_h,
                           int32_t channels, int32_t batch, int64_t src_stride, int64_t dst_stride);
/* u8 dilate / erode over a structuring element `mask` (HOST, kernel_w * kernel_h bytes, 1 = active,
 * at most 32x32; anchor = (kernel_h/2, kernel_w/2)).  Border = PaddingMode (P/padding.rs:6-31);
 * `constant_value`: HOST pointer to `channels` bytes (KH_BORDER_CONSTANT only).  open / close are
 * erode-then-dilate / dilate-then-erode through a caller-provided temporary, as in the reference
 * (P/morphology/ops.rs:227-275).  src != dst.                                                      */
enum { KH_MORPH_DILATE = 0, KH_MORPH_ERODE = 1 };
enum { KH_BORDER_CONSTANT = 0, KH_BORDER_REPLICATE = 1, KH_BORDER_REFLECT101 = 2, KH_BORDER_REFLECT = 3, KH_BORDER_WRAP = 4 };
enum { KH_MORPH_BOX = 0, KH_MORPH_CROSS = 1, KH_MORPH_ELLIPSE = 2 };
KH_API int32_t kh_morph_kernel(int32_t shape, int32_t width, int32_t height, uint8_t* out_mask);
KH_API int32_t kh_morphology_u8(kh_stream_t stream, const uint8_t* src, uint8_t* dst, int32_t width, int32_t height,
                                int32_t channels, int32_t op, const uint8_t* mask, int32_t kernel_w, int32_t kernel_h,
                                int32_t border, const uint8_t* constant_value, int32_t batch, int64_t src_stride,
                                int64_t dst_stride);

/* ------------------------------------------------------------------------------------------ */
/* Fused per-pixel pipelines — replaces FusedPipeline::{build, build_batched, launch,
 * launch_batched, generated_source} and the v1 stage library (P/cuda/fusion.rs:196-690): a source
 * stage produces an f32 RGB value for each destination pixel, map stages transform it in registers,
 * a sink stage writes it.  No run-time compiler: the stage program travels in the kernel arguments.
 * Stage parameters:  READ_U8RGB_BILINEAR u = {src_w, src_h, dst_w, dst_h} (half-pixel grid, value in
 * [0, 255]);  NORMALIZE f = {scale r,g,b, bias r,g,b};  RGB_TO_GRAY, WRITE_CHW_F32 (3 planes),
 * WRITE_C1_F32 (lane x): none.  `batch` images are read through per-image pointers and written
 * `out_elems_per_image` floats apart (ignored for batch == 1).  Errors follow FusionError: shape
 * problems -> KH_ERR_INVALID_ARG ("invalid pipeline: ..."), too many stages -> KH_ERR_TOO_LARGE,
 * short buffers at launch -> KH_ERR_SLICE_TOO_SMALL (pass 0 to skip a length check).           */
enum { KH_FUSE_READ_U8RGB_BILINEAR = 1, KH_FUSE_NORMALIZE = 16, KH_FUSE_RGB_TO_GRAY = 17, KH_FUSE_WRITE_CHW_F32 = 32,
       KH_FUSE_WRITE_C1_F32 = 33 };
typedef struct kh_fused_stage {
    int32_t kind;
    int32_t u[4];
    float f[6];
} kh_fused_stage;
typedef struct kh_fused_pipeline_s* kh_fused_pipeline_t;
KH_API int32_t kh_fused_pipeline_build(const kh_fused_stage* stages, int32_t nstages, int32_t dst_w, int32_t dst_h,
                                       int32_t batch, int64_t out_elems_per_image, kh_fused_pipeline_t* out);
/* srcs: HOST array of `nsrcs` device pointers (== the built batch)                              */
KH_API int32_t kh_fused_pipeline_launch(kh_fused_pipeline_t pipeline, kh_stream_t stream, const uint8_t* const* srcs,
                                        int32_t nsrcs, int64_t src_bytes_each, float* dst, int64_t dst_elems);
/* the stage program as text (analogue of generated_source); returns its length                 */
KH_API int32_t kh_fused_pipeline_describe(kh_fused_pipeline_t pipeline, char* buf, size_t n);
KH_API void kh_fused_pipeline_destroy(kh_fused_pipeline_t pipeline);

/* ------------------------------------------------------------------------------------------ */
/* normalize / crop / flip (P/normalize.rs:56-420, P/crop.rs:187-240, P/flip.rs:39-360).  The
 * reference has no device twin for normalize; these follow its CPU arithmetic: true division
 * in normalize_mean_std, `(x-min_v)*(max-min)/(max_v-min_v)+min` in normalize_min_max, the
 * scalar `x*scale+offset` in normalize_rgb_u8.                                                 */
/* mean / std: HOST pointers to `channels` floats, channels in 1..4                             */
KH_API int32_t kh_normalize_mean_std_f32(kh_stream_t stream, const float* src, float* dst, int64_t npixels,
                                         int32_t channels, const float* mean, const float* std);
/* RGB8 -> f32 with per-channel scale/offset (HOST pointers to 3 floats)                        */
KH_API int32_t kh_normalize_rgb_u8_f32(kh_stream_t stream, const uint8_t* src, float* dst, int64_t npixels,
                                       const float* scale, const float* offset);
/* min / max over `n` floats, no host round trip: minmax_device <- {min, max} (2 floats, device),
 * scratch_device = 2 x uint32 of device scratch.  NaNs lose every comparison, except a NaN first
 * element which yields (NaN, NaN) — the behaviour of the reference loop (P/normalize.rs:123-146). */
KH_API int32_t kh_find_min_max_f32(kh_stream_t stream, const float* src, int64_t n, float* minmax_device,
                                   uint32_t* scratch_device);
KH_API int32_t kh_normalize_min_max_f32(kh_stream_t stream, const float* src, float* dst, int64_t n, float min,
                                        float max, float* minmax_device, uint32_t* scratch_device);
/* crop a dst_w x dst_h window at (x, y); pixel_bytes = channels * sizeof(T)                    */
KH_API int32_t kh_crop(kh_stream_t stream, const void* src, void* dst, int32_t src_w, int32_t src_h, int32_t dst_w,
                       int32_t dst_h, int32_t x, int32_t y, int32_t pixel_bytes);
/* horizontal != 0: mirror columns; 0: mirror rows                                              */
KH_API int32_t kh_flip(kh_stream_t stream, const void* src, void* dst, int32_t width, int32_t height,
                       int32_t pixel_bytes, int32_t horizontal);

#ifdef __cplusplus
}
#endif
#endif /* KORNIA_HIP_H */
