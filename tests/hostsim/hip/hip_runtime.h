// TEST INFRASTRUCTURE — a host stand-in for <hip/hip_runtime.h> so that the product's .hip sources (kernels AND host
// entries, unmodified) compile with clang++ for x86 and run on a GPU-less box: tests/hostsim/build.py.
// Never part of the product: libkornia_hip.so is always built by hipcc for gfx950 (kornia-rs_amd/Makefile).
//
// Execution model: a launch runs its blocks one after another; inside a block every HIP thread is a fiber (ucontext) on
// one OS thread.  __syncthreads() and the wave-level operations (__shfl*, __builtin_amdgcn_wave_barrier) are fiber
// barriers, so LDS hand-offs and cross-lane reads see the lock-step semantics a 64-wide wave gives them on the GPU.
// `__shared__` is static storage (one block at a time).  Streams are synchronous, device memory is host memory.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct hostsim_stream;
namespace hostsim {
struct Lane { dim3 tid, bid, bdim, gdim; };
extern Lane* cur;                       // the fiber that is running
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void block_barrier();
void wave_barrier();
void launch_on(struct ::hostsim_stream* stream, dim3 grid, dim3 block, const std::function<void()>& body);
uint32_t shfl_bits(uint32_t v, int src_lane);  // value of `v` in lane `src_lane` of the caller's wave
int lane_id();
}  // namespace hostsim

#define threadIdx (hostsim::cur->tid)
#define blockIdx (hostsim::cur->bid)
#define blockDim (hostsim::cur->bdim)
#define gridDim (hostsim::cur->gdim)
// a launch on a stream that belongs to another device than the calling thread's current one is refused (sticky error, read by
// hipGetLastError) — stricter than some runtimes, so that a host layer that forgets hipSetDevice fails here, not on an 8-GPU node
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hostsim::launch_on((stream), dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

// ---- device intrinsics used by the kernels -------------------------------------------------------------------------
inline void __syncthreads() { hostsim::block_barrier(); }
#define __builtin_amdgcn_wave_barrier() hostsim::wave_barrier()
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)
inline int __shfl(int v, int lane) { return (int)hostsim::shfl_bits((uint32_t)v, lane); }
inline int __shfl_xor(int v, int mask) { return (int)hostsim::shfl_bits((uint32_t)v, hostsim::lane_id() ^ mask); }
// HIP: lanes whose source falls outside the wave keep their own value
inline int __shfl_up(int v, unsigned delta) { const int l = hostsim::lane_id(), s = l - (int)delta; return (int)hostsim::shfl_bits((uint32_t)v, s >= 0 ? s : l); }
inline int __shfl_down(int v, unsigned delta) { const int l = hostsim::lane_id(), s = l + (int)delta; return (int)hostsim::shfl_bits((uint32_t)v, s < 64 ? s : l); }
// v_mov_b32_dpp wave_shr:1 (0x138) / wave_shl:1 (0x130): lane i takes lane i -/+ 1, the end lane keeps `old`
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) {
    const int l = hostsim::lane_id(), s = ctrl == 0x138 ? l - 1 : l + 1;
    const int v = (int)hostsim::shfl_bits((uint32_t)src, s >= 0 && s < 64 ? s : l);
    return s >= 0 && s < 64 ? v : old;
}
inline float __shfl_xor(float v, int mask) {
    uint32_t b;
    memcpy(&b, &v, 4);
    b = hostsim::shfl_bits(b, hostsim::lane_id() ^ mask);
    memcpy(&v, &b, 4);
    return v;
}
inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
// v_perm_b32: the selector's byte i picks a byte of the 64-bit value {s0 (high dword), s1 (low dword)}: 0..3 -> s1, 4..7 -> s0,
// 8..11 -> sign of bytes 1 / 3 of s1 / s0 replicated, 12 -> 0x00, 13.. -> 0xff
inline uint32_t hostsim_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    const uint64_t v = ((uint64_t)s0 << 32) | s1;
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t k = (sel >> (8 * i)) & 0xffu;
        uint32_t b;
        if (k <= 7) b = (uint32_t)(v >> (8 * k)) & 0xffu;
        else if (k <= 11) b = ((v >> (16 * (k - 8) + 15)) & 1u) ? 0xffu : 0x00u;
        else b = k == 12 ? 0x00u : 0xffu;
        out |= b << (8 * i);
    }
    return out;
}
#define __builtin_amdgcn_perm(a, b, sel) hostsim_perm((a), (b), (sel))
#define __builtin_amdgcn_alignbyte(hi, lo, s) ((uint32_t)(((((uint64_t)(uint32_t)(hi)) << 32) | (uint32_t)(lo)) >> (8 * ((s) & 3))))
#define __builtin_amdgcn_sdot4(a, b, c, clamp) ((int)((c) + (int)(int8_t)((uint32_t)(a)) * (int)(int8_t)((uint32_t)(b)) + (int)(int8_t)((uint32_t)(a) >> 8) * (int)(int8_t)((uint32_t)(b) >> 8) + (int)(int8_t)((uint32_t)(a) >> 16) * (int)(int8_t)((uint32_t)(b) >> 16) + (int)(int8_t)((uint32_t)(a) >> 24) * (int)(int8_t)((uint32_t)(b) >> 24)))
#define __builtin_amdgcn_sdot2(a, b, c, clamp) ((int)((c) + (int)(a)[0] * (int)(b)[0] + (int)(a)[1] * (int)(b)[1]))
#define __builtin_amdgcn_udot4(a, b, c, clamp) ((uint32_t)((c) + ((uint32_t)(a) & 0xffu) * ((uint32_t)(b) & 0xffu) + (((uint32_t)(a) >> 8) & 0xffu) * (((uint32_t)(b) >> 8) & 0xffu) + \
                                                          (((uint32_t)(a) >> 16) & 0xffu) * (((uint32_t)(b) >> 16) & 0xffu) + ((uint32_t)(a) >> 24) * ((uint32_t)(b) >> 24)))
#define __builtin_amdgcn_udot2(a, b, c, clamp) ((uint32_t)((c) + (uint32_t)(a)[0] * (uint32_t)(b)[0] + (uint32_t)(a)[1] * (uint32_t)(b)[1]))
#define __builtin_amdgcn_readfirstlane(v) (v)  /* only ever applied to wave-uniform values */
// raw buffer accesses: base + soffset + voffset, dropped / zero when voffset + size runs past num_records (the hardware range check)
struct hostsim_rsrc { char* base; uint32_t n; };
typedef hostsim_rsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, n, flags) hostsim_rsrc{(char*)(p), (uint32_t)(n)}
template <typename V>
inline void hostsim_buf_store(V v, hostsim_rsrc r, int voff, int soff) {
    if ((uint64_t)(uint32_t)voff + sizeof(V) <= r.n) memcpy(r.base + soff + (uint32_t)voff, &v, sizeof(V));
}
template <int BYTES, typename V>  // a 3-element ext vector is padded to 16 bytes: store its first BYTES only
inline void hostsim_buf_store_n(V v, hostsim_rsrc r, int voff, int soff) {
    if ((uint64_t)(uint32_t)voff + BYTES <= r.n) memcpy(r.base + soff + (uint32_t)voff, &v, BYTES);
}
inline uint32_t hostsim_buf_load32(hostsim_rsrc r, int voff, int soff) {
    uint32_t v = 0;
    if ((uint64_t)(uint32_t)voff + 4 <= r.n) memcpy(&v, r.base + soff + (uint32_t)voff, 4);
    return v;
}
#define __builtin_amdgcn_raw_buffer_store_b128(v, r, vo, so, aux) hostsim_buf_store((v), (r), (vo), (so))
#define __builtin_amdgcn_raw_buffer_store_b96(v, r, vo, so, aux) hostsim_buf_store_n<12>((v), (r), (vo), (so))
#define __builtin_amdgcn_raw_buffer_store_b64(v, r, vo, so, aux) hostsim_buf_store((v), (r), (vo), (so))
#define __builtin_amdgcn_raw_buffer_store_b32(v, r, vo, so, aux) hostsim_buf_store((v), (r), (vo), (so))
#define __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, aux) hostsim_buf_load32((r), (vo), (so))

// one block, one fiber at a time: plain read-modify-write is atomic here
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (o < v) *p = v; return o; }
template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }

template <typename T> inline T min(T a, T b) { return b < a ? b : a; }
template <typename T> inline T max(T a, T b) { return a < b ? b : a; }
inline long long min(long long a, int b) { return a < b ? a : b; }
inline long long max(long long a, int b) { return a > b ? a : b; }
inline long long min(int a, long long b) { return a < b ? a : b; }
inline long long max(int a, long long b) { return a > b ? a : b; }
inline unsigned min(unsigned a, int b) { return a < (unsigned)b ? a : (unsigned)b; }
inline unsigned min(int a, unsigned b) { return (unsigned)a < b ? (unsigned)a : b; }
inline unsigned max(unsigned a, int b) { return a > (unsigned)b ? a : (unsigned)b; }
inline unsigned max(int a, unsigned b) { return (unsigned)a > b ? (unsigned)a : b; }

// ---- runtime API (tests/hostsim/hostsim.cpp) ---------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorInvalidHandle = 400, hipErrorNotSupported = 801 };
typedef struct hostsim_stream* hipStream_t;
typedef struct hostsim_event* hipEvent_t;
typedef struct hostsim_pool* hipMemPool_t;
typedef struct hostsim_graph* hipGraph_t;
typedef struct hostsim_graph_exec* hipGraphExec_t;
typedef struct hostsim_graph_node* hipGraphNode_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipMemAttachGlobal = 1 };
enum hipMemPoolAttr { hipMemPoolAttrReleaseThreshold = 4 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1, hipStreamCaptureStatusInvalidated = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t { hipMemoryType type; int device; };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMemGetInfo(size_t* f, size_t* t);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g);
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* status);
typedef int hipDevice_t;
hipError_t hipStreamGetDevice(hipStream_t s, hipDevice_t* device);
hipError_t hipEventQuery(hipEvent_t e);
#define HIP_VERSION 0
inline hipError_t hipRuntimeGetVersion(int* v) { *v = 0; return hipSuccess; }
hipError_t hipGraphGetNodes(hipGraph_t g, hipGraphNode_t* nodes, size_t* n);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t* err, char* log, size_t n);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);
hipError_t hipDeviceGetDefaultMemPool(hipMemPool_t* p, int d);
hipError_t hipMemPoolSetAttribute(hipMemPool_t p, hipMemPoolAttr a, void* v);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipMallocAsync(void** p, size_t n, hipStream_t s);
hipError_t hipFree(void* p);
hipError_t hipFreeAsync(void* p, hipStream_t s);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMallocManaged(void** p, size_t n, unsigned flags);
hipError_t hipMemset(void* p, int v, size_t n);
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t s);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, hipMemcpyKind k, hipStream_t st);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v);
