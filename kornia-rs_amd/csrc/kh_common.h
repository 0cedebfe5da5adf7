// Shared host/device helpers for libkornia_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

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

constexpr int kWave = 64;       // gfx950 wavefront
constexpr int kBlock = 256;     // 4 waves: one per SIMD of a CU
constexpr int64_t kI32Max = 2147483647LL;

inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

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
