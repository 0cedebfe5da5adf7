/*
 * kornia_hip_testing.h — TEST HOOKS of libkornia_hip.so.  NOT part of the drop-in boundary (include/kornia_hip.h): a host binding
 * (the Rust `kornia-hip-sys` crate, cgo, JNI ...) does not declare these, and nothing in the product's host layer calls them.
 * They exist so that the parity tests can reach code paths that convenient inputs would not take.
 */
#ifndef KORNIA_HIP_TESTING_H
#define KORNIA_HIP_TESTING_H

#include "kornia_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* floor(n / d) through the multiply-shift division the kernels use for tile ids (n < 2^31), evaluated on the host.     */
KH_API uint32_t kh_debug_fast_quot(uint32_t n, uint32_t d);

/* Force one of the alternate code paths a launcher can take on inputs that would not reach it (the IEEE-division / four-tap /
 * LDS-tile / per-pixel fallbacks other geometries use anyway), e.g. ("pre_grid", 0), ("filter_force_tile", 1),
 * ("warp_u8_direct", 1); value -1 restores the production choice; an unknown name is KH_ERR_INVALID_ARG.
 * Scope: the CALLING THREAD's launches only (thread-local state) — no other thread of the process is rerouted, and there is
 * no process-wide switch to race on.  The library never reads the environment.                                           */
KH_API int32_t kh_debug_set_option(const char* name, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* KORNIA_HIP_TESTING_H */
