// Dev micro-benchmark (round 5, not shipped): how many cycles does a wave64 VALU instruction occupy its SIMD for on gfx950?
// Counter-derived "VALU busy" figures of rounds 3-4 assumed 4 (a 16-lane SIMD); the guide says 2 (SIMD-32); the four-tap letterbox
// experiment (r05b/c) only makes sense with 2.  Here: every SIMD of the chip runs 8 waves of a dependent-free stream of one instruction
// type; instructions per second per SIMD against the shader clock (s_memtime) gives the issue cost.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)


#define INSTR_LIST(X) \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %0", 1) \
    X(1, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %0", 2) \
    X(2, "v_mul_f32", "v_mul_f32 %0, %0, %1", 1) \
    X(3, "v_add_u32", "v_add_u32 %0, %0, %1", 0) \
    X(4, "v_and_b32", "v_and_b32 %0, %0, %1", 0) \
    X(5, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0", 0) \
    X(6, "v_ashrrev_i32", "v_ashrrev_i32 %0, 3, %0", 0) \
    X(7, "v_bfe_u32", "v_bfe_u32 %0, %0, 8, 8", 0) \
    X(8, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc", 0) \
    X(9, "v_perm_b32", "v_perm_b32 %0, %0, %1, %0", 0) \
    X(10, "v_add3_u32", "v_add3_u32 %0, %0, %1, %0", 0) \
    X(11, "v_mul_i32_i24", "v_mul_i32_i24 %0, %0, %1", 0) \
    X(12, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %0", 0) \
    X(13, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %0", 0) \
    X(14, "v_med3_i32", "v_med3_i32 %0, %0, %1, %0", 0) \
    X(15, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", 0) \
    X(16, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0", 0) \
    X(17, "v_dot4_i32_i8", "v_dot4_i32_i8 %0, %1, %1, %0", 0) \
    X(18, "v_dot2_i32_i16", "v_dot2_i32_i16 %0, %1, %1, %0", 0) \
    X(19, "v_lshrrev_b64", "v_lshrrev_b64 %0, 3, %0", 3) \
    X(20, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", 0) \
    X(21, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1", 2) \
    X(22, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1", 2) \
    X(23, "v_add_u32_sdwa", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", 0) \
    X(24, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %1, %0 bitop3:0x96", 0)
constexpr int kKinds = 25;

// TYPE 0: 32-bit integer register, 1: f32, 2: 64-bit pair (packed f32), 3: 64-bit integer
template <int KIND>
__global__ __launch_bounds__(256) void spin(uint32_t* out, int iters) {
    uint32_t a[16];
    uint64_t d[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 17 + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = ((uint64_t)a[2 * i] << 32) | a[2 * i + 1];
    for (int it = 0; it < iters; ++it) {
#define X(K, NAME, TXT, TYPE) \
        if constexpr (KIND == K) { \
            if constexpr (TYPE == 2 || TYPE == 3) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(TXT : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7])); } \
            else { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(TXT : "+v"(a[i]) : "v"(a[(i + 1) & 15])); } \
        }
        INSTR_LIST(X)
#undef X
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (uint32_t)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int K>
void launch_kind(int kind, int blocks, uint32_t* out, int iters) {
    if constexpr (K < kKinds) {
        if (kind == K) hipLaunchKernelGGL(spin<K>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else launch_kind<K + 1>(kind, blocks, out, iters);
    }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 20000;
    uint32_t* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[kKinds] = {
#define X(K, NAME, TXT, TYPE) NAME,
        INSTR_LIST(X)
#undef X
    };
    printf("# %s, %d CUs, clockRate %d kHz; %d x 16 instructions per wave (8 independent chains for the 64-bit forms)\n", p.gcnArchName, cus, p.clockRate, iters);
    printf("# light: 8 blocks on the whole chip (one wave per SIMD on 8 CUs: no power throttling, clock ~ clockRate); full: 8 waves on every SIMD\n");
    printf("%-18s %10s %26s %10s %30s\n", "instruction", "light ms", "cycles / instr @ clockRate", "full ms", "full: ns per wave-instr per SIMD");
    for (int k = 0; k < kKinds; ++k) {
        float ms[2];
        for (int mode = 0; mode < 2; ++mode) {
            const int blocks = mode == 0 ? 8 : cus * 8;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                launch_kind<0>(k, blocks, out, iters);
                CK(hipGetLastError());
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            }
            CK(hipEventElapsedTime(&ms[mode], e0, e1));
        }
        const double n = (double)iters * 16;
        printf("%-18s %10.3f %26.2f %10.3f %30.3f\n", names[k], ms[0], ms[0] * 1e-3 * p.clockRate * 1e3 / n, ms[1], ms[1] * 1e6 / (8.0 * n));
    }
    return 0;
}
