// Dev check (round 3): v_mov_b32_dpp wave_shr:1 / wave_shl:1 on gfx950 — lane i takes lane i -/+ 1, the end lane keeps `old`.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const int* a, int* b) {
    int v = a[threadIdx.x], h = a[threadIdx.x + 64];
    b[threadIdx.x] = __builtin_amdgcn_update_dpp(h, v, 0x138, 0xf, 0xf, false);
    b[threadIdx.x + 64] = __builtin_amdgcn_update_dpp(h, v, 0x130, 0xf, 0xf, false);
}
int main() {
    int ha[128], hb[128], *a, *b;
    for (int i = 0; i < 128; ++i) ha[i] = i < 64 ? 1000 + i : 5000 + i;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMemcpy(a, ha, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b); hipMemcpy(hb, b, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int shr = i == 0 ? 5000 + 64 : 1000 + i - 1, shl = i == 63 ? 5000 + 127 : 1000 + i + 1;
        if (hb[i] != shr || hb[64 + i] != shl) { ++bad; printf("lane %d: shr %d (want %d) shl %d (want %d)\n", i, hb[i], shr, hb[64 + i], shl); }
    }
    printf("dpp wave shifts: %d lanes differ from the expectation\n", bad);
    return bad != 0;
}
