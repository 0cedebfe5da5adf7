/* A plain-C99 consumer of include/kornia_hip.h: what a cgo / JNI / Rust `extern "C"` binding sees.  Host-only entries are
 * called for real (they need no device); one compute entry is called with a bad argument to read the thread-local error text.
 * Built with `gcc -std=c99 -pedantic` and linked against libkornia_hip.so by tests/test_abi.py. */
#include <stdio.h>
#include <string.h>

#include "kornia_hip.h"

#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) { printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main(void) {
    float taps[5], inv[9], m[9] = {1, 0, 2, 0, 1, 3, 0, 0, 1};
    int32_t boxes[5], radius = 0, ntaps = 0;
    char msg[256];
    CHECK(kh_gaussian_kernel_1d(5, 0.5f, taps) == KH_OK);
    CHECK(taps[2] > 0.78f && taps[2] < 0.79f && taps[0] == taps[4]);
    CHECK(kh_box_blur_fast_kernels_1d(1.0f, 5, boxes) == KH_OK && boxes[0] == 1 && boxes[4] == 3);
    CHECK(kh_invert_homography(m, inv) == KH_OK && inv[2] == -2.0f && inv[5] == -3.0f);
    CHECK(kh_bilateral_tables(5, 50.0, 50.0, 0, &radius, &ntaps, NULL, NULL, NULL, NULL, NULL) == KH_OK && radius == 2 && ntaps == 13);
    CHECK(kh_median_blur_u8(NULL, (const uint8_t*)0x1000, (uint8_t*)0x2000, 8, 8, 3, 4, 1, 0, 0) == KH_ERR_INVALID_ARG);
    CHECK(kh_last_error(msg, sizeof msg) > 0 && strstr(msg, "kernel length 4") != NULL);
    CHECK(strlen(kh_version()) > 0);
    printf("c-abi ok: %s\n", kh_version());
    return 0;
}
