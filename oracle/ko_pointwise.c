/* CPU oracle (TEST INFRASTRUCTURE, never linked into the product): normalize_mean_std.
 * Follows crates/kornia-imgproc/src/normalize.rs:56-87: per pixel, per channel, `(src - mean[c]) / std[c]` — a true IEEE
 * division, no reciprocal — scheduled like parallel::par_iter_rows (P/parallel.rs:19-60: one task per row of the image).
 * parity pinned: the doc example of normalize.rs:30-56 (tests/test_oracle_geom_filter.py) and numpy's f32 arithmetic. */
#include "ko_oracle.h"

void ko_normalize_mean_std_f32(const float* src, float* dst, int width, int height, int channels, const float* mean,
                               const float* std) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; ++y) {
        const float* s = src + (size_t)y * width * channels;
        float* d = dst + (size_t)y * width * channels;
        for (int x = 0; x < width; ++x)
            for (int c = 0; c < channels; ++c) d[x * channels + c] = (s[x * channels + c] - mean[c]) / std[c];
    }
}
