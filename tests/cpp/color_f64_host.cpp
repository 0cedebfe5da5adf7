// Host build of the product's f64 per-pixel colour arithmetic (kornia-rs_amd/csrc/kh_color_f64.h — the SAME source the
// gfx950 kernel compiles) so a GPU-less box can compare it with the CPU restatement of the reference.
#include <cstddef>

#include "kh_color_f64.h"

extern "C" int host_color_convert_f64(const double* src, double* dst, size_t npixels, int conv) {
    if (conv < 0 || conv >= kh_f64::kCount) return -1;
    const int cin = kh_f64::channels_in(conv), cout = kh_f64::channels_out(conv);
    for (size_t i = 0; i < npixels; ++i) {
        double in[3] = {0, 0, 0}, out[3];
        for (int c = 0; c < cin; ++c) in[c] = src[i * cin + c];
        kh_f64::convert_pixel(conv, in, out);
        for (int c = 0; c < cout; ++c) dst[i * cout + c] = out[c];
    }
    return 0;
}
