// The rest of include/kornia_hip.hpp (colour, camera formats, u8 twins, pyramid, morphology, crop / flip, min-max, maps, graphs).
// `host`: every wrapper classifies host operands and throws the typed error without touching a device.
// `gpu`:  known answers on a device.  Each check runs in its own try block: a value mismatch prints "FAIL", an exception
//         prints "THROW" (under scripts/glue_dryrun.py --cpp, whose mock computes nothing, only THROW lines are bugs).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>

#include "kornia_hip.hpp"

using namespace kornia;
using K = ImageError::Kind;

static int failures = 0, throws = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)
static void section(const char* name, const std::function<void()>& body) {
    try { body(); } catch (const std::exception& e) { std::printf("THROW %s: %s\n", name, e.what()); ++throws; }
}
template <typename F>
static bool throws_kind(K kind, F&& f) {
    try { f(); } catch (const ImageError& e) { return e.kind == kind; } catch (...) { return false; }
    return false;
}
template <typename T, int C>
static Image<T, C> host_img(size_t w, size_t h, T v = T()) { return Image<T, C>::from_size_val({w, h}, v); }

static void host_only() {
    auto u3 = host_img<uint8_t, 3>(8, 6), u3b = host_img<uint8_t, 3>(8, 6);
    auto u1 = host_img<uint8_t, 1>(8, 6);
    auto u4 = host_img<uint8_t, 4>(8, 6);
    auto f3 = host_img<float, 3>(8, 6), f3b = host_img<float, 3>(8, 6), f3s = host_img<float, 3>(4, 3);
    auto f1 = host_img<float, 1>(8, 6), f1b = host_img<float, 1>(8, 6);
    auto f4 = host_img<float, 4>(8, 6);
    auto d3 = host_img<double, 3>(8, 6), d3b = host_img<double, 3>(8, 6);
    auto d1 = host_img<double, 1>(8, 6);
    const K H = K::HostPathUnavailable;
    EXPECT(throws_kind(H, [&] { imgproc::rgb_from_gray(u1, u3); }));
    EXPECT(throws_kind(H, [&] { imgproc::rgb_from_gray(f1, f3); }));
    EXPECT(throws_kind(H, [&] { imgproc::bgr_from_rgb(u3, u3b); }));
    EXPECT(throws_kind(H, [&] { imgproc::bgr_from_rgb(f3, f3b); }));
    EXPECT(throws_kind(H, [&] { imgproc::rgba_from_rgb(u3, u4); }));
    EXPECT(throws_kind(H, [&] { imgproc::rgba_from_rgb(f3, f4, true); }));
    EXPECT(throws_kind(H, [&] { imgproc::rgb_from_rgba(u4, u3); }));
    EXPECT(throws_kind(H, [&] { imgproc::ycc_from_rgb(u3, u3b, imgproc::ChromaOrder::YCrCb); }));
    EXPECT(throws_kind(H, [&] { imgproc::rgb_from_ycc(f3, f3b, imgproc::ChromaOrder::YuvCbCr); }));
    EXPECT(throws_kind(H, [&] { imgproc::hsv_from_rgb(f3, f3b); }));
    EXPECT(throws_kind(H, [&] { imgproc::rgb_from_hls(f3, f3b); }));
    EXPECT(throws_kind(H, [&] { imgproc::sepia_from_rgb(u3, u3b); }));
    EXPECT(throws_kind(H, [&] { imgproc::cie_convert(f3, f3b, KH_CIE_LAB_FROM_RGB); }));
    EXPECT(throws_kind(H, [&] { imgproc::color_convert_f64(d3, d3b, KH_F64_HSV_FROM_RGB); }));
    EXPECT(throws_kind(K::InvalidChannelShape, [&] { imgproc::color_convert_f64(d3, d3b, KH_F64_GRAY_FROM_RGB); }));
    EXPECT(throws_kind(H, [&] { imgproc::color_convert_f64(d3, d1, KH_F64_GRAY_FROM_RGB); }));
    EXPECT(throws_kind(H, [&] { imgproc::apply_colormap(u1, u3, std::array<uint8_t, 768>{}); }));
    EXPECT(throws_kind(H, [&] { imgproc::rgb_from_planar420(nullptr, 0, u3, imgproc::Planar420::Nv12); }));
    EXPECT(throws_kind(H, [&] { imgproc::rgb_from_packed422(nullptr, 0, u3, imgproc::Packed422::Uyvy); }));
    EXPECT(throws_kind(H, [&] { imgproc::convert_yuyv_to_rgb_u8(nullptr, 0, u3, imgproc::YuvToRgbMode::Bt709Full); }));
    EXPECT(throws_kind(H, [&] { imgproc::rgb_from_bayer(u1, imgproc::BayerPattern::Grbg, u3); }));
    EXPECT(throws_kind(H, [&] { imgproc::nv12_from_rgb(u3, nullptr); }));
    EXPECT(throws_kind(H, [&] { imgproc::yuyv_from_rgb(u3, nullptr); }));
    EXPECT(throws_kind(H, [&] { imgproc::resize_mapped(f3, f3s, InterpolationMode::Bicubic, imgproc::PixelMapping::AlignCorners); }));
    EXPECT(throws_kind(H, [&] { imgproc::resize_bilinear_normalize(f3, f3s, {0, 0, 0}, {1, 1, 1}); }));
    auto u3s = host_img<uint8_t, 3>(4, 3);
    EXPECT(throws_kind(H, [&] { imgproc::resize_opencv(u3, u3s, InterpolationMode::Bilinear); }));
    EXPECT(throws_kind(H, [&] { imgproc::resize_opencv(f3, f3s, InterpolationMode::Nearest); }));
    EXPECT(throws_kind(H, [&] { imgproc::remap(u3, u3b, f1, f1b, InterpolationMode::Bilinear); }));
    EXPECT(throws_kind(H, [&] { imgproc::warp_perspective_u8(u3, u3b, {1, 0, 0, 0, 1, 0, 0, 0, 1}); }));
    EXPECT(throws_kind(H, [&] { imgproc::box_blur(u3, u3b, {3, 3}); }));
    EXPECT(throws_kind(H, [&] { imgproc::separable_filter(f3, f3b, {0.25f, 0.5f, 0.25f}, {1.0f}); }));
    EXPECT(throws_kind(H, [&] { imgproc::sobel(f3, f3b, 3); }));
    EXPECT(throws_kind(H, [&] { imgproc::scharr(f3, f3b); }));
    EXPECT(throws_kind(H, [&] { imgproc::spatial_gradient_float(f3, f3b, f3b); }));
    EXPECT(throws_kind(H, [&] { imgproc::scharr_spatial_gradient_float(f3, f3b, f3b); }));
    EXPECT(throws_kind(H, [&] { imgproc::box_blur_fast(f3, f3b, {1.0f, 1.0f}); }));
    EXPECT(throws_kind(H, [&] { imgproc::median_blur(u3, u3b, 3); }));
    EXPECT(throws_kind(H, [&] { imgproc::bilateral_filter(u1, u1, 5, 50.0, 50.0); }));
    EXPECT(throws_kind(H, [&] { imgproc::pyrdown(f3, f3s); }));
    EXPECT(throws_kind(H, [&] { imgproc::pyrup(u3s, u3); }));
    imgproc::Kernel cross(imgproc::KernelShape::Cross, 3);  // host helper: no device needed
    EXPECT(cross.mask == std::vector<uint8_t>({0, 1, 0, 1, 1, 1, 0, 1, 0}));
    EXPECT(throws_kind(K::InvalidImageSize, [] { imgproc::Kernel(imgproc::KernelShape::Box, 3, 5); }));  // box / cross are square
    EXPECT(throws_kind(H, [&] { imgproc::dilate(u3, u3b, cross, imgproc::PaddingMode::Replicate); }));
    EXPECT(throws_kind(H, [&] { imgproc::erode(u1, u1, cross, imgproc::PaddingMode::Constant, {7}); }));
    EXPECT(throws_kind(H, [&] { imgproc::crop_image(u3, u3s, 1, 1); }));
    EXPECT(throws_kind(H, [&] { imgproc::horizontal_flip(f3, f3b); }));
    EXPECT(throws_kind(H, [&] { imgproc::vertical_flip(u4, u4); }));
    EXPECT(throws_kind(H, [&] { (void)imgproc::find_min_max(f3); }));
    EXPECT(throws_kind(H, [&] { imgproc::normalize_min_max(f3, f3b, 0.0f, 1.0f); }));
    EXPECT(throws_kind(H, [&] { imgproc::generate_correction_map_polynomial(f1, f1b, {500, 500, 4, 3}, {}); }));
    EXPECT(throws_kind(K::InvalidImageSize, [] { Graph::capture(Stream::from_handle(nullptr, 0), [] {}); }));  // default stream
}

template <typename T, int C>
static Image<T, C> up(const Stream& s, size_t w, size_t h, std::vector<T> v) { return Image<T, C>::from_size_vec({w, h}, std::move(v)).to_hip(s); }

static void on_device() {
    Stream s = Stream::create(0);
    section("swizzles", [&] {
        auto rgb = up<uint8_t, 3>(s, 2, 1, {1, 2, 3, 4, 5, 6});
        auto bgr = Image<uint8_t, 3>::zeros_hip({2, 1}, s);
        imgproc::bgr_from_rgb(rgb, bgr);
        EXPECT(bgr.to_host().as_slice() == std::vector<uint8_t>({3, 2, 1, 6, 5, 4}));
        auto rgba = Image<uint8_t, 4>::zeros_hip({2, 1}, s);
        imgproc::rgba_from_rgb(rgb, rgba);
        EXPECT(rgba.to_host().as_slice() == std::vector<uint8_t>({1, 2, 3, 255, 4, 5, 6, 255}));  // P/color/convert.rs:546-560
        auto back = Image<uint8_t, 3>::zeros_hip({2, 1}, s);
        imgproc::rgb_from_rgba(rgba, back, nullptr, true);
        EXPECT(back.to_host().as_slice() == std::vector<uint8_t>({3, 2, 1, 6, 5, 4}));
        auto half = up<uint8_t, 4>(s, 1, 1, {255, 0, 0, 128});
        auto one = Image<uint8_t, 3>::zeros_hip({1, 1}, s);
        const std::array<uint8_t, 3> bg{100, 100, 100};
        imgproc::rgb_from_rgba(half, one, &bg);
        EXPECT(one.to_host().as_slice() == std::vector<uint8_t>({178, 50, 50}));  // convert.rs:498-520
        auto g = up<float, 1>(s, 2, 1, {0.25f, 1.0f});
        auto g3 = Image<float, 3>::zeros_hip({2, 1}, s);
        imgproc::rgb_from_gray(g, g3);
        EXPECT(g3.to_host().as_slice() == std::vector<float>({0.25f, 0.25f, 0.25f, 1.0f, 1.0f, 1.0f}));
    });
    section("colour spaces", [&] {
        auto grey = up<uint8_t, 3>(s, 1, 1, {128, 128, 128});
        auto ycc = Image<uint8_t, 3>::zeros_hip({1, 1}, s);
        imgproc::ycc_from_rgb(grey, ycc, imgproc::ChromaOrder::YCrCb);
        EXPECT(ycc.to_host().as_slice() == std::vector<uint8_t>({128, 128, 128}));  // ycc_u8_known_value_gray
        auto red = up<float, 3>(s, 1, 1, {255.0f, 0.0f, 0.0f});
        auto hsv = Image<float, 3>::zeros_hip({1, 1}, s), rgb = Image<float, 3>::zeros_hip({1, 1}, s);
        imgproc::hsv_from_rgb(red, hsv);
        EXPECT(hsv.to_host().as_slice() == std::vector<float>({0.0f, 255.0f, 255.0f}));
        imgproc::rgb_from_hsv(hsv, rgb);
        EXPECT(std::fabs(rgb.to_host().as_slice()[0] - 255.0f) < 1e-3f);
        auto white = up<uint8_t, 3>(s, 1, 1, {255, 255, 255});
        auto sep = Image<uint8_t, 3>::zeros_hip({1, 1}, s);
        imgproc::sepia_from_rgb(white, sep);
        EXPECT(sep.to_host().as_slice() == std::vector<uint8_t>({255, 255, 240}));  // sepia_u8_known_value
        auto unit = up<float, 3>(s, 1, 1, {1.0f, 1.0f, 1.0f});
        auto lab = Image<float, 3>::zeros_hip({1, 1}, s);
        imgproc::cie_convert(unit, lab, KH_CIE_LAB_FROM_RGB);
        EXPECT(std::fabs(lab.to_host().as_slice()[0] - 100.0f) < 1e-2f);  // white: L* = 100
        auto d = up<double, 3>(s, 1, 1, {1.0, 0.0, 0.0});
        auto dg = Image<double, 1>::zeros_hip({1, 1}, s);
        imgproc::color_convert_f64(d, dg, KH_F64_GRAY_FROM_RGB);
        EXPECT(dg.to_host().as_slice()[0] == 0.299);
        auto idx = up<uint8_t, 1>(s, 2, 1, {0, 255});
        auto col = Image<uint8_t, 3>::zeros_hip({2, 1}, s);
        std::array<uint8_t, 768> lut{};
        for (int i = 0; i < 256; ++i) { lut[i] = (uint8_t)i; lut[256 + i] = (uint8_t)(255 - i); lut[512 + i] = 7; }
        imgproc::apply_colormap(idx, col, lut);
        EXPECT(col.to_host().as_slice() == std::vector<uint8_t>({0, 255, 7, 255, 0, 7}));
    });
    section("camera formats", [&] {
        auto yuyv = up<uint8_t, 1>(s, 4, 1, {16, 128, 16, 128});  // Y=16, U=V=128 -> black (limited range), packed422_known_gray
        auto rgb = Image<uint8_t, 3>::from_size_val({2, 1}, 9).to_hip(s);
        imgproc::rgb_from_packed422(yuyv.device_ptr(), 4, rgb, imgproc::Packed422::Yuyv);
        EXPECT(rgb.to_host().as_slice() == std::vector<uint8_t>(6, 0));
        imgproc::convert_yuyv_to_rgb_u8(yuyv.device_ptr(), 4, rgb, imgproc::YuvToRgbMode::Bt601Full);
        EXPECT(rgb.to_host().as_slice() == std::vector<uint8_t>(6, 16));  // full range: Y passes through at neutral chroma
        auto mosaic = up<uint8_t, 1>(s, 4, 4, {10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110, 120, 130, 140, 150, 160});
        auto demo = Image<uint8_t, 3>::zeros_hip({4, 4}, s);
        imgproc::rgb_from_bayer(mosaic, imgproc::BayerPattern::Rggb, demo);  // rggb_interior_known_value, corners_use_replicate_border
        const auto dm = demo.to_host();
        EXPECT(dm.as_slice()[(4 + 1) * 3] == 60 && dm.as_slice()[(4 + 1) * 3 + 1] == 60 && dm.as_slice()[(4 + 2) * 3 + 2] == 70 && dm.as_slice()[0] == 60);
        auto solid = Image<uint8_t, 3>::from_size_val({4, 2}, 200).to_hip(s);
        auto nv12 = Image<uint8_t, 1>::zeros_hip({4 * 2 * 3 / 2, 1}, s);
        imgproc::nv12_from_rgb(solid, nv12.device_ptr_mut());
        auto round = Image<uint8_t, 3>::zeros_hip({4, 2}, s);
        imgproc::rgb_from_planar420(nv12.device_ptr(), 12, round, imgproc::Planar420::Nv12);
        const auto host1 = round.to_host();  // keep the copy alive for the loop (a range-for over a temporary's member dangles)
        for (uint8_t v : host1.as_slice()) EXPECT(std::abs((int)v - 200) <= 2);  // encode_decode_constant_is_exact
        EXPECT(throws_kind(K::InvalidImageSize, [&] { imgproc::rgb_from_planar420(nv12.device_ptr(), 11, round, imgproc::Planar420::Nv12); }));
    });
    section("resize family", [&] {
        std::vector<float> ramp(5 * 5 * 3);
        for (size_t i = 0; i < ramp.size(); ++i) ramp[i] = (float)i;
        auto src = up<float, 3>(s, 5, 5, ramp);
        auto dst = Image<float, 3>::zeros_hip({3, 3}, s);
        imgproc::resize_mapped(src, dst, InterpolationMode::Bicubic, imgproc::PixelMapping::AlignCorners);  // (5-1)/(3-1) = 2: samples on pixels
        auto out = dst.to_host();
        for (int y = 0; y < 3; ++y)
            for (int x = 0; x < 3; ++x)
                for (int c = 0; c < 3; ++c) EXPECT(out.as_slice()[(y * 3 + x) * 3 + c] == ramp[((2 * y) * 5 + 2 * x) * 3 + c]);
        imgproc::resize_bilinear_normalize(src, dst, {1.0f, 2.0f, 3.0f}, {2.0f, 2.0f, 2.0f}, imgproc::PixelMapping::AlignCorners);
        EXPECT(dst.to_host().as_slice()[0] == (ramp[0] - 1.0f) * 0.5f && dst.to_host().as_slice()[26] == (ramp[74] - 3.0f) * 0.5f);
        EXPECT(throws_kind(K::InvalidImageSize, [&] { imgproc::resize_bilinear_normalize(src, dst, {0, 0, 0}, {1, 0, 1}); }));
        auto u = up<uint8_t, 1>(s, 4, 1, {10, 20, 30, 40});
        auto h = Image<uint8_t, 1>::zeros_hip({2, 1}, s);
        imgproc::resize_opencv(u, h, InterpolationMode::Nearest);
        EXPECT(h.to_host().as_slice() == std::vector<uint8_t>({10, 30}));  // nearest_uses_floor_semantics
    });
    section("u8 twins, filters", [&] {
        auto src = up<uint8_t, 1>(s, 4, 2, {10, 20, 30, 40, 50, 60, 70, 80});
        auto dst = Image<uint8_t, 1>::zeros_hip({4, 2}, s);
        imgproc::warp_perspective_u8(src, dst, {-1, 0, 3, 0, 1, 0, 0, 0, 1});
        EXPECT(dst.to_host().as_slice() == std::vector<uint8_t>({40, 30, 20, 10, 80, 70, 60, 50}));
        auto mx = up<float, 1>(s, 4, 2, {3, 2, 1, 0, 3, 2, 1, 0}), my = up<float, 1>(s, 4, 2, {0, 0, 0, 0, 1, 1, 1, 1});
        imgproc::remap(src, dst, mx, my, InterpolationMode::Nearest);
        EXPECT(dst.to_host().as_slice() == std::vector<uint8_t>({40, 30, 20, 10, 80, 70, 60, 50}));
        auto flat = Image<uint8_t, 3>::from_size_val({9, 7}, 77).to_hip(s);
        auto blur = Image<uint8_t, 3>::zeros_hip({9, 7}, s);
        imgproc::box_blur(flat, blur, {3, 3});
        const auto host2 = blur.to_host();  // keep the copy alive for the loop (a range-for over a temporary's member dangles)
        for (uint8_t v : host2.as_slice()) EXPECT(v == 77);
        std::vector<float> imp(25, 0.0f);
        imp[12] = 9.0f;
        auto f = up<float, 1>(s, 5, 5, imp);
        auto g = Image<float, 1>::zeros_hip({5, 5}, s);
        imgproc::separable_filter(f, g, {1.0f, 1.0f, 1.0f}, {1.0f, 1.0f, 1.0f});  // test_separable_filter_f32: a 3x3 plateau
        auto r = g.to_host();
        EXPECT(r.as_slice()[6] == 9.0f && r.as_slice()[12] == 9.0f && r.as_slice()[18] == 9.0f && r.as_slice()[0] == 0.0f);
        auto cst = Image<float, 1>::from_size_val({6, 6}, 3.0f).to_hip(s);
        auto mag = Image<float, 1>::from_size_val({6, 6}, 1.0f).to_hip(s);
        imgproc::sobel(cst, mag, 3);
        EXPECT(mag.to_host().as_slice()[14] == 0.0f);  // no gradient inside a constant image
        imgproc::scharr(cst, mag);
        EXPECT(mag.to_host().as_slice()[21] == 0.0f);
        std::vector<float> ramp25(25);
        for (int i = 0; i < 25; ++i) ramp25[i] = (float)i;
        auto rsrc = up<float, 1>(s, 5, 5, ramp25);
        auto gx = Image<float, 1>::zeros_hip({5, 5}, s), gy = Image<float, 1>::zeros_hip({5, 5}, s);
        imgproc::spatial_gradient_float(rsrc, gx, gy);  // test_spatial_gradient: 1 per x step, 5 per y step, halved on the border
        EXPECT(gx.to_host().as_slice()[12] == 1.0f && gx.to_host().as_slice()[10] == 0.5f && gy.to_host().as_slice()[12] == 5.0f && gy.to_host().as_slice()[2] == 2.5f);
        imgproc::scharr_spatial_gradient_float(rsrc, gx, gy);  // test_scharr_spatial_gradient
        EXPECT(gx.to_host().as_slice()[12] == 1.0f && gy.to_host().as_slice()[12] == 5.0f);
        auto fast = Image<float, 1>::zeros_hip({5, 5}, s);
        imgproc::box_blur_fast(rsrc, fast, {0.5f, 0.5f});  // test_box_blur_fast: exact floats
        EXPECT(fast.to_host().as_slice()[0] == 4.444444f && fast.to_host().as_slice()[12] == 12.0f && fast.to_host().as_slice()[24] == 19.555555f);
        auto flat8 = Image<uint8_t, 1>::from_size_val({16, 12}, 200).to_hip(s);
        auto out8 = Image<uint8_t, 1>::zeros_hip({16, 12}, s);
        imgproc::median_blur(flat8, out8, 5);  // constant_image_unchanged (median.rs, bilateral.rs)
        const auto m8 = out8.to_host();
        for (uint8_t v : m8.as_slice()) EXPECT(v == 200);
        imgproc::bilateral_filter(flat8, out8, 5, 50.0, 50.0);
        const auto b8 = out8.to_host();
        for (uint8_t v : b8.as_slice()) EXPECT(v == 200);
        EXPECT(throws_kind(K::InvalidImageSize, [&] { imgproc::median_blur(flat8, out8, 4); }));  // rejects_bad_ksize_and_size_mismatch
    });
    section("pyramid, morphology", [&] {
        std::vector<float> ramp(16);
        for (int i = 0; i < 16; ++i) ramp[i] = (float)i;
        auto src = up<float, 1>(s, 4, 4, ramp);
        auto dn = Image<float, 1>::zeros_hip({2, 2}, s);
        imgproc::pyrdown(src, dn);
        const float want[4] = {3.75f, 4.875f, 8.25f, 9.375f};  // test_pyrdown, "verified with opencv"
        auto got = dn.to_host();
        for (int i = 0; i < 4; ++i) EXPECT(std::fabs(got.as_slice()[i] - want[i]) < 1e-4f);
        auto flat = Image<uint8_t, 3>::from_size_val({3, 2}, 200).to_hip(s);
        auto upi = Image<uint8_t, 3>::zeros_hip({6, 4}, s);
        imgproc::pyrup(flat, upi);
        const auto host3 = upi.to_host();  // keep the copy alive for the loop (a range-for over a temporary's member dangles)
        for (uint8_t v : host3.as_slice()) EXPECT(v == 200);
        EXPECT(throws_kind(K::InvalidImageSize, [&] { imgproc::pyrdown(src, src); }));
        std::vector<uint8_t> dot(25, 0);
        dot[12] = 255;
        auto img = up<uint8_t, 1>(s, 5, 5, dot);
        auto out = Image<uint8_t, 1>::zeros_hip({5, 5}, s);
        imgproc::dilate(img, out, imgproc::Kernel(imgproc::KernelShape::Box, 3), imgproc::PaddingMode::Constant);
        auto d = out.to_host();
        int lit = 0;
        for (uint8_t v : d.as_slice()) lit += v == 255;
        EXPECT(lit == 9 && d.as_slice()[6] == 255 && d.as_slice()[0] == 0);  // test_dilate_3x3
        imgproc::erode(out, img, imgproc::Kernel(imgproc::KernelShape::Box, 3), imgproc::PaddingMode::Constant);
        lit = 0;
        const auto host4 = img.to_host();  // keep the copy alive for the loop (a range-for over a temporary's member dangles)
        for (uint8_t v : host4.as_slice()) lit += v == 255;
        EXPECT(lit == 1);
    });
    section("crop, flip, min / max, maps", [&] {
        auto src = up<uint8_t, 1>(s, 3, 2, {1, 2, 3, 4, 5, 6});
        auto dst = Image<uint8_t, 1>::zeros_hip({3, 2}, s);
        imgproc::horizontal_flip(src, dst);
        EXPECT(dst.to_host().as_slice() == std::vector<uint8_t>({3, 2, 1, 6, 5, 4}));  // test_hflip
        imgproc::vertical_flip(src, dst);
        EXPECT(dst.to_host().as_slice() == std::vector<uint8_t>({4, 5, 6, 1, 2, 3}));  // test_vflip
        auto win = Image<uint8_t, 1>::zeros_hip({2, 1}, s);
        imgproc::crop_image(src, win, 1, 1);
        EXPECT(win.to_host().as_slice() == std::vector<uint8_t>({5, 6}));
        EXPECT(throws_kind(K::InvalidImageSize, [&] { imgproc::crop_image(src, win, 2, 1); }));  // test_crop_oob_returns_err
        auto f = up<float, 1>(s, 2, 2, {0.0f, 1.0f, 2.0f, 3.0f});
        auto mm = imgproc::find_min_max(f);
        EXPECT(mm.first == 0.0f && mm.second == 3.0f);
        auto n = Image<float, 1>::zeros_hip({2, 2}, s);
        imgproc::normalize_min_max(f, n, 0.0f, 1.0f);
        auto r = n.to_host();
        EXPECT(r.as_slice()[0] == 0.0f && std::fabs(r.as_slice()[1] - 1.0f / 3.0f) < 1e-6f && r.as_slice()[3] == 1.0f);
        auto mx = Image<float, 1>::zeros_hip({8, 4}, s), my = Image<float, 1>::zeros_hip({8, 4}, s);
        imgproc::generate_correction_map_polynomial(mx, my, {500.0, 500.0, 4.0, 2.0}, {});  // no distortion: identity maps
        auto hx = mx.to_host(), hy = my.to_host();
        EXPECT(std::fabs(hx.as_slice()[8 + 5] - 5.0f) < 1e-4f && std::fabs(hy.as_slice()[3 * 8 + 1] - 3.0f) < 1e-4f);
    });
    section("graph", [&] {
        auto src = up<uint8_t, 3>(s, 2, 1, {0, 128, 255, 128, 0, 128});
        auto gray = Image<uint8_t, 1>::zeros_hip({2, 1}, s);
        s.synchronize();
        Graph g = Graph::capture(s, [&] { imgproc::gray_from_rgb(src, gray); });
        s.synchronize();
        g.replay();
        g.replay();
        auto out = gray.to_host();
        EXPECT(out.as_slice()[0] == 104 && out.as_slice()[1] == 53);
    });
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "host";
    section("host contract", host_only);
    if (mode == "gpu") on_device();
    std::printf("%s: %d failure(s), %d exception(s) [%s]\n", mode.c_str(), failures, throws, kh_version());
    return (failures || throws) ? 1 : 0;
}
