// Exercises include/kornia_hip.hpp.  `host` mode (no GPU): residency rules and typed errors only — no compute
// entry is reached.  `gpu` mode: the reference's own known answers through the C++ mirror on a device.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "kornia_hip.hpp"

using namespace kornia;
using K = ImageError::Kind;

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)
template <typename F>
static bool throws(K kind, F&& f) {
    try { f(); } catch (const ImageError& e) { return e.kind == kind; } catch (...) { return false; }
    return false;
}

static void host_only() {
    auto a = Image<float, 3>::from_size_val({4, 5}, 0.25f);
    EXPECT(a.width() == 4 && a.height() == 5 && a.numel() == 60 && !a.is_device() && a.domain() == MemoryDomain::Host);
    EXPECT(a.as_slice().size() == 60 && a.as_slice()[7] == 0.25f && a.stream() == nullptr);
    EXPECT(throws(K::InvalidChannelShape, [] { Image<float, 3>::from_size_vec({4, 5}, std::vector<float>(59)); }));
    EXPECT(throws(K::UnsupportedDevice, [&] { (void)a.device_ptr(); }));
    EXPECT(throws(K::UnsupportedDevice, [&] { (void)a.to_host(); }));
    auto b = Image<float, 3>::from_size_val({2, 3}, 0.0f);
    // host/host pair: this build has no CPU path and says so — never a silent fallback (P/cuda/dispatch.rs:203-211)
    EXPECT(throws(K::HostPathUnavailable, [&] { imgproc::resize(a, b, InterpolationMode::Bilinear); }));
    EXPECT(throws(K::HostPathUnavailable, [&] { imgproc::gaussian_blur(a, a, {3, 3}, {1.0f, 1.0f}); }));
    Stream borrowed = Stream::from_handle(nullptr, 0);  // no runtime call: a borrowed handle is only recorded
    EXPECT(borrowed.device() == 0 && borrowed.handle() == nullptr);
    EXPECT(throws(K::InvalidNormalize, [&] { Preprocessor(borrowed, ResizeMode::Letterbox, SourceFormat::Rgb8, {0, 0, 0}, {1, 0, 1}); }));
    EXPECT(throws(K::InvalidNormalize, [&] { Preprocessor(borrowed, ResizeMode::Letterbox, SourceFormat::Rgb8, {0, NAN, 0}, {1, 1, 1}); }));
    // raw-frame validation happens before any launch (SourceFormat::dims_ok / buffer_len, P/preprocess.rs:131-250)
    Preprocessor pre(borrowed, ResizeMode::Letterbox, SourceFormat::Nv12);
    EXPECT(throws(K::InvalidImageSize, [&] { pre.run_raw(nullptr, 1 << 20, 15, 8, nullptr, 8, 8); }));  // odd width
    EXPECT(throws(K::InvalidImageSize, [&] { pre.run_raw(nullptr, 16 * 8, 16, 8, nullptr, 8, 8); }));  // luma only: chroma plane missing
    Preprocessor yuyv(borrowed, ResizeMode::Stretch, SourceFormat::Yuyv);
    EXPECT(throws(K::InvalidImageSize, [&] { yuyv.run_raw(nullptr, 16 * 8 * 2 - 1, 16, 8, nullptr, 8, 8); }));
    // allocators (T/allocator.rs:73-146): layouts are validated, host allocations are zeroed and aligned, zero-size is legal
    using AK = TensorAllocatorError::Kind;
    auto alloc_throws = [](AK kind, auto&& f) {
        try { f(); } catch (const TensorAllocatorError& e) { return e.kind() == kind; } catch (...) { return false; }
        return false;
    };
    EXPECT(alloc_throws(AK::LayoutError, [] { Layout(16, 3); }));
    EXPECT(alloc_throws(AK::LayoutError, [] { Layout(16, 0); }));
    EXPECT(Layout::array<float>(10).size == 40 && Layout::array<float>(10).align == alignof(float));
    {
        const TensorAllocator& ha = host_alloc();
        EXPECT(&host_alloc() == &host_alloc() && ha.domain() == MemoryDomain::Host);
        MemoryResource r = ha.allocate(Layout(1000, 256));
        EXPECT(r.as_ptr() && r.len_bytes() == 1000 && r.domain() == MemoryDomain::Host && reinterpret_cast<uintptr_t>(r.as_ptr()) % 256 == 0 && !r.is_readonly());
        bool zero = true;
        for (size_t i = 0; i < 1000; ++i) zero = zero && static_cast<const unsigned char*>(r.as_ptr())[i] == 0;
        EXPECT(zero && r.stream() == nullptr);
        MemoryResource moved = std::move(r);
        EXPECT(r.as_ptr() == nullptr && moved.len_bytes() == 1000);
        MemoryResource empty = ha.allocate(Layout(0, 64));
        EXPECT(empty.as_ptr() == nullptr && empty.len_bytes() == 0);
    }
}

static void on_device() {
    Stream s = Stream::create(0), other = Stream::create(0);
    {   // the HIP allocators that replace CudaAllocator / PinnedAllocator / CudaUnifiedAllocator (T/cuda.rs:214-262, 355-380, 440-511)
        HipAllocator da(s);
        PinnedAllocator pa;
        HipUnifiedAllocator ua(s);
        EXPECT(da.domain() == MemoryDomain::Device && pa.domain() == MemoryDomain::Host && ua.domain() == MemoryDomain::Unified);
        const Layout l = Layout::array<uint32_t>(1024);
        MemoryResource d = da.allocate(l), pin = pa.allocate(l), uni = ua.allocate(l);
        int32_t dom = -1, dev = -1;
        EXPECT(kh_pointer_domain(d.as_ptr(), &dom, &dev) == KH_OK && dom == KH_DOMAIN_DEVICE && dev == 0);
        EXPECT(kh_pointer_domain(pin.as_ptr(), &dom, &dev) == KH_OK && dom == KH_DOMAIN_HOST_PINNED);
        EXPECT(kh_pointer_domain(uni.as_ptr(), &dom, &dev) == KH_OK && dom == KH_DOMAIN_UNIFIED);
        EXPECT(d.len_bytes() == 4096 && d.stream() && d.stream()->same_as(s) && uni.stream() && pin.stream() == nullptr);
        // zero-filled on the stream: device -> pinned copy reads zeros; then a pinned -> device -> managed round trip
        uint32_t* hp = static_cast<uint32_t*>(pin.as_ptr());
        for (int i = 0; i < 1024; ++i) hp[i] = 0xdeadbeefu;
        EXPECT(kh_memcpy_d2h_async(hp, d.as_ptr(), 4096, s.handle()) == KH_OK);
        s.synchronize();
        bool zero = true;
        for (int i = 0; i < 1024; ++i) zero = zero && hp[i] == 0;
        EXPECT(zero);
        for (int i = 0; i < 1024; ++i) hp[i] = 3u * i + 1u;
        EXPECT(kh_memcpy_h2d_async(d.as_ptr(), hp, 4096, s.handle()) == KH_OK);
        EXPECT(kh_memcpy_d2d_async(uni.as_ptr(), d.as_ptr(), 4096, s.handle()) == KH_OK);
        s.synchronize();
        bool same = true;
        for (int i = 0; i < 1024; ++i) same = same && static_cast<const uint32_t*>(uni.as_ptr())[i] == 3u * i + 1u;   // managed memory read by the host
        EXPECT(same);
        MemoryResource raw = HipAllocator(s, false).allocate(Layout(0, 16));   // uninit, zero-size
        EXPECT(raw.len_bytes() == 0);
    }
    // gray [0,128,255],[128,0,128] -> [104, 53]  (P/color/gray/mod.rs:395-412)
    auto rgb = Image<uint8_t, 3>::from_size_vec({2, 1}, {0, 128, 255, 128, 0, 128}).to_hip(s);
    auto gray = Image<uint8_t, 1>::zeros_hip({2, 1}, s);
    imgproc::gray_from_rgb(rgb, gray);
    auto g = gray.to_host();
    EXPECT(g.as_slice()[0] == 104 && g.as_slice()[1] == 53);
    EXPECT(throws(K::UnsupportedDevice, [&] { (void)gray.as_slice(); }));
    // resize smoke (P/resize/mod.rs:447-490): 4x3x3 ramp -> 3x2, expected 18 values
    std::vector<float> ramp(36);
    for (int i = 0; i < 36; ++i) ramp[i] = (float)i;
    auto src = Image<float, 3>::from_size_vec({3, 4}, ramp).to_hip(s);
    auto dst = Image<float, 3>::zeros_hip({2, 3}, other);  // allocated + zero-filled on ANOTHER stream: must be fenced in
    imgproc::resize(src, dst, InterpolationMode::Bilinear);
    const float want[18] = {2.25f, 3.25f, 4.25f, 6.75f, 7.75f, 8.75f, 14.25f, 15.25f, 16.25f, 18.75f, 19.75f, 20.75f, 26.25f, 27.25f, 28.25f,
                            30.75f, 31.75f, 32.75f};
    // no s.synchronize() here: the DeviceExec fenced the launch stream back into dst's stream, and to_host() drains that one
    auto out = dst.to_host();
    for (int i = 0; i < 18; ++i) EXPECT(std::fabs(out.as_slice()[i] - want[i]) < 1e-4f);
    // managed memory (MemoryDomain::Unified, I/cuda.rs:144-160): host writes reach the kernel, its result reaches the host view
    {
        auto uni = Image<uint8_t, 3>::zeros_hip_unified({2, 1}, s);
        EXPECT(uni.domain() == MemoryDomain::Unified && uni.is_device() && uni.is_unified() && uni.is_host_accessible());
        int32_t dom = -1, dev = -1;
        EXPECT(kh_pointer_domain(uni.device_ptr(), &dom, &dev) == KH_OK && dom == KH_DOMAIN_UNIFIED && dev == 0);
        const uint8_t px[6] = {0, 128, 255, 128, 0, 128};
        std::memcpy(uni.unified_data(), px, 6);
        auto ugray = Image<uint8_t, 1>::zeros_hip_unified({2, 1}, other);
        imgproc::gray_from_rgb(uni, ugray);
        const uint8_t* r = ugray.unified_data();  // drains ugray's stream, which was fenced behind the launch
        EXPECT(r[0] == 104 && r[1] == 53);
        auto up = Image<uint8_t, 3>::from_size_vec({2, 1}, {0, 128, 255, 128, 0, 128}).to_hip_unified(s);
        EXPECT(up.is_unified() && std::memcmp(up.unified_data(), px, 6) == 0);
        EXPECT(throws(K::UnsupportedDevice, [&] { (void)up.to_hip_unified(s); }));
        EXPECT(throws(K::UnsupportedDevice, [&] { (void)rgb.unified_data(); }));
    }
    // mixed residency and singular homography are typed errors (P/warp/cuda.rs:369-400, P/warp/perspective.rs:41-60)
    auto host_dst = Image<float, 3>::from_size_val({2, 3}, 0.0f);
    EXPECT(throws(K::MixedResidency, [&] { imgproc::resize(src, host_dst, InterpolationMode::Bilinear); }));
    auto same = Image<float, 3>::zeros_hip({3, 4}, s);
    EXPECT(throws(K::CannotComputeDeterminant, [&] { imgproc::warp_perspective(src, same, {1, 2, 3, 2, 4, 6, 3, 6, 9}, InterpolationMode::Bilinear); }));
    // horizontal flip through warp_affine lands every column (P/warp/affine.rs:471-495)
    auto row = Image<float, 1>::from_size_vec({4, 2}, {1, 2, 3, 4, 5, 6, 7, 8}).to_hip(s);
    auto flipped = Image<float, 1>::zeros_hip({4, 2}, s);
    imgproc::warp_affine(row, flipped, {-1, 0, 3, 0, 1, 0}, InterpolationMode::Nearest);
    auto f = flipped.to_host();
    const float wf[8] = {4, 3, 2, 1, 8, 7, 6, 5};
    for (int i = 0; i < 8; ++i) EXPECT(f.as_slice()[i] == wf[i]);
    // fused preprocess: a solid NV12 frame (Y=235,U=V=128 -> white) stretched, unit normalisation -> all 1.0 (P/preprocess.rs:1429)
    const int w = 16, h = 8;
    std::vector<uint8_t> nv(w * h * 3 / 2, 128);
    std::memset(nv.data(), 235, w * h);
    auto frame = Image<uint8_t, 1>::from_size_vec({(size_t)w * h * 3 / 2, 1}, nv).to_hip(s);
    auto chw = Image<float, 1>::zeros_hip({(size_t)3 * 12 * 6, 1}, s);
    Preprocessor pre(s, ResizeMode::Stretch, SourceFormat::Nv12);
    pre.run_raw(frame.device_ptr(), nv.size(), w, h, chw.device_ptr_mut(), 12, 6);
    auto t = chw.to_host();
    for (float v : t.as_slice()) EXPECT(std::fabs(v - 1.0f) < 1e-6f);
    EXPECT(throws(K::InvalidImageSize, [&] { pre.run_raw(frame.device_ptr(), nv.size() - 1, w, h, chw.device_ptr_mut(), 12, 6); }));
}

// Device affinity: everything on DEVICE 1 from a thread whose current device stays 0 — the operators bind the launch
// stream's device for the call (detail::DeviceExec) and restore the caller's.  Runs where two devices exist: the host simulator with
// KH_HOSTSIM_DEVICES=2 (tests/test_zz_hostsim.py) or a multi-GPU node.
static void on_second_device() {
    int32_t n = 0;
    if (kh_device_count(&n) != KH_OK || n < 2) { std::printf("second device: skipped (%d device(s))\n", (int)n); return; }
    EXPECT(kh_set_device(0) == KH_OK);
    Stream s1 = Stream::create(1), other1 = Stream::create(1);
    auto current = [] { int32_t d = -1; kh_get_device(&d); return d; };
    auto rgb = Image<uint8_t, 3>::from_size_vec({2, 1}, {0, 128, 255, 128, 0, 128}).to_hip(s1);
    auto gray = Image<uint8_t, 1>::zeros_hip({2, 1}, other1);   // another stream of device 1: fenced in and back
    int32_t dom = -1, dev = -1;
    EXPECT(kh_pointer_domain(rgb.device_ptr(), &dom, &dev) == KH_OK && dom == KH_DOMAIN_DEVICE && dev == 1);
    imgproc::gray_from_rgb(rgb, gray);
    EXPECT(current() == 0);
    auto g = gray.to_host();
    EXPECT(g.as_slice()[0] == 104 && g.as_slice()[1] == 53);
    // an operator with stream-ordered scratch and a cached table (Lanczos resize): both are filed under the CURRENT device
    std::vector<float> ramp(64 * 48 * 3);
    for (size_t i = 0; i < ramp.size(); ++i) ramp[i] = (float)((i * 7 + 13) % 251) / 255.0f;
    auto src0 = Image<float, 3>::from_size_vec({64, 48}, ramp).to_hip(Stream::create(0));
    auto src1 = Image<float, 3>::from_size_vec({64, 48}, ramp).to_hip(s1);
    auto d0 = Image<float, 3>::zeros_hip({23, 17}, *src0.stream());
    auto d1 = Image<float, 3>::zeros_hip({23, 17}, s1);
    imgproc::resize(src0, d0, InterpolationMode::Lanczos);
    imgproc::resize(src1, d1, InterpolationMode::Lanczos);
    EXPECT(current() == 0);
    auto h0 = d0.to_host(), h1 = d1.to_host();
    EXPECT(std::memcmp(h0.as_slice().data(), h1.as_slice().data(), h0.as_slice().size() * sizeof(float)) == 0);
    // operands on different devices are a typed error, never a peer access (P/cuda/dispatch.rs:51-53)
    EXPECT(throws(K::DeviceMismatch, [&] { imgproc::resize(src0, d1, InterpolationMode::Bilinear); }));
    EXPECT(current() == 0);
    // the fused preprocess on device 1
    const int w = 16, h = 8;
    std::vector<uint8_t> nv(w * h * 3 / 2, 128);
    std::memset(nv.data(), 235, w * h);
    auto frame = Image<uint8_t, 1>::from_size_vec({(size_t)w * h * 3 / 2, 1}, nv).to_hip(s1);
    auto chw = Image<float, 1>::zeros_hip({(size_t)3 * 12 * 6, 1}, s1);
    Preprocessor pre(s1, ResizeMode::Stretch, SourceFormat::Nv12);
    pre.run_raw(frame.device_ptr(), nv.size(), w, h, chw.device_ptr_mut(), 12, 6);
    EXPECT(current() == 0);
    auto t = chw.to_host();
    for (float v : t.as_slice()) EXPECT(std::fabs(v - 1.0f) < 1e-6f);
    std::printf("second device: checked\n");
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "host";
    host_only();
    if (mode == "gpu") { on_device(); on_second_device(); }
    std::printf("%s: %d failure(s) [%s]\n", mode.c_str(), failures, kh_version());
    return failures ? 1 : 0;
}
