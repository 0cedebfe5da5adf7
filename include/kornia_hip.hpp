// kornia_hip.hpp — header-only C++17 host mirror of the reference's Rust API for the imgproc hot path,
// layered on the C ABI of include/kornia_hip.h (link with -lkornia_hip).
//
// The reference is compiled Rust (crates/kornia-tensor, kornia-image, kornia-imgproc); this header keeps its
// host-side contract for non-Python hosts: `Image<T, C>` whose residency (Host / Device) is a run-time
// property of the storage (I/image.rs:138, T/storage.rs:53, T/resource.rs:19), explicit transfers only
// (`to_hip` / `to_host`, I/cuda.rs:53-221), host access to device memory is an error (T/storage.rs:102-110),
// and every operator classifies its operands first (pair_residency, P/cuda/dispatch.rs:105-130): mixed
// host/device pairs, different devices and unsupported dtype / channel combinations are TYPED errors, never
// a silent transfer or a CPU fallback.  A host/host pair has no implementation here — this build is the
// device backend only — and says so.
//
//   kornia::Stream s = kornia::Stream::create(0);
//   auto img  = kornia::Image<float, 3>::from_size_val({1920, 1080}, 0.5f);
//   auto dimg = img.to_hip(s);
//   auto out  = kornia::Image<float, 3>::zeros_hip({224, 224}, s);
//   kornia::imgproc::resize(dimg, out, kornia::InterpolationMode::Bilinear);
//   auto host = out.to_host();
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "kornia_hip.h"

namespace kornia {

// ---- errors (I/error.rs:3-80, P/cuda/dispatch.rs:166-211) ------------------------------------------------
class ImageError : public std::runtime_error {
public:
    enum class Kind {
        InvalidImageSize, InvalidChannelShape, MixedResidency, DeviceMismatch, UnsupportedDevice, HostPathUnavailable,
        NoDeviceKernel, CannotComputeDeterminant, InvalidSigmaValue, InvalidNormalize, DimensionsTooLarge, SliceTooSmall, Hip
    };
    ImageError(Kind k, const std::string& what) : std::runtime_error(what), kind(k) {}
    Kind kind;
};

namespace detail {
inline std::string last_error() {
    char buf[512];
    kh_last_error(buf, sizeof buf);
    return buf;
}
// status codes -> the reference's typed errors
inline void check(int32_t rc) {
    if (rc == KH_OK) return;
    using K = ImageError::Kind;
    K k = K::Hip;
    switch (rc) {
        case KH_ERR_INVALID_ARG: k = K::InvalidImageSize; break;
        case KH_ERR_UNSUPPORTED: k = K::NoDeviceKernel; break;
        case KH_ERR_TOO_LARGE: k = K::DimensionsTooLarge; break;
        case KH_ERR_SINGULAR: k = K::CannotComputeDeterminant; break;
        case KH_ERR_SLICE_TOO_SMALL: k = K::SliceTooSmall; break;
        default: break;
    }
    throw ImageError(k, last_error());
}
}  // namespace detail

// ---- streams / events (T/cuda.rs cuda_stream, PY/cuda_ext/mod.rs:61-120) ----------------------------------
class Stream {
public:
    // a new non-blocking stream on `device`
    static Stream create(int device = 0) {
        int prev = 0;
        detail::check(kh_get_device(&prev));
        detail::check(kh_set_device(device));
        kh_stream_t h = nullptr;
        const int32_t rc = kh_stream_create(&h);
        kh_set_device(prev);
        detail::check(rc);
        return Stream(std::shared_ptr<Handle>(new Handle{h, device, true}));
    }
    // borrow a caller-owned hipStream_t (never destroyed here)
    static Stream from_handle(void* hip_stream, int device) {
        return Stream(std::shared_ptr<Handle>(new Handle{static_cast<kh_stream_t>(hip_stream), device, false}));
    }
    void synchronize() const { detail::check(kh_stream_synchronize(h_->h)); }
    kh_stream_t handle() const { return h_->h; }
    int device() const { return h_->device; }
    bool same_as(const Stream& o) const { return h_->h == o.h_->h; }

private:
    struct Handle {
        kh_stream_t h;
        int device;
        bool owned;
        ~Handle() { if (owned && h) kh_stream_destroy(h); }
    };
    explicit Stream(std::shared_ptr<Handle> h) : h_(std::move(h)) {}
    std::shared_ptr<Handle> h_;
};

// ---- storage: residency is a run-time property (T/resource.rs:19-60) --------------------------------------
// Host / Device / Unified — where the bytes can be legally dereferenced (MemoryDomain, T/resource.rs:19-60).  Unified = managed
// memory (kh_malloc_managed): host slices AND device kernels work on the same allocation; dispatch treats it as device-resident.
enum class MemoryDomain { Host, Device, Unified };

// ---- allocators (T/allocator.rs:20-146; T/cuda.rs:214-262, 355-380, 440-511) ------------------------------
// `TensorAllocator` turns a `Layout` (size + power-of-two alignment, std::alloc::Layout) into an owning `MemoryResource`
// (as_ptr / len_bytes / domain, T/resource.rs:73-101).  CpuAllocator + host_alloc() are the reference's host side; HipAllocator,
// PinnedAllocator and HipUnifiedAllocator replace CudaAllocator / PinnedAllocator / CudaUnifiedAllocator over the C ABI
// (kh_malloc_async, kh_host_alloc, kh_malloc_managed).  Allocations are zero-filled unless the allocator says otherwise;
// a zero-size layout is legal and owns nothing.
class TensorAllocatorError : public std::runtime_error {
public:
    enum class Kind { LayoutError, NullPointer, CannotAllocateForeign };
    TensorAllocatorError(Kind k, const std::string& msg) : std::runtime_error(msg), kind_(k) {}
    Kind kind() const { return kind_; }

private:
    Kind kind_;
};

struct Layout {
    size_t size, align;
    Layout(size_t size_, size_t align_ = 1) : size(size_), align(align_) {
        if (align == 0 || (align & (align - 1)) || size > (static_cast<size_t>(1) << 63) - align)
            throw TensorAllocatorError(TensorAllocatorError::Kind::LayoutError,
                                       "invalid layout (size " + std::to_string(size) + ", align " + std::to_string(align) + ")");
    }
    template <typename T>
    static Layout array(size_t count) { return Layout(sizeof(T) * count, alignof(T)); }   // Layout::array::<T>(n)
};

// An owning, move-only block of memory.  Dropping it returns the block to where it came from (the stream-ordered pool on its
// stream, the pinned / managed heap after draining its stream, the process heap).
class MemoryResource {
public:
    MemoryResource(const MemoryResource&) = delete;
    MemoryResource& operator=(const MemoryResource&) = delete;
    MemoryResource(MemoryResource&& o) noexcept : ptr_(o.ptr_), bytes_(o.bytes_), domain_(o.domain_), how_(o.how_), stream_(std::move(o.stream_)) { o.ptr_ = nullptr; }
    MemoryResource& operator=(MemoryResource&& o) noexcept {
        if (this != &o) { release(); ptr_ = o.ptr_; bytes_ = o.bytes_; domain_ = o.domain_; how_ = o.how_; stream_ = std::move(o.stream_); o.ptr_ = nullptr; }
        return *this;
    }
    ~MemoryResource() { release(); }
    void* as_ptr() const { return ptr_; }
    size_t len_bytes() const { return bytes_; }
    MemoryDomain domain() const { return domain_; }
    bool is_readonly() const { return false; }
    const Stream* stream() const { return stream_.get(); }   // Device / Unified: the stream the block is ordered on

private:
    friend class CpuAllocator;
    friend class PinnedAllocator;
    friend class HipAllocator;
    friend class HipUnifiedAllocator;
    enum class How { Heap, Pinned, Pool, Managed };
    MemoryResource(void* p, size_t n, MemoryDomain d, How h, std::unique_ptr<Stream> s) : ptr_(p), bytes_(n), domain_(d), how_(h), stream_(std::move(s)) {}
    void release() {
        if (!ptr_) return;
        switch (how_) {
            case How::Heap: std::free(ptr_); break;
            case How::Pinned: kh_host_free(ptr_); break;
            case How::Pool: kh_free_async(ptr_, stream_ ? stream_->handle() : nullptr); break;
            case How::Managed:
                if (stream_) kh_stream_synchronize(stream_->handle());
                kh_free(ptr_);
                break;
        }
        ptr_ = nullptr;
    }
    void* ptr_ = nullptr;
    size_t bytes_ = 0;
    MemoryDomain domain_ = MemoryDomain::Host;
    How how_ = How::Heap;
    std::unique_ptr<Stream> stream_;
};

class TensorAllocator {   // trait TensorAllocator (T/allocator.rs:73-90)
public:
    virtual ~TensorAllocator() = default;
    virtual MemoryResource allocate(const Layout& layout) const = 0;
    virtual MemoryDomain domain() const = 0;
};

class CpuAllocator final : public TensorAllocator {   // zeroed, aligned process-heap memory (T/allocator.rs:107-130)
public:
    MemoryResource allocate(const Layout& l) const override {
        if (l.size == 0) return MemoryResource(nullptr, 0, MemoryDomain::Host, MemoryResource::How::Heap, nullptr);
        const size_t al = l.align < sizeof(void*) ? sizeof(void*) : l.align;
        void* p = std::aligned_alloc(al, (l.size + al - 1) / al * al);
        if (!p) throw TensorAllocatorError(TensorAllocatorError::Kind::NullPointer, "the host allocator returned null");
        std::memset(p, 0, l.size);
        return MemoryResource(p, l.size, MemoryDomain::Host, MemoryResource::How::Heap, nullptr);
    }
    MemoryDomain domain() const override { return MemoryDomain::Host; }
};
inline const CpuAllocator& host_alloc() {   // the process-global host allocator handle (T/allocator.rs:136-146)
    static const CpuAllocator a;
    return a;
}

class PinnedAllocator final : public TensorAllocator {   // zeroed page-locked host memory (T/cuda.rs:355-380)
public:
    MemoryResource allocate(const Layout& l) const override {
        if (l.size == 0) return MemoryResource(nullptr, 0, MemoryDomain::Host, MemoryResource::How::Pinned, nullptr);
        void* p = nullptr;
        detail::check(kh_host_alloc(&p, l.size));
        if (!p) throw TensorAllocatorError(TensorAllocatorError::Kind::NullPointer, "kh_host_alloc returned null");
        std::memset(p, 0, l.size);
        return MemoryResource(p, l.size, MemoryDomain::Host, MemoryResource::How::Pinned, nullptr);
    }
    MemoryDomain domain() const override { return MemoryDomain::Host; }
};

// replaces CudaAllocator (T/cuda.rs:214-262): stream-ordered pool memory on `stream`'s device, zero-filled on that stream
// (zeroed = false: uninit_cuda — the producer must overwrite every byte)
class HipAllocator final : public TensorAllocator {
public:
    explicit HipAllocator(const Stream& stream, bool zeroed = true) : stream_(stream), zeroed_(zeroed) {}
    MemoryResource allocate(const Layout& l) const override {
        void* p = nullptr;
        int prev = 0;
        detail::check(kh_get_device(&prev));
        detail::check(kh_set_device(stream_.device()));
        const int32_t rc = kh_malloc_async(&p, l.size, zeroed_ ? 1 : 0, stream_.handle());
        kh_set_device(prev);
        detail::check(rc);
        if (l.size && !p) throw TensorAllocatorError(TensorAllocatorError::Kind::NullPointer, "kh_malloc_async returned null");
        if (l.size && reinterpret_cast<uintptr_t>(p) % l.align) {
            kh_free_async(p, stream_.handle());
            throw TensorAllocatorError(TensorAllocatorError::Kind::LayoutError, "the device pool returned a pointer not aligned to " + std::to_string(l.align));
        }
        return MemoryResource(p, l.size, MemoryDomain::Device, MemoryResource::How::Pool, std::unique_ptr<Stream>(new Stream(stream_)));
    }
    MemoryDomain domain() const override { return MemoryDomain::Device; }

private:
    Stream stream_;
    bool zeroed_;
};

class HipUnifiedAllocator final : public TensorAllocator {   // managed memory carrying `stream` (CudaUnifiedAllocator, T/cuda.rs:440-511)
public:
    explicit HipUnifiedAllocator(const Stream& stream) : stream_(stream) {}
    MemoryResource allocate(const Layout& l) const override {
        void* p = nullptr;
        int prev = 0;
        detail::check(kh_get_device(&prev));
        detail::check(kh_set_device(stream_.device()));
        const int32_t rc = kh_malloc_managed(&p, l.size ? l.size : 1);   // the driver rejects a zero-size request
        kh_set_device(prev);
        detail::check(rc);
        if (!p) throw TensorAllocatorError(TensorAllocatorError::Kind::NullPointer, "kh_malloc_managed returned null");
        if (l.size) std::memset(p, 0, l.size);   // managed memory is host-writable
        return MemoryResource(p, l.size, MemoryDomain::Unified, MemoryResource::How::Managed, std::unique_ptr<Stream>(new Stream(stream_)));
    }
    MemoryDomain domain() const override { return MemoryDomain::Unified; }

private:
    Stream stream_;
};

struct ImageSize {
    size_t width, height;
    bool operator==(const ImageSize& o) const { return width == o.width && height == o.height; }
    bool operator!=(const ImageSize& o) const { return !(*this == o); }
};

enum class InterpolationMode { Nearest = KH_INTERP_NEAREST, Bilinear = KH_INTERP_BILINEAR, Bicubic = KH_INTERP_BICUBIC,
                               Lanczos = KH_INTERP_LANCZOS };

namespace detail {
template <typename T>
struct Storage {
    MemoryDomain domain = MemoryDomain::Host;
    std::vector<T> host;            // Host
    T* dev = nullptr;               // Device (stream-ordered allocation, freed on its stream) or Unified (managed, kh_free)
    size_t len = 0;
    std::unique_ptr<Stream> stream; // Device / Unified
    Storage() = default;
    Storage(const Storage&) = delete;
    Storage& operator=(const Storage&) = delete;
    ~Storage() {
        if (!dev) return;
        if (domain == MemoryDomain::Unified) {  // Backing::Managed (T/cuda.rs:139-169): drain the carried stream, then hipFree
            if (stream) kh_stream_synchronize(stream->handle());
            kh_free(dev);
        } else {
            kh_free_async(dev, stream ? stream->handle() : nullptr);
        }
    }
};
}  // namespace detail

// ---- Image<T, C> (I/image.rs:138-420, I/cuda.rs:53-221) ----------------------------------------------------
template <typename T, int C>
class Image {
    static_assert(C >= 1 && C <= 4, "1..4 channels");

public:
    static Image from_size_vec(ImageSize size, std::vector<T> data) {
        if (data.size() != size.width * size.height * C)
            throw ImageError(ImageError::Kind::InvalidChannelShape,
                             "data length " + std::to_string(data.size()) + " does not match the image shape " +
                                 std::to_string(size.height) + "x" + std::to_string(size.width) + "x" + std::to_string(C));
        Image img(size);
        img.s_->host = std::move(data);
        img.s_->len = img.s_->host.size();
        return img;
    }
    static Image from_size_val(ImageSize size, T val) { return from_size_vec(size, std::vector<T>(size.width * size.height * C, val)); }

    // zeros_cuda / uninit_cuda (I/cuda.rs:96-121): device-resident, allocated on `stream`
    static Image zeros_hip(ImageSize size, const Stream& stream) { return alloc_hip(size, stream, true); }
    static Image uninit_hip(ImageSize size, const Stream& stream) { return alloc_hip(size, stream, false); }
    // zeros_cuda_unified (I/cuda.rs:144-160): zero-filled managed memory carrying `stream`
    static Image zeros_hip_unified(ImageSize size, const Stream& stream) {
        Image img(size);
        img.s_->domain = MemoryDomain::Unified;
        img.s_->stream.reset(new Stream(stream));
        img.s_->len = size.width * size.height * C;
        void* p = nullptr;
        int prev = 0;
        detail::check(kh_get_device(&prev));
        detail::check(kh_set_device(stream.device()));
        const size_t bytes = img.s_->len * sizeof(T);
        const int32_t rc = kh_malloc_managed(&p, bytes ? bytes : 1);  // the driver rejects a zero-size request
        kh_set_device(prev);
        detail::check(rc);
        img.s_->dev = static_cast<T*>(p);
        return img;
    }

    ImageSize size() const { return size_; }
    size_t width() const { return size_.width; }
    size_t height() const { return size_.height; }
    size_t cols() const { return size_.width; }
    size_t rows() const { return size_.height; }
    static constexpr int num_channels() { return C; }
    size_t numel() const { return size_.width * size_.height * C; }
    MemoryDomain domain() const { return s_->domain; }
    // device- OR unified-resident: what residency dispatch asks (is_device, P/cuda/dispatch.rs:90-96)
    bool is_device() const { return s_->domain != MemoryDomain::Host; }
    bool is_unified() const { return s_->domain == MemoryDomain::Unified; }
    bool is_host_accessible() const { return s_->domain != MemoryDomain::Device; }
    // the stream a device image is ordered on (TensorStorage::cuda_stream, T/cuda.rs:1010); nullptr for host
    const Stream* stream() const { return s_->stream.get(); }

    // host access only — refuses device memory like TensorStorage::as_slice (T/storage.rs:102-110)
    const std::vector<T>& as_slice() const {
        if (is_device()) throw ImageError(ImageError::Kind::UnsupportedDevice, is_unified() ? "managed image: use unified_data() (a pointer, not a vector)"
                                                                                           : "host access to device-resident image data; call to_host() first");
        return s_->host;
    }
    std::vector<T>& as_slice_mut() { return const_cast<std::vector<T>&>(static_cast<const Image&>(*this).as_slice()); }
    // host view of a MANAGED image (as_slice on MemoryDomain::Unified, T/storage.rs:102-125): the carried stream is drained
    // first, so kernels that wrote the image have finished.  numel() elements.
    T* unified_data() {
        if (!is_unified()) throw ImageError(ImageError::Kind::UnsupportedDevice, "unified_data: the image is not in managed memory");
        s_->stream->synchronize();
        return s_->dev;
    }
    const T* unified_data() const { return const_cast<Image*>(this)->unified_data(); }
    // raw device pointer (as_cudaslice, I/cuda.rs:200-221); host images have none
    const T* device_ptr() const {
        if (!is_device()) throw ImageError(ImageError::Kind::UnsupportedDevice, "device pointer of a host-resident image; call to_hip(stream) first");
        return s_->dev;
    }
    T* device_ptr_mut() { return const_cast<T*>(static_cast<const Image&>(*this).device_ptr()); }

    // explicit transfers (to_cuda, to_host_owned: I/cuda.rs:53-95, 122-160)
    Image to_hip(const Stream& stream) const {
        if (is_device()) throw ImageError(ImageError::Kind::UnsupportedDevice, "to_hip: the image is already device-resident");
        Image out = alloc_hip(size_, stream, false);
        if (numel()) {
            detail::check(kh_memcpy_h2d_async(out.s_->dev, s_->host.data(), numel() * sizeof(T), stream.handle()));
            stream.synchronize();  // the pageable source may be released by the caller right after
        }
        return out;
    }
    // to_cuda_unified (I/cuda.rs:62-75): copy a HOST image into a new managed image carrying `stream`
    Image to_hip_unified(const Stream& stream) const {
        if (is_device()) throw ImageError(ImageError::Kind::UnsupportedDevice, "to_hip_unified: the image is not host-resident");
        Image out = zeros_hip_unified(size_, stream);
        if (numel()) std::copy(s_->host.begin(), s_->host.end(), out.s_->dev);  // managed memory is host-writable
        return out;
    }
    Image to_host() const {
        if (!is_device()) throw ImageError(ImageError::Kind::UnsupportedDevice, "to_host: the image is already host-resident");
        Image out(size_);
        out.s_->host.resize(numel());
        out.s_->len = numel();
        if (numel()) {
            s_->stream->synchronize();
            detail::check(kh_memcpy_d2h_async(out.s_->host.data(), s_->dev, numel() * sizeof(T), s_->stream->handle()));
            s_->stream->synchronize();
        }
        return out;
    }

private:
    explicit Image(ImageSize size) : size_(size), s_(std::make_shared<detail::Storage<T>>()) {}
    static Image alloc_hip(ImageSize size, const Stream& stream, bool zeroed) {
        Image img(size);
        img.s_->domain = MemoryDomain::Device;
        img.s_->stream.reset(new Stream(stream));
        img.s_->len = size.width * size.height * C;
        void* p = nullptr;
        int prev = 0;
        detail::check(kh_get_device(&prev));
        detail::check(kh_set_device(stream.device()));
        const int32_t rc = kh_malloc_async(&p, img.s_->len * sizeof(T), zeroed ? 1 : 0, stream.handle());
        kh_set_device(prev);
        detail::check(rc);
        img.s_->dev = static_cast<T*>(p);
        return img;
    }
    ImageSize size_;
    std::shared_ptr<detail::Storage<T>> s_;
};

// ---- residency dispatch (pair_residency / DeviceExec::for_streams, P/cuda/dispatch.rs:50-130) ---------------
namespace detail {
// `device` current for the scope, the caller's restored at its end: for the entries that take a stream and raw device pointers
// (camera-format decoders / encoders, the preprocessor, graph capture and replay) — the operators on Images bind through DeviceExec.
class DeviceScope {
public:
    explicit DeviceScope(int device) {   // never throws: without a usable device the entry that follows reports its own typed error
        int32_t cur = -1;
        if (kh_get_device(&cur) == KH_OK && cur != device && kh_set_device(device) == KH_OK) prev_ = cur;
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
    ~DeviceScope() { if (prev_ >= 0) kh_set_device(prev_); }

private:
    int prev_ = -1;
};
// DeviceExec (P/cuda/dispatch.rs:28-82): the stream an op launches on — the SOURCE image's — with the destination's
// stream fenced IN before the launch (for_streams) and the launch stream fenced BACK into it when the exec goes out of
// scope, i.e. right after the launch in every operator below (run): dst's own stream — its to_host(), the next op that
// reads it, its stream-ordered free — is ordered after the kernel that writes it.  Same-stream pairs cost nothing.
class DeviceExec {
public:
    // The exec also BINDS the launch stream's device for its lifetime (the reference binds its context per call,
    // `ctx.bind_to_thread()`): the C ABI keeps HIP's rule that the caller selects a stream's device before launching on it — the
    // table caches, the workspace registry and a NULL stream handle all mean "the current device" — so an operator on device-1
    // images works from a thread whose current device is 0, and the caller's device is restored when the exec goes out of scope.
    DeviceExec(const Stream& launch, const Stream& dst) : launch_(launch), dst_(dst), cross_(!launch.same_as(dst)) {
        int32_t cur = -1;
        if (kh_get_device(&cur) == KH_OK && cur != launch_.device() && kh_set_device(launch_.device()) == KH_OK) prev_device_ = cur;
        if (cross_) {
            const int32_t rc = kh_stream_fence(dst_.handle(), launch_.handle());  // dst's pending work first
            if (rc != 0) { restore(); check(rc); }
        }
    }
    DeviceExec(const DeviceExec&) = delete;
    DeviceExec& operator=(const DeviceExec&) = delete;
    DeviceExec(DeviceExec&& o) noexcept : launch_(o.launch_), dst_(o.dst_), cross_(o.cross_), prev_device_(o.prev_device_) {
        o.cross_ = false;
        o.prev_device_ = -1;
    }
    ~DeviceExec() {
        if (cross_) kh_stream_fence(launch_.handle(), dst_.handle());  // best effort in a destructor; a failed launch has already thrown
        restore();
    }
    kh_stream_t handle() const { return launch_.handle(); }
    const Stream& stream() const { return launch_; }
    operator const Stream&() const { return launch_; }

private:
    void restore() noexcept {
        if (prev_device_ >= 0) kh_set_device(prev_device_);
        prev_device_ = -1;
    }
    Stream launch_, dst_;
    bool cross_;
    int prev_device_ = -1;
};
template <typename TS, int CS, typename TD, int CD>
inline DeviceExec device_exec_for(const Image<TS, CS>& src, const Image<TD, CD>& dst, const char* what) {
    if (src.is_device() != dst.is_device())
        throw ImageError(ImageError::Kind::MixedResidency, std::string(what) + ": src and dst must both be host-resident or both device-resident "
                                                                                  "(no implicit transfers)");
    if (!src.is_device())
        throw ImageError(ImageError::Kind::HostPathUnavailable, std::string(what) + ": host images — this build provides the HIP device backend only; "
                                                                                       "move the images with to_hip(stream)");
    const Stream& ss = *src.stream();
    const Stream& ds = *dst.stream();
    if (ss.device() != ds.device())
        throw ImageError(ImageError::Kind::DeviceMismatch, std::string(what) + ": src is on device " + std::to_string(ss.device()) +
                                                               ", dst on device " + std::to_string(ds.device()));
    return DeviceExec(ss, ds);
}
template <typename T, int C>
inline void same_size(const Image<T, C>& a, const Image<T, C>& b, const char* what) {
    if (a.size() != b.size())
        throw ImageError(ImageError::Kind::InvalidImageSize, std::string(what) + ": image sizes differ: " + std::to_string(a.width()) + "x" +
                                                                 std::to_string(a.height()) + " vs " + std::to_string(b.width()) + "x" + std::to_string(b.height()));
}
inline int32_t i32(size_t v) { return static_cast<int32_t>(v); }
}  // namespace detail

// ---- operators: one call per reference launcher (names and argument meaning of kornia_imgproc) --------------
namespace imgproc {

// color::gray_from_rgb (P/color/gray/mod.rs:104-147)
inline void gray_from_rgb(const Image<uint8_t, 3>& src, Image<uint8_t, 1>& dst) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "gray_from_rgb");
    if (src.size() != dst.size()) throw ImageError(ImageError::Kind::InvalidImageSize, "gray_from_rgb: image sizes differ");
    detail::check(kh_gray_from_rgb_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), (int64_t)(src.width() * src.height())));
}
inline void gray_from_rgb(const Image<float, 3>& src, Image<float, 1>& dst) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "gray_from_rgb");
    if (src.size() != dst.size()) throw ImageError(ImageError::Kind::InvalidImageSize, "gray_from_rgb: image sizes differ");
    detail::check(kh_gray_from_rgb_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), (int64_t)(src.width() * src.height())));
}

// resize::resize (P/resize/mod.rs:114-238); f32, C in {1, 3, 4}
template <int C>
inline void resize(const Image<float, C>& src, Image<float, C>& dst, InterpolationMode interpolation) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "resize");
    detail::check(kh_resize_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()),
                                detail::i32(dst.width()), detail::i32(dst.height()), C, (int32_t)interpolation, 1, 0, 0));
}
// resize::resize_fast_u8_aa (P/resize/mod.rs:348)
template <int C>
inline void resize_fast(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, InterpolationMode interpolation, bool antialias = true) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "resize_fast");
    detail::check(kh_resize_fast_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()),
                                    detail::i32(dst.width()), detail::i32(dst.height()), C, (int32_t)interpolation, antialias ? 1 : 0, 1, 0, 0));
}

// filter::gaussian_blur / box_blur (P/filter/ops.rs:116, 39; u8: :639, :59)
template <int C>
inline void gaussian_blur(const Image<float, C>& src, Image<float, C>& dst, std::pair<int, int> kernel_size, std::pair<float, float> sigma) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "gaussian_blur");
    detail::same_size(src, dst, "gaussian_blur");
    detail::check(kh_gaussian_blur_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C,
                                       kernel_size.first, kernel_size.second, sigma.first, sigma.second, 1, 0, 0));
}
template <int C>
inline void gaussian_blur(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, std::pair<int, int> kernel_size, std::pair<float, float> sigma) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "gaussian_blur_u8");
    detail::same_size(src, dst, "gaussian_blur_u8");
    detail::check(kh_gaussian_blur_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C,
                                      kernel_size.first, kernel_size.second, sigma.first, sigma.second, 1, 0, 0));
}
template <int C>
inline void box_blur(const Image<float, C>& src, Image<float, C>& dst, std::pair<int, int> kernel_size) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "box_blur");
    detail::same_size(src, dst, "box_blur");
    detail::check(kh_box_blur_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C,
                                  kernel_size.first, kernel_size.second, 1, 0, 0));
}

// warp::warp_affine / warp_perspective (P/warp/affine.rs:123, P/warp/perspective.rs:115); m = FORWARD transform
template <int C>
inline void warp_affine(const Image<float, C>& src, Image<float, C>& dst, const std::array<float, 6>& m, InterpolationMode interpolation) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "warp_affine");
    detail::check(kh_warp_affine_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()),
                                     detail::i32(dst.width()), detail::i32(dst.height()), C, m.data(), (int32_t)interpolation, 1, 0, 0));
}
template <int C>
inline void warp_perspective(const Image<float, C>& src, Image<float, C>& dst, const std::array<float, 9>& m, InterpolationMode interpolation) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "warp_perspective");
    detail::check(kh_warp_perspective_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()),
                                          detail::i32(dst.width()), detail::i32(dst.height()), C, m.data(), (int32_t)interpolation, 1, 0, 0));
}
template <int C>
inline void warp_affine_u8(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, const std::array<float, 6>& m) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "warp_affine_u8");
    detail::check(kh_warp_affine_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()),
                                    detail::i32(dst.width()), detail::i32(dst.height()), C, m.data(), 1, 0, 0));
}

// interpolation::remap (P/interpolation/remap.rs:43): maps are single-channel f32 images of the destination size
template <int C>
inline void remap(const Image<float, C>& src, Image<float, C>& dst, const Image<float, 1>& map_x, const Image<float, 1>& map_y,
                  InterpolationMode interpolation) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "remap");
    if (map_x.size() != dst.size() || map_y.size() != dst.size())
        throw ImageError(ImageError::Kind::InvalidImageSize, "remap: map_x, map_y and dst must have the same size");
    if (!map_x.is_device() || !map_y.is_device())
        throw ImageError(ImageError::Kind::MixedResidency, "remap: map_x and map_y must be device-resident when src/dst are on the GPU");
    for (const Image<float, 1>* mp : {&map_x, &map_y})
        if (!mp->stream()->same_as(s.stream())) detail::check(kh_stream_fence(mp->stream()->handle(), s.handle()));
    detail::check(kh_remap_f32(s.handle(), src.device_ptr(), map_x.device_ptr(), map_y.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()),
                               detail::i32(src.height()), detail::i32(dst.width()), detail::i32(dst.height()), C, (int32_t)interpolation, 1, 0, 0));
}

// normalize::normalize_mean_std (P/normalize.rs:56)
template <int C>
inline void normalize_mean_std(const Image<float, C>& src, Image<float, C>& dst, const std::array<float, C>& mean, const std::array<float, C>& std_) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "normalize_mean_std");
    detail::same_size(src, dst, "normalize_mean_std");
    detail::check(kh_normalize_mean_std_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), (int64_t)(src.width() * src.height()), C, mean.data(),
                                            std_.data()));
}


// ---- the rest of the operator surface (same residency rules; every call cites the Rust function it mirrors) ------

namespace helpers {
template <typename TS, int CS, typename TD, int CD>
inline detail::DeviceExec map_pair(const Image<TS, CS>& src, const Image<TD, CD>& dst, const char* what) {
    detail::DeviceExec s = detail::device_exec_for(src, dst, what);
    if (src.size() != dst.size()) throw ImageError(ImageError::Kind::InvalidImageSize, std::string(what) + ": image sizes differ");
    return s;
}
template <typename T, int C>
inline int64_t npixels(const Image<T, C>& im) { return (int64_t)(im.width() * im.height()); }
}  // namespace helpers

// color::rgb_from_gray / bgr_from_rgb / rgba_from_rgb / bgra_from_rgb / rgb_from_rgba / rgb_from_bgra (P/color/gray/mod.rs:241,
// P/color/rgb/mod.rs:60-330)
inline void rgb_from_gray(const Image<uint8_t, 1>& src, Image<uint8_t, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "rgb_from_gray");  // classify operands BEFORE touching device pointers
    detail::check(kh_rgb_from_gray_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src)));
}
inline void rgb_from_gray(const Image<float, 1>& src, Image<float, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "rgb_from_gray");  // classify operands BEFORE touching device pointers
    detail::check(kh_rgb_from_gray_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src)));
}
inline void bgr_from_rgb(const Image<uint8_t, 3>& src, Image<uint8_t, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "bgr_from_rgb");  // classify operands BEFORE touching device pointers
    detail::check(kh_bgr_from_rgb_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src)));
}
inline void bgr_from_rgb(const Image<float, 3>& src, Image<float, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "bgr_from_rgb");  // classify operands BEFORE touching device pointers
    detail::check(kh_bgr_from_rgb_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src)));
}
inline void rgba_from_rgb(const Image<uint8_t, 3>& src, Image<uint8_t, 4>& dst, bool bgra = false) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "rgba_from_rgb");  // classify operands BEFORE touching device pointers
    detail::check(kh_rgba_from_rgb_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src), bgra ? 1 : 0));
}
inline void rgba_from_rgb(const Image<float, 3>& src, Image<float, 4>& dst, bool bgra = false) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "rgba_from_rgb");  // classify operands BEFORE touching device pointers
    detail::check(kh_rgba_from_rgb_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src), bgra ? 1 : 0));
}
// `background` = nullptr drops alpha, else blends over the RGB triple (rgb_from_rgba, P/color/rgb/mod.rs:60-126); `bgra` swaps R and B
inline void rgb_from_rgba(const Image<uint8_t, 4>& src, Image<uint8_t, 3>& dst, const std::array<uint8_t, 3>* background = nullptr, bool bgra = false) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "rgb_from_rgba");  // classify operands BEFORE touching device pointers
    detail::check(kh_rgb_from_rgba_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src),
                                      bgra ? 1 : 0, background ? background->data() : nullptr));
}

// color::ycbcr_from_rgb / yuv_from_rgb and inverses (P/color/yuv/mod.rs:150-185)
enum class ChromaOrder { YCrCb = KH_YCC_YCRCB, YuvCbCr = KH_YCC_YUV };
inline void ycc_from_rgb(const Image<uint8_t, 3>& src, Image<uint8_t, 3>& dst, ChromaOrder order) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "ycc_from_rgb");  // classify operands BEFORE touching device pointers
    detail::check(kh_ycc_from_rgb_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src), (int32_t)order));
}
inline void rgb_from_ycc(const Image<uint8_t, 3>& src, Image<uint8_t, 3>& dst, ChromaOrder order) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "rgb_from_ycc");  // classify operands BEFORE touching device pointers
    detail::check(kh_rgb_from_ycc_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src), (int32_t)order));
}
inline void ycc_from_rgb(const Image<float, 3>& src, Image<float, 3>& dst, ChromaOrder order) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "ycc_from_rgb");  // classify operands BEFORE touching device pointers
    detail::check(kh_ycc_from_rgb_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src), (int32_t)order));
}
inline void rgb_from_ycc(const Image<float, 3>& src, Image<float, 3>& dst, ChromaOrder order) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "rgb_from_ycc");  // classify operands BEFORE touching device pointers
    detail::check(kh_rgb_from_ycc_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src), (int32_t)order));
}

// color::{hsv,hls}_from_rgb and inverses, sepia_from_rgb (P/color/hsv/mod.rs, P/color/hls/mod.rs, P/color/sepia.rs); [0, 255] domain
inline void hsv_from_rgb(const Image<float, 3>& src, Image<float, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "hsv_from_rgb");  // classify operands BEFORE touching device pointers
    detail::check(kh_hsv_from_rgb_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src)));
}
inline void rgb_from_hsv(const Image<float, 3>& src, Image<float, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "rgb_from_hsv");  // classify operands BEFORE touching device pointers
    detail::check(kh_rgb_from_hsv_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src)));
}
inline void hls_from_rgb(const Image<float, 3>& src, Image<float, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "hls_from_rgb");  // classify operands BEFORE touching device pointers
    detail::check(kh_hls_from_rgb_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src)));
}
inline void rgb_from_hls(const Image<float, 3>& src, Image<float, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "rgb_from_hls");  // classify operands BEFORE touching device pointers
    detail::check(kh_rgb_from_hls_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src)));
}
inline void sepia_from_rgb(const Image<uint8_t, 3>& src, Image<uint8_t, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "sepia_from_rgb");  // classify operands BEFORE touching device pointers
    detail::check(kh_sepia_from_rgb_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src)));
}
inline void sepia_from_rgb(const Image<float, 3>& src, Image<float, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "sepia_from_rgb");  // classify operands BEFORE touching device pointers
    detail::check(kh_sepia_from_rgb_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src)));
}

// CIE family (P/color/cie/mod.rs:58-160) for f32, and every f64 colour conversion (gray, hsv / hls, ycbcr / yuv, CIE:
// P/color/cuda_dispatch.rs:48-61,111-135).  `conversion` is a KH_CIE_* (both) or KH_F64_* (f64) code.
inline void cie_convert(const Image<float, 3>& src, Image<float, 3>& dst, int conversion) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "cie_convert");  // classify operands BEFORE touching device pointers
    detail::check(kh_cie_convert_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src), conversion));
}
template <int CS, int CD>
inline void color_convert_f64(const Image<double, CS>& src, Image<double, CD>& dst, int conversion) {
    const int want_in = conversion == KH_F64_RGB_FROM_GRAY ? 1 : 3, want_out = conversion == KH_F64_GRAY_FROM_RGB ? 1 : 3;
    if (CS != want_in || CD != want_out)
        throw ImageError(ImageError::Kind::InvalidChannelShape, "color_convert_f64: conversion " + std::to_string(conversion) + " maps " +
                                                                     std::to_string(want_in) + " -> " + std::to_string(want_out) + " channels");
    const detail::DeviceExec s = helpers::map_pair(src, dst, "color_convert_f64");  // classify operands BEFORE touching device pointers
    detail::check(kh_color_convert_f64(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src), conversion));
}

// color::apply_colormap with a caller-provided table r[256] g[256] b[256] (P/color/colormap.rs:252-300)
inline void apply_colormap(const Image<uint8_t, 1>& src, Image<uint8_t, 3>& dst, const std::array<uint8_t, 768>& lut) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "apply_colormap");
    void* dlut = nullptr;
    detail::check(kh_malloc_async(&dlut, 768, 0, s.handle()));
    int32_t rc = kh_memcpy_h2d_async(dlut, lut.data(), 768, s.handle());
    if (rc == KH_OK) rc = kh_stream_synchronize(s.handle());  // the table is caller memory: land it before returning
    if (rc == KH_OK) rc = kh_apply_colormap_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), helpers::npixels(src), static_cast<const uint8_t*>(dlut));
    kh_free_async(dlut, s.handle());
    detail::check(rc);
}

// camera formats <-> RGB8 (P/color/yuv/mod.rs:209-310, 342): raw device buffers in, RGB8 image out (or back)
enum class Planar420 { Nv12 = 0, Nv21 = 1, I420 = 2, Yv12 = 3 };
enum class Packed422 { Yuyv = 0, Uyvy = 1, Yvyu = 2 };
enum class YuvToRgbMode { Bt601Full = KH_YUV_BT601_FULL, Bt709Full = KH_YUV_BT709_FULL, Bt601Limited = KH_YUV_BT601_LIMITED };
namespace helpers {
inline const Stream& raw_to_image(const Image<uint8_t, 3>& dst, const uint8_t* raw, size_t have, size_t need, const char* what) {
    if (!dst.is_device()) throw ImageError(ImageError::Kind::HostPathUnavailable, std::string(what) + ": host image — device backend only");
    if (!raw || have < need) throw ImageError(ImageError::Kind::InvalidImageSize, std::string(what) + ": buffer holds " + std::to_string(have) + " bytes, the format needs " + std::to_string(need));
    return *dst.stream();
}
}  // namespace helpers
inline void rgb_from_planar420(const uint8_t* raw_device, size_t raw_bytes, Image<uint8_t, 3>& dst, Planar420 layout) {
    const Stream& s = helpers::raw_to_image(dst, raw_device, raw_bytes, dst.width() * dst.height() * 3 / 2, "rgb_from_planar420");
    const detail::DeviceScope bind(s.device());
    detail::check(kh_rgb_from_planar420_u8(s.handle(), raw_device, dst.device_ptr_mut(), detail::i32(dst.width()), detail::i32(dst.height()), (int32_t)layout));
}
inline void rgb_from_packed422(const uint8_t* raw_device, size_t raw_bytes, Image<uint8_t, 3>& dst, Packed422 layout) {
    const Stream& s = helpers::raw_to_image(dst, raw_device, raw_bytes, dst.width() * dst.height() * 2, "rgb_from_packed422");
    const detail::DeviceScope bind(s.device());
    detail::check(kh_rgb_from_packed422_u8(s.handle(), raw_device, dst.device_ptr_mut(), detail::i32(dst.width()), detail::i32(dst.height()), (int32_t)layout));
}
inline void convert_yuyv_to_rgb_u8(const uint8_t* raw_device, size_t raw_bytes, Image<uint8_t, 3>& dst, YuvToRgbMode mode) {
    const Stream& s = helpers::raw_to_image(dst, raw_device, raw_bytes, dst.width() * dst.height() * 2, "convert_yuyv_to_rgb_u8");
    const detail::DeviceScope bind(s.device());
    detail::check(kh_yuyv_to_rgb_mode_u8(s.handle(), raw_device, dst.device_ptr_mut(), detail::i32(dst.width()), detail::i32(dst.height()), (int32_t)mode));
}
// encoders write width*height*3/2 (NV12) or width*height*2 (YUYV) bytes at `out_device`
inline void nv12_from_rgb(const Image<uint8_t, 3>& src, uint8_t* out_device) {
    if (!src.is_device()) throw ImageError(ImageError::Kind::HostPathUnavailable, "nv12_from_rgb: host image — device backend only");
    const detail::DeviceScope bind(src.stream()->device());
    detail::check(kh_nv12_from_rgb_u8(src.stream()->handle(), src.device_ptr(), out_device, detail::i32(src.width()), detail::i32(src.height())));
}
inline void yuyv_from_rgb(const Image<uint8_t, 3>& src, uint8_t* out_device) {
    if (!src.is_device()) throw ImageError(ImageError::Kind::HostPathUnavailable, "yuyv_from_rgb: host image — device backend only");
    const detail::DeviceScope bind(src.stream()->device());
    detail::check(kh_yuyv_from_rgb_u8(src.stream()->handle(), src.device_ptr(), out_device, detail::i32(src.width()), detail::i32(src.height())));
}

// color::rgb_from_bayer (P/color/bayer/mod.rs:37-70): bilinear, cv2-compatible demosaic
enum class BayerPattern { Rggb = KH_BAYER_RGGB, Bggr = KH_BAYER_BGGR, Grbg = KH_BAYER_GRBG, Gbrg = KH_BAYER_GBRG };
inline void rgb_from_bayer(const Image<uint8_t, 1>& src, BayerPattern pattern, Image<uint8_t, 3>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "rgb_from_bayer");
    detail::check(kh_rgb_from_bayer_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), (int32_t)pattern));
}

// resize launchers with their PixelMapping, the fused resize + normalise, cv2-compatible resize (P/cuda/resize.rs:433-930,
// P/resize/opencv_compat.rs:76-250)
enum class PixelMapping { HalfPixel = KH_MAP_HALF_PIXEL, AlignCorners = KH_MAP_ALIGN_CORNERS };
template <int C>
inline void resize_mapped(const Image<float, C>& src, Image<float, C>& dst, InterpolationMode interpolation, PixelMapping mapping) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "resize");
    detail::check(kh_resize_mapped_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()),
                                       detail::i32(dst.width()), detail::i32(dst.height()), C, (int32_t)interpolation, (int32_t)mapping, 1, 0, 0));
}
inline void resize_bilinear_normalize(const Image<float, 3>& src, Image<float, 3>& dst, const std::array<float, 3>& mean,
                                      const std::array<float, 3>& std_, PixelMapping mapping = PixelMapping::HalfPixel) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "resize_bilinear_normalize");
    detail::check(kh_resize_bilinear_normalize_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()),
                                                   detail::i32(dst.width()), detail::i32(dst.height()), mean.data(), std_.data(), (int32_t)mapping, 1, 0, 0));
}
template <int C>
inline void resize_opencv(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, InterpolationMode interpolation) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "resize_opencv");
    detail::check(kh_resize_opencv_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()),
                                      detail::i32(dst.width()), detail::i32(dst.height()), C, (int32_t)interpolation, 1, 0, 0));
}
template <int C>
inline void resize_opencv(const Image<float, C>& src, Image<float, C>& dst, InterpolationMode interpolation) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "resize_opencv");
    detail::check(kh_resize_opencv_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()),
                                       detail::i32(dst.width()), detail::i32(dst.height()), C, (int32_t)interpolation, 1, 0, 0));
}

// u8 gathers / filters (P/interpolation/remap.rs:157, P/warp/perspective.rs:179, P/filter/ops.rs:59)
template <int C>
inline void remap(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, const Image<float, 1>& map_x, const Image<float, 1>& map_y,
                  InterpolationMode interpolation) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "remap_u8");
    if (map_x.size() != dst.size() || map_y.size() != dst.size())
        throw ImageError(ImageError::Kind::InvalidImageSize, "remap: map_x, map_y and dst must have the same size");
    if (!map_x.is_device() || !map_y.is_device())
        throw ImageError(ImageError::Kind::MixedResidency, "remap: map_x and map_y must be device-resident when src/dst are on the GPU");
    for (const Image<float, 1>* mp : {&map_x, &map_y})
        if (!mp->stream()->same_as(s.stream())) detail::check(kh_stream_fence(mp->stream()->handle(), s.handle()));
    detail::check(kh_remap_u8(s.handle(), src.device_ptr(), map_x.device_ptr(), map_y.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()),
                              detail::i32(src.height()), detail::i32(dst.width()), detail::i32(dst.height()), C, (int32_t)interpolation, 1, 0, 0));
}
template <int C>
inline void warp_perspective_u8(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, const std::array<float, 9>& m) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "warp_perspective_u8");
    detail::check(kh_warp_perspective_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()),
                                         detail::i32(dst.width()), detail::i32(dst.height()), C, m.data(), 1, 0, 0));
}
template <int C>
inline void box_blur(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, std::pair<int, int> kernel_size) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "box_blur_u8");
    detail::same_size(src, dst, "box_blur_u8");
    detail::check(kh_box_blur_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C,
                                 kernel_size.first, kernel_size.second, 1, 0, 0));
}

// filter::separable_filter, sobel, scharr (P/filter/separable_filter.rs:166, P/filter/ops.rs:174, 214)
template <int C>
inline void separable_filter(const Image<float, C>& src, Image<float, C>& dst, const std::vector<float>& kernel_x, const std::vector<float>& kernel_y) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "separable_filter");
    detail::same_size(src, dst, "separable_filter");
    detail::check(kh_separable_filter_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C,
                                          kernel_x.data(), detail::i32(kernel_x.size()), kernel_y.data(), detail::i32(kernel_y.size()), 1, 0, 0));
}
template <int C>
inline void sobel(const Image<float, C>& src, Image<float, C>& dst, int kernel_size) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "sobel");
    detail::same_size(src, dst, "sobel");
    detail::check(kh_gradient_magnitude_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C,
                                            KH_GRAD_SOBEL, kernel_size, 1, 0, 0));
}
template <int C>
inline void scharr(const Image<float, C>& src, Image<float, C>& dst) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "scharr");
    detail::same_size(src, dst, "scharr");
    detail::check(kh_gradient_magnitude_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C,
                                            KH_GRAD_SCHARR, 3, 1, 0, 0));
}

// filter::spatial_gradient_float / scharr_spatial_gradient_float (P/filter/ops.rs:287, 511): normalised 3x3 derivatives
template <int C>
inline void spatial_gradient_float(const Image<float, C>& src, Image<float, C>& dx, Image<float, C>& dy) {
    const detail::DeviceExec s = detail::device_exec_for(src, dx, "spatial_gradient_float");
    detail::device_exec_for(src, dy, "spatial_gradient_float");
    detail::same_size(src, dx, "spatial_gradient_float");
    detail::same_size(src, dy, "spatial_gradient_float");
    detail::check(kh_spatial_gradient_f32(s.handle(), src.device_ptr(), dx.device_ptr_mut(), dy.device_ptr_mut(), detail::i32(src.width()),
                                          detail::i32(src.height()), C, KH_GRAD_SOBEL, 1, 0, 0));
}
template <int C>
inline void scharr_spatial_gradient_float(const Image<float, C>& src, Image<float, C>& dx, Image<float, C>& dy) {
    const detail::DeviceExec s = detail::device_exec_for(src, dx, "scharr_spatial_gradient_float");
    detail::device_exec_for(src, dy, "scharr_spatial_gradient_float");
    detail::same_size(src, dx, "scharr_spatial_gradient_float");
    detail::same_size(src, dy, "scharr_spatial_gradient_float");
    detail::check(kh_spatial_gradient_f32(s.handle(), src.device_ptr(), dx.device_ptr_mut(), dy.device_ptr_mut(), detail::i32(src.width()),
                                          detail::i32(src.height()), C, KH_GRAD_SCHARR, 1, 0, 0));
}
// filter::box_blur_fast (P/filter/ops.rs:252): the transposed intermediate is a scratch image on the source's stream
template <int C>
inline void box_blur_fast(const Image<float, C>& src, Image<float, C>& dst, std::pair<float, float> sigma) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "box_blur_fast");
    detail::same_size(src, dst, "box_blur_fast");
    auto scratch = Image<float, C>::uninit_hip(src.size(), s.stream());
    detail::check(kh_box_blur_fast_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), scratch.device_ptr_mut(), detail::i32(src.width()),
                                       detail::i32(src.height()), C, sigma.first, sigma.second, 1, 0, 0));
}
// filter::median_blur (P/filter/median.rs:174), filter::bilateral_filter (P/filter/bilateral.rs:172)
template <int C>
inline void median_blur(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, int ksize) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "median_blur");
    detail::same_size(src, dst, "median_blur");
    detail::check(kh_median_blur_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C, ksize,
                                    1, 0, 0));
}
inline void bilateral_filter(const Image<uint8_t, 1>& src, Image<uint8_t, 1>& dst, int d, double sigma_color, double sigma_space) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "bilateral_filter");
    detail::same_size(src, dst, "bilateral_filter");
    detail::check(kh_bilateral_filter_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), d,
                                         sigma_color, sigma_space, 1, 0, 0));
}

// pyramid::pyrdown / pyrup (P/pyramid.rs:180-520): dst is ceil(src / 2) resp. 2 * src
namespace helpers {
template <typename T, int C>
inline detail::DeviceExec pyr_pair(const Image<T, C>& src, const Image<T, C>& dst, bool up, const char* what) {
    detail::DeviceExec s = detail::device_exec_for(src, dst, what);
    const size_t w = up ? src.width() * 2 : (src.width() + 1) / 2, h = up ? src.height() * 2 : (src.height() + 1) / 2;
    if (dst.width() != w || dst.height() != h)
        throw ImageError(ImageError::Kind::InvalidImageSize, std::string(what) + ": destination must be " + std::to_string(w) + "x" + std::to_string(h));
    return s;
}
}  // namespace helpers
template <int C>
inline void pyrdown(const Image<float, C>& src, Image<float, C>& dst) {
    const detail::DeviceExec s = helpers::pyr_pair(src, dst, false, "pyrdown");  // classify operands BEFORE touching device pointers
    detail::check(kh_pyrdown_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C, 1, 0, 0));
}
template <int C>
inline void pyrdown(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst) {
    const detail::DeviceExec s = helpers::pyr_pair(src, dst, false, "pyrdown");  // classify operands BEFORE touching device pointers
    detail::check(kh_pyrdown_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C, 1, 0, 0));
}
template <int C>
inline void pyrup(const Image<float, C>& src, Image<float, C>& dst) {
    const detail::DeviceExec s = helpers::pyr_pair(src, dst, true, "pyrup");  // classify operands BEFORE touching device pointers
    detail::check(kh_pyrup_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C, 1, 0, 0));
}
template <int C>
inline void pyrup(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst) {
    const detail::DeviceExec s = helpers::pyr_pair(src, dst, true, "pyrup");  // classify operands BEFORE touching device pointers
    detail::check(kh_pyrup_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C, 1, 0, 0));
}

// morphology::{Kernel, dilate, erode} (P/morphology/kernels.rs:60-185, ops.rs:120-268)
enum class KernelShape { Box = KH_MORPH_BOX, Cross = KH_MORPH_CROSS, Ellipse = KH_MORPH_ELLIPSE };
enum class PaddingMode { Constant = KH_BORDER_CONSTANT, Replicate = KH_BORDER_REPLICATE, Reflect101 = KH_BORDER_REFLECT101,
                         Reflect = KH_BORDER_REFLECT, Wrap = KH_BORDER_WRAP };
struct Kernel {
    int width = 0, height = 0;
    std::vector<uint8_t> mask;  // row-major, 1 = active tap; anchor (height/2, width/2)
    Kernel(KernelShape shape, int w, int h) : width(w), height(h), mask((size_t)(w > 0 && h > 0 ? w * h : 0)) {
        detail::check(kh_morph_kernel((int32_t)shape, w, h, mask.data()));
    }
    Kernel(KernelShape shape, int size) : Kernel(shape, size, size) {}
};
namespace helpers {
template <int C>
inline void morph(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, const Kernel& k, int op, PaddingMode mode, const std::array<uint8_t, C>& constant, const char* what) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, what);
    detail::same_size(src, dst, what);
    detail::check(kh_morphology_u8(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), C, op, k.mask.data(), k.width, k.height,
                           (int32_t)mode, constant.data(), 1, 0, 0));
}
}  // namespace helpers
template <int C>
inline void dilate(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, const Kernel& kernel, PaddingMode mode, const std::array<uint8_t, C>& constant_value = {}) {
    helpers::morph<C>(src, dst, kernel, KH_MORPH_DILATE, mode, constant_value, "dilate");
}
template <int C>
inline void erode(const Image<uint8_t, C>& src, Image<uint8_t, C>& dst, const Kernel& kernel, PaddingMode mode, const std::array<uint8_t, C>& constant_value = {}) {
    helpers::morph<C>(src, dst, kernel, KH_MORPH_ERODE, mode, constant_value, "erode");
}

// crop::crop_image, flip::{horizontal,vertical}_flip (P/crop.rs:187, P/flip.rs:39, 305)
template <typename T, int C>
inline void crop_image(const Image<T, C>& src, Image<T, C>& dst, size_t x, size_t y) {
    const detail::DeviceExec s = detail::device_exec_for(src, dst, "crop_image");
    detail::check(kh_crop(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()), detail::i32(src.height()), detail::i32(dst.width()),
                          detail::i32(dst.height()), detail::i32(x), detail::i32(y), (int32_t)(C * sizeof(T))));
}
template <typename T, int C>
inline void horizontal_flip(const Image<T, C>& src, Image<T, C>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "horizontal_flip");  // classify operands BEFORE touching device pointers
    detail::check(kh_flip(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()),
                          detail::i32(src.height()), (int32_t)(C * sizeof(T)), 1));
}
template <typename T, int C>
inline void vertical_flip(const Image<T, C>& src, Image<T, C>& dst) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "vertical_flip");  // classify operands BEFORE touching device pointers
    detail::check(kh_flip(s.handle(), src.device_ptr(), dst.device_ptr_mut(), detail::i32(src.width()),
                          detail::i32(src.height()), (int32_t)(C * sizeof(T)), 0));
}

// normalize::find_min_max / normalize_min_max (P/normalize.rs:123, 191): the reduction stays on the device, the pair is read back
template <int C>
inline std::pair<float, float> find_min_max(const Image<float, C>& src) {
    if (!src.is_device()) throw ImageError(ImageError::Kind::HostPathUnavailable, "find_min_max: host image — device backend only");
    const Stream& s = *src.stream();
    const detail::DeviceScope bind(s.device());
    void* scratch = nullptr;
    detail::check(kh_malloc_async(&scratch, 16, 1, s.handle()));
    float* mm = static_cast<float*>(scratch);
    float out[2] = {0.0f, 0.0f};
    int32_t rc = kh_find_min_max_f32(s.handle(), src.device_ptr(), (int64_t)src.numel(), mm, reinterpret_cast<uint32_t*>(mm + 2));
    if (rc == KH_OK) rc = kh_stream_synchronize(s.handle());
    if (rc == KH_OK) rc = kh_memcpy_d2h_async(out, mm, sizeof out, s.handle());
    if (rc == KH_OK) rc = kh_stream_synchronize(s.handle());
    kh_free_async(scratch, s.handle());
    detail::check(rc);
    return {out[0], out[1]};
}
template <int C>
inline void normalize_min_max(const Image<float, C>& src, Image<float, C>& dst, float min, float max) {
    const detail::DeviceExec s = helpers::map_pair(src, dst, "normalize_min_max");
    void* scratch = nullptr;
    detail::check(kh_malloc_async(&scratch, 16, 1, s.handle()));
    float* mm = static_cast<float*>(scratch);
    const int32_t rc = kh_normalize_min_max_f32(s.handle(), src.device_ptr(), dst.device_ptr_mut(), (int64_t)src.numel(), min, max, mm,
                                                reinterpret_cast<uint32_t*>(mm + 2));
    kh_free_async(scratch, s.handle());  // stream-ordered: released after the kernels that use it
    detail::check(rc);
}

// calibration::distortion::generate_correction_map_polynomial (P/calibration/distortion.rs:135-152): intrinsic = {fx, fy, cx, cy},
// distortion = {k1..k6, p1, p2}
inline void generate_correction_map_polynomial(Image<float, 1>& map_x, Image<float, 1>& map_y, const std::array<double, 4>& intrinsic,
                                               const std::array<double, 8>& distortion) {
    const detail::DeviceExec s = helpers::map_pair(map_x, map_y, "generate_correction_map_polynomial");
    detail::check(kh_correction_map_polynomial_f32(s.handle(), map_x.device_ptr_mut(), map_y.device_ptr_mut(), detail::i32(map_x.width()),
                                                   detail::i32(map_x.height()), intrinsic.data(), distortion.data()));
}

}  // namespace imgproc

// ---- fused camera preprocess (Preprocessor / PreprocessorBuilder, P/preprocess.rs:654-1282) ------------------
enum class ResizeMode { Stretch, Letterbox };
enum class SourceFormat { Rgb8 = KH_FMT_RGB, Bgr8 = KH_FMT_BGR, Gray8 = KH_FMT_GRAY, Nv12 = KH_FMT_NV12, Yuyv = KH_FMT_YUYV };

class Preprocessor {
public:
    Preprocessor(const Stream& stream, ResizeMode mode, SourceFormat format, std::array<float, 3> mean = {0, 0, 0}, std::array<float, 3> std_ = {1, 1, 1},
                 float pad_value = 114.0f, int sampling = KH_SAMPLE_BILINEAR)
        : stream_(stream), mode_(mode), format_(format), mean_(mean), pad_value_(pad_value), sampling_(sampling) {
        // Normalize::mean_inv_std (P/preprocess.rs:107-121): finite mean, finite std > 0
        for (int c = 0; c < 3; ++c) {
            if (!(std_[c] > 0.0f) || !(std_[c] <= 3.4028234e38f) || !(mean[c] == mean[c]) || !(mean[c] >= -3.4028234e38f && mean[c] <= 3.4028234e38f))
                throw ImageError(ImageError::Kind::InvalidNormalize, "invalid normalize: mean must be finite, std must be finite and > 0");
            inv_std_[c] = 1.0f / std_[c];
        }
    }
    // run_raw (P/preprocess.rs:1184-1222): one raw device frame -> dst [1, 3, dst_h, dst_w] f32 (device pointers)
    void run_raw(const uint8_t* src_device, size_t src_bytes, int src_w, int src_h, float* dst_device, int dst_w, int dst_h, int nframes = 1,
                 int64_t src_frame_stride = 0) const {
        // SourceFormat::{bpp, pitch, buffer_len, dims_ok} (P/preprocess.rs:131-250)
        const bool nv12 = format_ == SourceFormat::Nv12, yuyv = format_ == SourceFormat::Yuyv;
        const int bpp = (format_ == SourceFormat::Rgb8 || format_ == SourceFormat::Bgr8) ? 3 : (yuyv ? 2 : 1);
        const size_t need = (size_t)src_w * bpp * src_h + (nv12 ? (size_t)src_w * src_h / 2 : 0);
        if (src_w <= 0 || src_h <= 0 || src_bytes < need || (nv12 && ((src_w | src_h) & 1)) || (yuyv && (src_w & 1)))
            throw ImageError(ImageError::Kind::InvalidImageSize, "preprocess: invalid raw source (need " + std::to_string(need) + " bytes, even dimensions for 4:2:x)");
        kh_preprocess_params p{};
        // Affine (P/preprocess.rs:350-369), f32 arithmetic: letterbox s = min(dw/sw, dh/sh), pad = (d - src*s) * 0.5
        if (mode_ == ResizeMode::Stretch) {
            p.scale_x = (float)dst_w / (float)src_w; p.scale_y = (float)dst_h / (float)src_h; p.pad_x = 0.0f; p.pad_y = 0.0f;
        } else {
            const float sx = (float)dst_w / (float)src_w, sy = (float)dst_h / (float)src_h, sc = sx < sy ? sx : sy;
            p.scale_x = sc; p.scale_y = sc;
            p.pad_x = ((float)dst_w - (float)src_w * sc) * 0.5f; p.pad_y = ((float)dst_h - (float)src_h * sc) * 0.5f;
        }
        p.src_w = src_w; p.src_h = src_h; p.src_pitch = src_w * bpp; p.src_bpp = bpp; p.fmt = (int32_t)format_;
        p.dst_w = dst_w; p.dst_h = dst_h;
        for (int c = 0; c < 3; ++c) { p.mean[c] = mean_[c]; p.inv_std[c] = inv_std_[c]; }
        p.pad_value = pad_value_; p.sampling = sampling_; p.out_dtype = KH_OUT_F32; p.nframes = nframes; p.flags = 0;
        p.src_frame_stride = src_frame_stride; p.dst_frame_stride = (int64_t)3 * dst_w * dst_h;
        const detail::DeviceScope bind(stream_.device());
        detail::check(kh_preprocess_to_chw(stream_.handle(), src_device, dst_device, &p));
    }
    const Stream& stream() const { return stream_; }

private:
    Stream stream_;
    ResizeMode mode_;
    SourceFormat format_;
    std::array<float, 3> mean_, inv_std_;
    float pad_value_;
    int sampling_;
};


// ---- captured graph (kornia_rs.cuda.Graph, PY/cuda_ext/mod.rs:1684-1790): record allocation-free work once, replay per frame ----
class Graph {
public:
    // f() enqueues the work on `stream` (preallocated outputs only); the capture is always ended, also when f throws
    template <typename F>
    static Graph capture(const Stream& stream, F&& f) {
        const detail::DeviceScope bind(stream.device());   // the captured launches and the capture itself: the stream's device
        detail::check(kh_graph_capture_begin(stream.handle()));
        kh_graph_t g = nullptr;
        try {
            f();
        } catch (...) {
            if (kh_graph_capture_end(stream.handle(), &g) == KH_OK) kh_graph_destroy(g);
            throw;
        }
        detail::check(kh_graph_capture_end(stream.handle(), &g));
        return Graph(g, stream);
    }
    void replay() const {
        const detail::DeviceScope bind(stream_.device());
        detail::check(kh_graph_launch(g_.get(), stream_.handle()));
    }

private:
    Graph(kh_graph_t g, const Stream& s) : g_(g, [](kh_graph_t p) { kh_graph_destroy(p); }), stream_(s) {}
    std::shared_ptr<kh_graph_s> g_;
    Stream stream_;
};

}  // namespace kornia
