// Device runtime + HIP allocator behind the C ABI (include/kornia_hip.h).
//
// Stands in for the cudarc-based half of the reference's tensor crate
// (crates/kornia-tensor/src/cuda.rs): stream-ordered zeroed/uninit device allocation from the
// device mem-pool (:238-298, :860, :891), pinned host buffers (:355-380), managed memory
// (:440-511), H2D/D2H copies on the buffer's stream (:1208-1384) and the cross-stream event
// fence of the residency dispatch (crates/kornia-imgproc/src/cuda/dispatch.rs:50-82).
#include <stdarg.h>
#include <string.h>

#include <atomic>
#include <iterator>
#include <map>
#include <utility>
#include <mutex>
#include <set>
#include <string>

#include "kh_common.h"
#include "kornia_hip_testing.h"   // the two kh_debug_* test hooks are defined here

namespace kh {

// development / test options (kh_common.h::DevOpt); -1 = unset.  PER THREAD: a launcher reads the options of the thread that calls
// it, so a test that forces a fallback kernel reroutes its own launches only — no other thread of the process (another component
// sharing the library, a sharding worker) is affected, and there is nothing to race on.
struct DevOpts {
    int v[kOptCount];
    DevOpts() { for (int& x : v) x = -1; }
};
static thread_local DevOpts g_dev_opts;
int dev_opt(DevOpt o) { return g_dev_opts.v[o]; }
void set_dev_opt(int o, int value) { g_dev_opts.v[o] = value; }

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int32_t fail_hip(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return KH_ERR_HIP;
}

namespace {
// Workspaces are keyed by (device, stream handle): handle 0 is the default stream of EVERY device, so the handle alone would hand a
// buffer registered for device 0 to work launched on device 1's default stream (round-2 ADVICE).  Entries are erased when their
// stream is destroyed (a later stream may reuse the handle value) and when the owner unregisters.
struct Workspace { void* ptr; size_t bytes; };
using WsKey = std::pair<int, kh_stream_t>;
std::mutex g_ws_mu;
std::map<WsKey, Workspace>& workspaces() { static auto& m = *new std::map<WsKey, Workspace>(); return m; }
thread_local size_t g_last_scratch = 0;
int current_device_or_zero() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d;
}
}  // namespace

Scratch::~Scratch() {
    if (pooled && ptr) (void)hipFreeAsync(ptr, as_hip(stream));
}

int32_t get_scratch(kh_stream_t stream, size_t bytes, const char* what, Scratch& out) {
    g_last_scratch = bytes;
    out.stream = stream;
    if (bytes == 0) return KH_OK;
    {
        std::lock_guard<std::mutex> lock(g_ws_mu);
        auto it = workspaces().find(WsKey{current_device_or_zero(), stream});  // launches go to the current device
        if (it != workspaces().end() && it->second.bytes >= bytes) {
            out.ptr = it->second.ptr;  // caller-owned: nothing to allocate, nothing to free
            return KH_OK;
        }
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(as_hip(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(KH_ERR_INVALID_ARG, "%s needs %zu bytes of scratch and the stream is being captured: captured work must not allocate — "
                                        "register a device buffer of at least that size with kh_stream_set_workspace before the capture", what, bytes);
    if (int32_t rc = kh_malloc_async(&out.ptr, bytes, 0, stream)) return rc;
    out.pooled = true;
    return KH_OK;
}

int32_t check_list(const char* what, const void* const* srcs, void* const* dsts, int n) {
    KH_REQUIRE(n >= 0 && n <= 65535, KH_ERR_TOO_LARGE, "%s: %d images outside [0, 65535]", what, n);
    if (n == 0) return KH_OK;
    KH_REQUIRE(srcs && dsts, KH_ERR_INVALID_ARG, "%s: null pointer list", what);
    for (int k = 0; k < n; ++k) {
        KH_REQUIRE(srcs[k] && dsts[k], KH_ERR_INVALID_ARG, "%s: null device pointer at list index %d", what, k);
        KH_REQUIRE(srcs[k] != dsts[k], KH_ERR_INVALID_ARG, "%s: source and destination alias at list index %d", what, k);
    }
    return KH_OK;
}

bool batch_aligned(const BatchRef& b, size_t align, size_t elem) {
    if (!b.listed())
        return reinterpret_cast<uintptr_t>(b.src) % align == 0 && reinterpret_cast<uintptr_t>(b.dst) % align == 0 &&
               (b.n <= 1 || ((uint64_t)b.ss * elem % align == 0 && (uint64_t)b.ds * elem % align == 0));
    for (int k = 0; k < b.n; ++k)
        if (reinterpret_cast<uintptr_t>(b.srcs[k]) % align || reinterpret_cast<uintptr_t>(b.dsts[k]) % align) return false;
    return true;
}

}  // namespace kh

using namespace kh;

extern "C" {

size_t kh_last_error(char* buf, size_t cap) {
    size_t n = strlen(g_err);
    if (buf && cap) {
        size_t m = n < cap - 1 ? n : cap - 1;
        memcpy(buf, g_err, m);
        buf[m] = 0;
    }
    return n;
}

const char* kh_version(void) { return "kornia-hip 0.1.0 (gfx950)"; }

// A DLManagedTensor deleter that does nothing.  The Python host swaps it into every still-exported tensor when the
// interpreter starts to finalise: a consumer (e.g. a torch tensor destroyed during shutdown) may call the deleter after the
// Python callback it was exported with can no longer run; the process is exiting, so the memory is simply not released.
void kh_dlpack_noop_deleter(void* managed_tensor) { (void)managed_tensor; }

// test hook: the launch-constant division used to decode tile ids (kh_common.h::FastDiv), on the host
uint32_t kh_debug_fast_quot(uint32_t n, uint32_t d) { return kh::fast_quot(n, kh::fast_div(d)); }

// test hook (include/kornia_hip_testing.h): force one of the alternate code paths a launcher can take (kh_common.h::DevOpt) for the
// CALLING THREAD's launches.  `value` -1 restores the production choice.  Unknown names are an error, so a stale test cannot
// silently test nothing.
int32_t kh_debug_set_option(const char* name, int32_t value) {
    static const char* const names[kh::kOptCount] = {
        "pre_ieee_div", "pre_grid", "pre_quads", "filter_force_tile", "filter_four_columns", "grad_scalar", "hfilter_direct",
        "resize_u8_gather", "pyr_direct", "pyr_roll", "morph_direct", "morph_roll", "u8_blur_rgb", "u8_blur_swar", "warp_u8_direct", "warp_u8_spans", "warp_u8_rows", "resize_rows", "warp_f32_px", "resize_u8_px", "row_stores", "pre_f16_lut"};
    if (name)
        for (int i = 0; i < kh::kOptCount; ++i)
            if (strcmp(name, names[i]) == 0) {
                kh::set_dev_opt(i, value);
                return KH_OK;
            }
    return kh::fail(KH_ERR_INVALID_ARG, "kh_debug_set_option: unknown option '%s'", name ? name : "(null)");
}

// How many HIP runtime images (libamdhip64*) are mapped into this process, and which.  More than one is unsafe: each
// brings its own HSA runtime, and copies / stream waits issued through one do not order against the other (the
// round-1 "incomplete copy at hipStreamSynchronize" finding: libkornia_hip.so bound to /opt/rocm's runtime, then
// `import torch` mapped the wheel's bundled copy beside it — profiles/r02*_runtimes.log).  A host that mixes this
// library with another HIP user must make both resolve to ONE runtime image (load order / RPATH); this entry lets it
// check.  `buf` receives the paths separated by newlines.  Linux only (/proc/self/maps); returns 0 images elsewhere.
int32_t kh_hip_runtime_images(char* buf, size_t cap) {
    std::set<std::string> images;
    if (FILE* f = fopen("/proc/self/maps", "r")) {
        char line[4352];
        while (fgets(line, sizeof(line), f)) {
            const char* path = strchr(line, '/');
            if (!path) continue;
            const char* base = strrchr(path, '/');
            if (strncmp(base + 1, "libamdhip64", 11) != 0) continue;
            std::string p(path);
            while (!p.empty() && (p.back() == '\n' || p.back() == ' ')) p.pop_back();
            images.insert(p);
        }
        fclose(f);
    }
    if (buf && cap) {
        std::string all;
        for (const auto& p : images) { all += p; all += '\n'; }
        const size_t m = all.size() < cap - 1 ? all.size() : cap - 1;
        memcpy(buf, all.data(), m);
        buf[m] = 0;
    }
    return (int32_t)images.size();
}

// The HIP version this library was COMPILED against and the one the runtime image it is bound to reports (both encoded
// major * 10 000 000 + minor * 100 000 + patch).  The Python host binds the library to the torch wheel's bundled runtime when one is
// installed (one runtime per process, see above), which can be OLDER than the ROCm the library was built with: the host layer warns
// when the runtime's major differs or its (major, minor) is below the build's, instead of finding out through a missing symbol.
int32_t kh_hip_versions(int32_t* build_version, int32_t* runtime_version) {
    KH_REQUIRE(build_version && runtime_version, KH_ERR_INVALID_ARG, "kh_hip_versions: null out pointer");
    *build_version = HIP_VERSION;
    int rv = 0;
    if (hipRuntimeGetVersion(&rv) != hipSuccess) { (void)hipGetLastError(); rv = 0; }
    *runtime_version = rv;
    return KH_OK;
}

int32_t kh_device_count(int32_t* count) {
    KH_REQUIRE(count, KH_ERR_INVALID_ARG, "kh_device_count: null out pointer");
    int n = 0;
    KH_HIP(hipGetDeviceCount(&n));
    *count = n;
    return KH_OK;
}

int32_t kh_set_device(int32_t device) {
    KH_HIP(hipSetDevice(device));
    return KH_OK;
}

int32_t kh_get_device(int32_t* device) {
    KH_REQUIRE(device, KH_ERR_INVALID_ARG, "kh_get_device: null out pointer");
    int d = 0;
    KH_HIP(hipGetDevice(&d));
    *device = d;
    return KH_OK;
}

int32_t kh_device_info(int32_t device, char* name, size_t name_cap, int32_t* cu_count,
                       uint64_t* total_mem_bytes) {
    hipDeviceProp_t prop;
    KH_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_cap) {
        snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (total_mem_bytes) *total_mem_bytes = (uint64_t)prop.totalGlobalMem;
    return KH_OK;
}

int32_t kh_stream_create(kh_stream_t* out) {
    KH_REQUIRE(out, KH_ERR_INVALID_ARG, "kh_stream_create: null out pointer");
    hipStream_t s = nullptr;
    KH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = reinterpret_cast<kh_stream_t>(s);
    return KH_OK;
}

int32_t kh_stream_destroy(kh_stream_t stream) {
    KH_REQUIRE(stream, KH_ERR_INVALID_ARG, "kh_stream_destroy: the default stream is not owned");
    {   // a workspace registered for this stream dies with it: the handle value may be reused by a later stream
        std::lock_guard<std::mutex> lock(g_ws_mu);
        auto& m = workspaces();
        for (auto it = m.begin(); it != m.end();) it = it->first.second == stream ? m.erase(it) : std::next(it);
    }
    KH_HIP(hipStreamDestroy(as_hip(stream)));
    return KH_OK;
}

int32_t kh_stream_synchronize(kh_stream_t stream) {
    KH_HIP(hipStreamSynchronize(as_hip(stream)));
    return KH_OK;
}

int32_t kh_stream_wait_event(kh_stream_t stream, kh_event_t event) {
    KH_REQUIRE(event, KH_ERR_INVALID_ARG, "kh_stream_wait_event: null event");
    KH_HIP(hipStreamWaitEvent(as_hip(stream), as_hip(event), 0));
    return KH_OK;
}

int32_t kh_event_create(kh_event_t* out, int32_t enable_timing) {
    KH_REQUIRE(out, KH_ERR_INVALID_ARG, "kh_event_create: null out pointer");
    hipEvent_t e = nullptr;
    KH_HIP(hipEventCreateWithFlags(&e, enable_timing ? hipEventDefault : hipEventDisableTiming));
    *out = reinterpret_cast<kh_event_t>(e);
    return KH_OK;
}

int32_t kh_event_destroy(kh_event_t event) {
    KH_REQUIRE(event, KH_ERR_INVALID_ARG, "kh_event_destroy: null event");
    KH_HIP(hipEventDestroy(as_hip(event)));
    return KH_OK;
}

int32_t kh_event_record(kh_event_t event, kh_stream_t stream) {
    KH_REQUIRE(event, KH_ERR_INVALID_ARG, "kh_event_record: null event");
    KH_HIP(hipEventRecord(as_hip(event), as_hip(stream)));
    return KH_OK;
}

int32_t kh_event_synchronize(kh_event_t event) {
    KH_REQUIRE(event, KH_ERR_INVALID_ARG, "kh_event_synchronize: null event");
    KH_HIP(hipEventSynchronize(as_hip(event)));
    return KH_OK;
}

int32_t kh_event_elapsed_ms(kh_event_t start, kh_event_t stop, float* ms) {
    KH_REQUIRE(start && stop && ms, KH_ERR_INVALID_ARG, "kh_event_elapsed_ms: null argument");
    KH_HIP(hipEventElapsedTime(ms, as_hip(start), as_hip(stop)));
    return KH_OK;
}

int32_t kh_stream_fence(kh_stream_t producer, kh_stream_t consumer) {
    if (producer == consumer) return KH_OK;  // same queue: already ordered
    // The event must belong to the PRODUCER's device: hipEventRecord rejects an event of another device (hipErrorInvalidHandle),
    // while hipStreamWaitEvent accepts one — that is how a consumer stream of GPU b is ordered after a producer of GPU a.  A
    // null producer is the current device's null stream, so the current device is already the right one for it.
    int prev = -1;
    hipDevice_t pdev = -1;
    KH_HIP(hipGetDevice(&prev));
    const bool hop = producer && hipStreamGetDevice(as_hip(producer), &pdev) == hipSuccess && (int)pdev != prev;
    if (hop) KH_HIP(hipSetDevice((int)pdev));
    hipEvent_t e = nullptr;
    hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (r == hipSuccess) r = hipEventRecord(e, as_hip(producer));
    // back on the CALLER's device before the wait: a NULL consumer is the caller's current-device null stream, as the header says —
    // waiting while the producer's device is current would order that device's null stream instead
    const hipError_t b = hop ? hipSetDevice(prev) : hipSuccess;
    if (r == hipSuccess && b == hipSuccess) r = hipStreamWaitEvent(as_hip(consumer), e, 0);
    const hipError_t d = e ? hipEventDestroy(e) : hipSuccess;  // safe: the wait keeps its own reference
    if (r != hipSuccess) return fail_hip(r, "kh_stream_fence");
    if (d != hipSuccess) return fail_hip(d, "kh_stream_fence(destroy)");
    if (b != hipSuccess) return fail_hip(b, "kh_stream_fence(restore device)");
    return KH_OK;
}

// cuMemGetInfo twin (kornia_rs.cuda.mem_get_info, PY/cuda_ext/mod.rs): free / total bytes of the current device.
int32_t kh_mem_get_info(uint64_t* free_bytes, uint64_t* total_bytes) {
    KH_REQUIRE(free_bytes && total_bytes, KH_ERR_INVALID_ARG, "kh_mem_get_info: null output");
    size_t f = 0, t = 0;
    KH_HIP(hipMemGetInfo(&f, &t));
    *free_bytes = f; *total_bytes = t;
    return KH_OK;
}

// ---- stream capture -> executable graph (kornia_rs.cuda.Graph, PY/cuda_ext/mod.rs:1684-1790) -------------------
// Record the launches a caller enqueues on `stream` between begin and end into a hipGraph and replay them at the
// cost of one launch.  Thread-local capture mode, as the reference uses.  The captured work must be allocation-free
// (preallocated outputs): the launchers that take stream-ordered scratch say so in the header.
struct kh_graph_s { hipGraph_t graph; hipGraphExec_t exec; };

int32_t kh_graph_capture_begin(kh_stream_t stream) {
    KH_REQUIRE(stream, KH_ERR_INVALID_ARG, "kh_graph_capture_begin: capture needs a non-default stream");
    KH_HIP(hipStreamBeginCapture(as_hip(stream), hipStreamCaptureModeThreadLocal));
    return KH_OK;
}

int32_t kh_graph_capture_end(kh_stream_t stream, kh_graph_t* out) {
    KH_REQUIRE(stream && out, KH_ERR_INVALID_ARG, "kh_graph_capture_end: null argument");
    *out = nullptr;
    hipGraph_t g = nullptr;
    KH_HIP(hipStreamEndCapture(as_hip(stream), &g));  // always ends the capture, so the stream is usable again
    KH_REQUIRE(g, KH_ERR_INVALID_ARG, "kh_graph_capture_end: nothing was captured (no device work was enqueued on this stream)");
    size_t nodes = 0;
    hipError_t r = hipGraphGetNodes(g, nullptr, &nodes);
    if (r == hipSuccess && nodes == 0) {
        (void)hipGraphDestroy(g);
        return fail(KH_ERR_INVALID_ARG, "kh_graph_capture_end: nothing was captured (no device work was enqueued on this stream)");
    }
    hipGraphExec_t exec = nullptr;
    if (r == hipSuccess) r = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    if (r != hipSuccess) {
        (void)hipGraphDestroy(g);
        return fail_hip(r, "kh_graph_capture_end");
    }
    *out = new kh_graph_s{g, exec};
    return KH_OK;
}

int32_t kh_graph_launch(kh_graph_t graph, kh_stream_t stream) {
    KH_REQUIRE(graph, KH_ERR_INVALID_ARG, "kh_graph_launch: null graph");
    KH_HIP(hipGraphLaunch(graph->exec, as_hip(stream)));
    return KH_OK;
}

int32_t kh_graph_destroy(kh_graph_t graph) {
    if (!graph) return KH_OK;
    hipError_t a = hipGraphExecDestroy(graph->exec), b = hipGraphDestroy(graph->graph);
    delete graph;
    if (a != hipSuccess) return fail_hip(a, "kh_graph_destroy");
    if (b != hipSuccess) return fail_hip(b, "kh_graph_destroy");
    return KH_OK;
}

// The default mem-pool trims itself to the release threshold (0) at every synchronisation point.
// The reference raises the threshold so steady-state alloc/free never goes back to the driver
// (crates/kornia-tensor/src/cuda.rs:238-262); here it is also a correctness matter: on ROCm 7.2 /
// gfx950 we observed a trim of a large freed block invalidate a small block re-allocated from it
// (reads came back as zeros).  Keep everything cached; kh_mempool_set_release_threshold can lower it.
static int32_t pin_pool_once() {
    // 0 = not done, 1 = done.  Racing first calls may both set the attribute (idempotent); nobody reads a torn flag.
    static std::atomic<uint8_t> done[64];
    int dev = 0;
    KH_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || done[dev].load(std::memory_order_acquire)) return KH_OK;
    hipMemPool_t pool = nullptr;
    KH_HIP(hipDeviceGetDefaultMemPool(&pool, dev));
    uint64_t v = UINT64_MAX;
    KH_HIP(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &v));
    done[dev].store(1, std::memory_order_release);
    return KH_OK;
}

int32_t kh_malloc_async(void** out, size_t bytes, int32_t zeroed, kh_stream_t stream) {
    KH_REQUIRE(out, KH_ERR_INVALID_ARG, "kh_malloc_async: null out pointer");
    *out = nullptr;
    if (bytes == 0) return KH_OK;  // empty tensors own no storage
    if (int32_t rc = pin_pool_once()) return rc;
    void* p = nullptr;
    KH_HIP(hipMallocAsync(&p, bytes, as_hip(stream)));
    if (zeroed) {
        hipError_t e = hipMemsetAsync(p, 0, bytes, as_hip(stream));
        if (e != hipSuccess) {
            (void)hipFreeAsync(p, as_hip(stream));
            return fail_hip(e, "hipMemsetAsync (zeroed allocation)");
        }
    }
    *out = p;
    return KH_OK;
}

int32_t kh_free_async(void* ptr, kh_stream_t stream) {
    if (!ptr) return KH_OK;
    KH_HIP(hipFreeAsync(ptr, as_hip(stream)));
    return KH_OK;
}

int32_t kh_mempool_set_release_threshold(int32_t device, uint64_t bytes) {
    hipMemPool_t pool = nullptr;
    KH_HIP(hipDeviceGetDefaultMemPool(&pool, device));
    uint64_t v = bytes;
    KH_HIP(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &v));
    return KH_OK;
}

int32_t kh_stream_set_workspace(kh_stream_t stream, void* device_ptr, size_t bytes) {
    KH_REQUIRE((device_ptr != nullptr) == (bytes != 0), KH_ERR_INVALID_ARG, "kh_stream_set_workspace: pointer and size must both be set, or both zero to unregister");
    // The entry belongs to the CURRENT device (the one `stream` was created on; hosts select it before the call, as for every
    // launch).  A buffer that lives on another device is refused instead of being handed to kernels that cannot reach it.
    const int dev = current_device_or_zero();
    if (stream) {   // a created stream knows its device: a registration made with another device selected would be filed where no
                    // launch on this stream ever looks — refuse it instead of losing the workspace silently
        hipDevice_t sdev = 0;
        if (hipStreamGetDevice(as_hip(stream), &sdev) == hipSuccess) {
            KH_REQUIRE((int)sdev == dev, KH_ERR_INVALID_ARG,
                       "kh_stream_set_workspace: the stream belongs to device %d but device %d is current — select the stream's device first", (int)sdev, dev);
        } else {
            (void)hipGetLastError();
        }
    }
    if (device_ptr) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, device_ptr) == hipSuccess) {
            KH_REQUIRE(at.type != hipMemoryTypeDevice || at.device == dev, KH_ERR_INVALID_ARG,
                       "kh_stream_set_workspace: the buffer lives on device %d but the current device is %d", at.device, dev);
        } else {
            (void)hipGetLastError();
        }
    }
    std::lock_guard<std::mutex> lock(g_ws_mu);
    if (!device_ptr) workspaces().erase(WsKey{dev, stream});
    else workspaces()[WsKey{dev, stream}] = Workspace{device_ptr, bytes};
    return KH_OK;
}

int32_t kh_stream_workspace_bytes(kh_stream_t stream, size_t* bytes) {
    KH_REQUIRE(bytes, KH_ERR_INVALID_ARG, "kh_stream_workspace_bytes: null out pointer");
    std::lock_guard<std::mutex> lock(g_ws_mu);
    auto it = workspaces().find(WsKey{current_device_or_zero(), stream});
    *bytes = it == workspaces().end() ? 0 : it->second.bytes;
    return KH_OK;
}

int32_t kh_last_workspace_bytes(size_t* bytes) {
    KH_REQUIRE(bytes, KH_ERR_INVALID_ARG, "kh_last_workspace_bytes: null out pointer");
    *bytes = g_last_scratch;
    return KH_OK;
}

int32_t kh_host_alloc(void** out, size_t bytes) {
    KH_REQUIRE(out, KH_ERR_INVALID_ARG, "kh_host_alloc: null out pointer");
    *out = nullptr;
    if (bytes == 0) return KH_OK;
    void* p = nullptr;
    KH_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    memset(p, 0, bytes);
    *out = p;
    return KH_OK;
}

int32_t kh_host_free(void* ptr) {
    if (!ptr) return KH_OK;
    KH_HIP(hipHostFree(ptr));
    return KH_OK;
}

int32_t kh_malloc_managed(void** out, size_t bytes) {
    KH_REQUIRE(out, KH_ERR_INVALID_ARG, "kh_malloc_managed: null out pointer");
    *out = nullptr;
    if (bytes == 0) return KH_OK;
    void* p = nullptr;
    KH_HIP(hipMallocManaged(&p, bytes, hipMemAttachGlobal));
    hipError_t e = hipMemset(p, 0, bytes);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail_hip(e, "hipMemset (managed allocation)");
    }
    *out = p;
    return KH_OK;
}

int32_t kh_free(void* ptr) {
    if (!ptr) return KH_OK;
    KH_HIP(hipFree(ptr));
    return KH_OK;
}

int32_t kh_memcpy_h2d_async(void* dst, const void* src, size_t bytes, kh_stream_t stream) {
    if (bytes == 0) return KH_OK;
    KH_REQUIRE(dst && src, KH_ERR_INVALID_ARG, "kh_memcpy_h2d_async: null pointer");
    KH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_hip(stream)));
    return KH_OK;
}

int32_t kh_memcpy_d2h_async(void* dst, const void* src, size_t bytes, kh_stream_t stream) {
    if (bytes == 0) return KH_OK;
    KH_REQUIRE(dst && src, KH_ERR_INVALID_ARG, "kh_memcpy_d2h_async: null pointer");
    KH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_hip(stream)));
    return KH_OK;
}

int32_t kh_memcpy_d2d_async(void* dst, const void* src, size_t bytes, kh_stream_t stream) {
    if (bytes == 0) return KH_OK;
    KH_REQUIRE(dst && src, KH_ERR_INVALID_ARG, "kh_memcpy_d2d_async: null pointer");
    KH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_hip(stream)));
    return KH_OK;
}

int32_t kh_memset_async(void* dst, int32_t value, size_t bytes, kh_stream_t stream) {
    if (bytes == 0) return KH_OK;
    KH_REQUIRE(dst, KH_ERR_INVALID_ARG, "kh_memset_async: null pointer");
    KH_HIP(hipMemsetAsync(dst, value, bytes, as_hip(stream)));
    return KH_OK;
}

int32_t kh_pointer_domain(const void* ptr, int32_t* domain, int32_t* device) {
    KH_REQUIRE(domain && device, KH_ERR_INVALID_ARG, "kh_pointer_domain: null out pointer");
    *domain = KH_DOMAIN_HOST;
    *device = -1;
    if (!ptr) return KH_OK;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess) {
        // Plain malloc'd memory is "invalid value" to the runtime: that IS the Host answer.
        (void)hipGetLastError();
        return KH_OK;
    }
    switch (attr.type) {
        case hipMemoryTypeDevice:
            *domain = KH_DOMAIN_DEVICE;
            *device = attr.device;
            break;
        case hipMemoryTypeManaged:
            *domain = KH_DOMAIN_UNIFIED;
            *device = attr.device;
            break;
        case hipMemoryTypeHost:
            *domain = KH_DOMAIN_HOST_PINNED;
            *device = attr.device;
            break;
        default:
            break;
    }
    return KH_OK;
}

}  // extern "C"
