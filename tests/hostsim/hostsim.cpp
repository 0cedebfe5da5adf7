// TEST INFRASTRUCTURE — fiber scheduler + HIP runtime emulation behind tests/hostsim/hip/hip_runtime.h.
#include <ucontext.h>

#include <map>
#include <mutex>
#include <vector>

#include "hip/hip_runtime.h"

namespace hostsim {

namespace {
constexpr int kWaveSize = 64;
constexpr size_t kStack = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    Lane lane;
    bool done = false;
    int wave = 0, lane_in_wave = 0;
    char* stack = nullptr;
};
struct Wave {
    int alive = 0, arrived = 0;
    unsigned generation = 0;
    uint32_t slot[kWaveSize];
};

ucontext_t g_sched;
std::vector<Fiber> g_fibers;
std::vector<Wave> g_waves;
std::vector<char*> g_stacks;
Fiber* g_me = nullptr;
const std::function<void()>* g_body = nullptr;
int g_block_alive = 0, g_block_arrived = 0;
unsigned g_block_generation = 0;
std::mutex g_launch_lock;  // one launch at a time (the emulated device is a single queue)

void yield() { swapcontext(&g_me->ctx, &g_sched); }

void retire(Fiber* f) {  // an exited lane no longer takes part in barriers: release waiters it would have completed
    f->done = true;
    Wave& w = g_waves[f->wave];
    if (--w.alive > 0 && w.arrived == w.alive) { w.arrived = 0; ++w.generation; }
    if (--g_block_alive > 0 && g_block_arrived == g_block_alive) { g_block_arrived = 0; ++g_block_generation; }
}

void trampoline() {
    (*g_body)();
    retire(g_me);
    swapcontext(&g_me->ctx, &g_sched);
}
}  // namespace

Lane* cur = nullptr;

int lane_id() { return g_me->lane_in_wave; }

void block_barrier() {
    const unsigned gen = g_block_generation;
    if (++g_block_arrived == g_block_alive) { g_block_arrived = 0; ++g_block_generation; return; }
    while (g_block_generation == gen) yield();
}

void wave_barrier() {
    Wave& w = g_waves[g_me->wave];
    const unsigned gen = w.generation;
    if (++w.arrived == w.alive) { w.arrived = 0; ++w.generation; return; }
    while (w.generation == gen) yield();
}

uint32_t shfl_bits(uint32_t v, int src_lane) {
    Wave& w = g_waves[g_me->wave];
    w.slot[g_me->lane_in_wave] = v;
    wave_barrier();                       // every active lane has published
    const uint32_t r = w.slot[src_lane & (kWaveSize - 1)];
    wave_barrier();                       // every active lane has read before the slots are reused
    return r;
}

struct Recorded { dim3 grid, block; std::function<void()> body; };
std::vector<Recorded>* g_capture = nullptr;  // non-null while a stream capture is open: launches are recorded, not run

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    if (g_capture) { g_capture->push_back(Recorded{grid, block, body}); return; }
    std::lock_guard<std::mutex> guard(g_launch_lock);
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || (size_t)grid.x * grid.y * grid.z == 0) return;
    while ((int)g_stacks.size() < nthreads) g_stacks.push_back((char*)malloc(kStack));
    g_fibers.assign(nthreads, Fiber());
    g_body = &body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_waves.assign((nthreads + kWaveSize - 1) / kWaveSize, Wave());
                g_block_alive = nthreads; g_block_arrived = 0;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = g_fibers[t];
                    f.done = false;
                    f.lane.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.lane.bid = dim3(bx, by, bz); f.lane.bdim = block; f.lane.gdim = grid;
                    f.wave = t / kWaveSize; f.lane_in_wave = t % kWaveSize;  // waves are consecutive linear thread ids, as on the GPU
                    ++g_waves[f.wave].alive;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = g_stacks[t]; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, trampoline, 0);
                }
                int remaining = nthreads;
                while (remaining > 0) {
                    remaining = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = g_fibers[t];
                        if (f.done) continue;
                        g_me = &f; cur = &f.lane;
                        swapcontext(&g_sched, &f.ctx);
                        if (!f.done) ++remaining;
                    }
                }
            }
    g_me = nullptr; cur = nullptr; g_body = nullptr;
}

}  // namespace hostsim

// ---- runtime: KH_HOSTSIM_DEVICES (default 1) synchronous devices, device memory = host memory ---------------------------------
// Several devices are modelled as far as the HOST LAYER can get them wrong: the current device is per thread, a stream and an event
// belong to the device that was current when they were created, an allocation to the device current at that moment;
// hipEventRecord refuses an event of another device than the stream's (hipErrorInvalidHandle, as HIP does), a launch on a stream of
// another device than the current one is refused, a NULL stream is the current device's.  (the double-buffered copies
// recorded per-thread events on streams of whatever device — found by review, because no test box has two GPUs.)
struct hostsim_stream { int device; };
struct hostsim_event { int device; };
namespace {
int device_count() {
    static const int n = [] { const char* e = getenv("KH_HOSTSIM_DEVICES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v > 16 ? 16 : v; }();
    return n;
}
thread_local int t_device = 0;
thread_local hipError_t t_last_error = hipSuccess;
int stream_device(hipStream_t s) { return s ? s->device : t_device; }

struct Alloc { hipMemoryType type; size_t bytes; int device; };
std::mutex g_mem_lock;
std::map<const void*, Alloc> g_allocs;  // base -> (kind, bytes, device): kh_pointer_domain, the staging ring's pinned-source test
hipError_t track(void** p, size_t n, hipMemoryType type) {
    *p = calloc(n ? n : 1, 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> g(g_mem_lock);
    g_allocs[*p] = Alloc{type, n ? n : 1, t_device};
    return hipSuccess;
}
hipError_t untrack(void* p) {
    if (!p) return hipSuccess;
    { std::lock_guard<std::mutex> g(g_mem_lock); g_allocs.erase(p); }
    free(p);
    return hipSuccess;
}
}  // namespace

namespace hostsim {
void launch_on(hipStream_t stream, dim3 grid, dim3 block, const std::function<void()>& body) {
    if (stream_device(stream) != t_device) { t_last_error = hipErrorInvalidHandle; return; }   // the launch does not happen
    launch(grid, block, body);
}
}  // namespace hostsim

const char* hipGetErrorString(hipError_t e) {
    return e == hipSuccess ? "success" : e == hipErrorNotSupported ? "not supported by the host simulator"
         : e == hipErrorInvalidHandle ? "invalid resource handle (event / stream / launch on another device than the one it belongs to)"
         : e == hipErrorInvalidDevice ? "invalid device ordinal" : "error";
}
hipError_t hipGetLastError() { const hipError_t e = t_last_error; t_last_error = hipSuccess; return e; }
hipError_t hipGetDeviceCount(int* n) { *n = device_count(); return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= device_count()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d) {
    if (d < 0 || d >= device_count()) return hipErrorInvalidDevice;
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "host simulator");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "x86 fibers");
    p->multiProcessorCount = 1; p->totalGlobalMem = (size_t)64 << 30;
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hostsim_stream{t_device}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // any device's event: how GPUs are ordered against each other
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hostsim_event{t_device}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { return e->device == stream_device(s) ? hipSuccess : hipErrorInvalidHandle; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)32 << 30; *t = (size_t)64 << 30; return hipSuccess; }
// stream capture: kernel launches between begin and end are recorded (arguments by value, like graph kernel nodes) and
// replayed by hipGraphLaunch; memory operations during a capture are not modelled (the API contract forbids them anyway)
struct hostsim_graph { std::vector<hostsim::Recorded> nodes; };
struct hostsim_graph_exec { std::vector<hostsim::Recorded> nodes; };
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
    if (hostsim::g_capture) return hipErrorInvalidValue;
    hostsim::g_capture = new std::vector<hostsim::Recorded>();
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
    *g = nullptr;
    if (!hostsim::g_capture) return hipErrorInvalidValue;
    *g = new hostsim_graph{std::move(*hostsim::g_capture)};
    delete hostsim::g_capture;
    hostsim::g_capture = nullptr;
    return hipSuccess;
}
hipError_t hipStreamGetDevice(hipStream_t s, hipDevice_t* device) { *device = stream_device(s); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* status) {
    *status = hostsim::g_capture ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
    return hipSuccess;
}
hipError_t hipGraphGetNodes(hipGraph_t g, hipGraphNode_t*, size_t* n) { *n = g->nodes.size(); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t*, char*, size_t) { *e = new hostsim_graph_exec{g->nodes}; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
    for (const auto& n : e->nodes) hostsim::launch(n.grid, n.block, n.body);
    return hipSuccess;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
hipError_t hipDeviceGetDefaultMemPool(hipMemPool_t* p, int) { *p = nullptr; return hipSuccess; }
hipError_t hipMemPoolSetAttribute(hipMemPool_t, hipMemPoolAttr, void*) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { return track(p, n, hipMemoryTypeDevice); }
hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { return track(p, n, hipMemoryTypeDevice); }
hipError_t hipFree(void* p) { return untrack(p); }
hipError_t hipFreeAsync(void* p, hipStream_t) { return untrack(p); }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return track(p, n, hipMemoryTypeHost); }
hipError_t hipHostFree(void* p) { return untrack(p); }
hipError_t hipMallocManaged(void** p, size_t n, unsigned) { return track(p, n, hipMemoryTypeManaged); }
hipError_t hipMemset(void* p, int v, size_t n) { if (n) memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { if (n) memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, w);
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    std::lock_guard<std::mutex> g(g_mem_lock);
    auto it = g_allocs.upper_bound(p);  // the allocation that starts at or before p
    if (it == g_allocs.begin()) return hipErrorInvalidValue;
    --it;
    if ((const char*)p >= (const char*)it->first + it->second.bytes) return hipErrorInvalidValue;   // past the end of that allocation: plain host memory
    a->type = it->second.type; a->device = it->second.device;
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
