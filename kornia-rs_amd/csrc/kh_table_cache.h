// Device-resident lookup tables (resize contribution tables, bilateral weights) cached per key, with safe eviction.
//
// The reference caches its tap / LUT tables per (device, bits) behind a mutex with a sync-before-publish rule and never
// evicts (P/filter/cuda.rs:51-79, P/resize/cuda.rs:151-190).  A batch server that sees many geometries (random-resize
// augmentation, multi-threaded loaders) needs a bound, and a bound needs an eviction that cannot pull a table from under
// a launch.  Rules here:
//   * a caller holds a LEASE (shared ownership) from lookup until it has enqueued its launch and called used_on(stream),
//     which records the entry's event behind that launch: a table is freed only when the cache AND every lease have let
//     go of it, after a host wait on that event — never while a thread still has to launch with it;
//   * eviction is LRU, one entry at a time, only of entries nobody leases and no graph pinned;
//   * a table first used while its stream is being captured is PINNED for the life of the process: a replayed graph bakes
//     the pointer;
//   * a cache MISS during capture is refused with a clear message — building a table allocates and copies synchronously,
//     which is illegal under capture: warm the operator up once before capturing.
#pragma once

#include <atomic>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "kh_common.h"

namespace kh {

struct DevTable {
    void* dev = nullptr;
    size_t bytes = 0;
    int meta[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // small per-table facts (tap count, radius ...)
    std::atomic<bool> pinned{false};  // referenced by a captured graph: never evicted (read by the evictor, written by users)
    int device = 0;
    DevTable() = default;
    DevTable(const DevTable&) = delete;
    DevTable& operator=(const DevTable&) = delete;
    ~DevTable() {
        // host-wait for the last launch of EVERY stream that used the table, then free it
        for (auto& kv : last_use_) { (void)hipEventSynchronize(kv.second); (void)hipEventDestroy(kv.second); }
        if (dev) (void)hipFree(dev);
    }
    // Call after the launch that reads the table has been enqueued on `stream`.  One event PER STREAM: two streams (or threads)
    // leasing the same table each leave their own fence — a single re-recorded event covered only the latest user.
    void used_on(hipStream_t stream) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            pinned.store(true);  // the graph holds the pointer; an event cannot be recorded into a capture for a later host wait
            return;
        }
        std::lock_guard<std::mutex> lock(use_mu_);
        auto it = last_use_.find(stream);
        if (it == last_use_.end() && last_use_.size() >= kMaxFences) {
            // A long-lived table in a process that creates and destroys many streams must not collect one event per stream handle it
            // ever saw: fences whose launch has completed protect nothing any more — drop them before adding another.
            for (auto f = last_use_.begin(); f != last_use_.end();) {
                if (hipEventQuery(f->second) == hipSuccess) { (void)hipEventDestroy(f->second); f = last_use_.erase(f); }
                else { (void)hipGetLastError(); ++f; }
            }
        }
        if (it == last_use_.end()) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {  // no fence possible: make the user wait now
                (void)hipGetLastError();
                (void)hipStreamSynchronize(stream);
                return;
            }
            it = last_use_.emplace(stream, ev).first;
        }
        (void)hipEventRecord(it->second, stream);
    }

private:
    static constexpr size_t kMaxFences = 8;
    std::mutex use_mu_;
    std::map<hipStream_t, hipEvent_t> last_use_;
};
using TableLease = std::shared_ptr<DevTable>;

template <typename Key>
class TableCache {
public:
    explicit TableCache(size_t max_entries) : max_(max_entries) {}

    // `build(table)` fills dev / bytes / meta with a blocking upload into a fresh allocation (sync-before-publish).
    template <typename Build>
    int32_t lookup(const Key& key, hipStream_t stream, const char* what, Build&& build, TableLease& out) {
        std::vector<TableLease> doomed;  // evicted tables die AFTER the lock is released: their destructor host-waits and hipFrees
        std::lock_guard<std::mutex> lock(mu_);
        auto it = map_.find(key);
        if (it != map_.end()) {
            lru_.splice(lru_.begin(), lru_, it->second.pos);  // most recently used first
            out = it->second.tab;
            return KH_OK;
        }
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(KH_ERR_INVALID_ARG, "%s: the lookup table for this geometry is not on the device yet and cannot be built while the "
                                            "stream is being captured (it needs an allocation and a blocking copy) — run the operator once "
                                            "with these parameters before kh_graph_capture_begin", what);
        evict_locked(doomed);
        auto tab = std::make_shared<DevTable>();
        (void)hipGetDevice(&tab->device);
        if (int32_t rc = build(*tab)) return rc;
        lru_.push_front(key);
        map_.emplace(key, Slot{tab, lru_.begin()});
        out = tab;
        return KH_OK;
    }
    size_t size() {
        std::lock_guard<std::mutex> lock(mu_);
        return map_.size();
    }

private:
    struct Slot { TableLease tab; typename std::list<Key>::iterator pos; };
    // Drop least-recently-used entries nobody else holds until there is room.  The evicted tables are handed to the caller, which
    // lets go of them outside the cache mutex: the DevTable destructor host-waits for the last launch of every stream that used
    // the table and calls hipFree (a device-wide sync) — neither may stall other threads' lookups.
    void evict_locked(std::vector<TableLease>& doomed) {
        if (map_.size() < max_) return;
        for (auto pos = lru_.end(); pos != lru_.begin() && map_.size() >= max_;) {
            --pos;
            auto it = map_.find(*pos);
            if (it->second.tab.use_count() == 1 && !it->second.tab->pinned.load()) {
                doomed.push_back(std::move(it->second.tab));
                map_.erase(it);
                pos = lru_.erase(pos);
            }
        }
        // everything leased or pinned: grow past the bound rather than free a live table
    }
    std::mutex mu_;
    std::map<Key, Slot> map_;
    std::list<Key> lru_;
    size_t max_;
};

}  // namespace kh
