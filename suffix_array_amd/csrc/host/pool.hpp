// host/pool.hpp -- process-wide pool of device blocks, streams and pinned host blocks (mutex-protected).
// One device block = text + SA + workspace of one build, so neither repeated calls of the host-pointer entry points
// (the contract of `saca()`, reference src/saca.rs:9-15) nor the worker threads of sa_amd_saca_batch pay hipMalloc /
// hipFree per call.  What the pool keeps PER DEVICE follows the work: at most twice the largest block any of the device's last
// eight builds asked for (two workers of a batch alternate on two blocks), never more than SA_AMD_CACHE_MAX_BYTES (default
// 128 GiB of a device's 288 GB: one 1 GiB text is a 58 GiB block) -- a process that indexed one 1 GiB text and goes on with
// 64 MiB ones gives the 58 GiB back after eight of them instead of holding them for its lifetime -- and blocks that have not
// been used for SA_AMD_CACHE_IDLE_MS (default 60 s: a block of 14 GiB costs half a second of hipMalloc to get back, and the copy engine reads a re-allocated block at half rate, host_path.hpp) are freed by the next call that touches the pool.
// sa_amd_release_cache() empties it.  Pinned blocks remember the NUMA node they were first touched on (helpers.hpp).
#pragma once
#include "helpers.hpp"

#include <mutex>

namespace sa {

static size_t cache_limit()
{
    return (size_t)env_int("SA_AMD_CACHE_MAX_BYTES", (int64_t)128 << 30, 0, (int64_t)1 << 50);
}

// ---- pool of device blocks, streams and pinned staging buffers (process-wide, mutex-protected) ----
struct DevBlock { int device = -1; void *p = nullptr; size_t bytes = 0; uint64_t stamp = 0; double idle_since = 0; };      // stamp: order in which blocks were handed back; idle_since: when (ms)
struct PinBlock { void *p = nullptr; size_t bytes = 0; int node = -1; uint64_t stamp = 0; };      // node: NUMA node its pages were first touched on (-1: wherever); stamp: when it was last handed back

class ResourcePool {
    std::mutex mu_;
    std::vector<DevBlock> blocks_;                 // free device blocks
    std::vector<std::pair<int, hipStream_t>> streams_;
    std::vector<PinBlock> pinned_;
    size_t retained_ = 0, pinned_retained_ = 0;
    uint64_t clock_ = 0;
    std::map<int, std::deque<std::pair<double, size_t>>> recent_;     // per device: (time, bytes) of what its builds of the last SA_AMD_CACHE_IDLE_MS asked for

public:
    // a free block of `device` with at least `need` bytes (the smallest such), else a new allocation
    int acquire(int device, size_t need, DevBlock *out)
    {
        trim_idle();
        {
            std::lock_guard<std::mutex> lk(mu_);
            // What the device's builds of the last SA_AMD_CACHE_IDLE_MS asked for (not "the last eight": a dozen 1 MiB texts between
            // two 256 MiB ones dropped the large block, and the block allocated in its place downloads at half the rate -- see
            // host_path.hpp, k_copy_to_host; a block nobody has asked for in that time goes back anyway, trim_idle)
            std::deque<std::pair<double, size_t>> &rq = recent_[device];
            const double now = now_ms(), keep_ms = idle_ms();
            while (!rq.empty() && (now - rq.front().first > keep_ms || rq.size() >= 256)) rq.pop_front();
            if (!rq.empty() && rq.back().second == need) rq.back().first = now;      // (a run of equal requests is one entry)
            else rq.push_back(std::make_pair(now, need));
            int best = -1;
            for (int i = 0; i < (int)blocks_.size(); ++i)
                if (blocks_[i].device == device && blocks_[i].bytes >= need && (best < 0 || blocks_[i].bytes < blocks_[best].bytes)) best = i;
            if (best >= 0) {
                *out = blocks_[best];
                retained_ -= out->bytes;
                blocks_.erase(blocks_.begin() + best);
                return SA_AMD_OK;
            }
        }
        // slack: a slowly growing series of texts reuses the block -- but never so much that a block which fits the cache
        // limit is pushed over it and dropped on release (a 1 GiB text needs 57.9 GiB: with an eighth on top it was 65.1 GiB
        // against the 64 GiB limit of round 2, and every call of that size paid hipMalloc + hipFree)
        size_t want = need + (need / 8 < ((size_t)1 << 30) ? need / 8 : ((size_t)1 << 30));      // (at most 1 GiB of slack)
        {
            const size_t limit = cache_limit();
            if (want > limit) want = need > limit ? need : limit;
        }
        if (device_limit_locked(device) < want) want = need;          // (slack that would be dropped on release anyway)
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            trim(device, 0);                                         // give the pool's memory back and try the exact size
            want = need;
            e = hipMalloc(&p, want);
            if (e != hipSuccess) { (void)hipGetLastError(); return hip_status(e); }
        }
        out->device = device; out->p = p; out->bytes = want;
        return SA_AMD_OK;
    }
    void release(const DevBlock &b)
    {
        if (!b.p) return;
        std::vector<DevBlock> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            const size_t limit = device_limit(b.device);
            if (b.bytes > limit) drop.push_back(b);
            else {
                // evict the device's blocks that have been idle longest until the newcomer fits (it is the size the caller is working
                // at).  Not "the largest first": two workers of a batch alternate on two large blocks next to an older small one, and
                // dropping the large idle block made the next call pay a 30 GiB hipMalloc -- 2 s, with the other worker's
                // copies stalled behind it (profiles/r03_host_path.txt)
                for (;;) {
                    size_t held = 0;
                    int old = -1;
                    for (int i = 0; i < (int)blocks_.size(); ++i)
                        if (blocks_[i].device == b.device) {
                            held += blocks_[i].bytes;
                            if (old < 0 || blocks_[i].stamp < blocks_[old].stamp) old = i;
                        }
                    if (held + b.bytes <= limit || old < 0) break;
                    retained_ -= blocks_[old].bytes;
                    drop.push_back(blocks_[old]);
                    blocks_.erase(blocks_.begin() + old);
                }
                DevBlock kept = b;
                kept.stamp = ++clock_;
                kept.idle_since = now_ms();
                blocks_.push_back(kept);
                retained_ += b.bytes;
            }
        }
        for (auto &d : drop) (void)hipFree(d.p);
    }
    // blocks nobody has asked for in SA_AMD_CACHE_IDLE_MS go back to the device (checked by whoever uses the pool next: there is no
    // background thread -- a process that never calls again keeps its last blocks until sa_amd_release_cache or exit)
    static double idle_ms() { return (double)env_int("SA_AMD_CACHE_IDLE_MS", 60000, 0, (int64_t)1 << 40); }
    void trim_idle()
    {
        const double idle_ms = ResourcePool::idle_ms();
        if (idle_ms <= 0) return;
        const double now = now_ms();
        std::vector<DevBlock> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = (int)blocks_.size() - 1; i >= 0; --i)
                if (now - blocks_[i].idle_since > idle_ms) {
                    retained_ -= blocks_[i].bytes;
                    drop.push_back(blocks_[i]);
                    blocks_.erase(blocks_.begin() + i);
                }
        }
        for (auto &d : drop) (void)hipFree(d.p);
    }
    // what the pool may keep of a device's memory (mu_ held): twice the largest recent request, at least 256 MiB, at most the cap
    size_t device_limit(int device)
    {
        size_t big = 0;
        auto it = recent_.find(device);
        if (it != recent_.end()) for (const auto &v : it->second) big = v.second > big ? v.second : big;
        size_t lim = 2 * (big + (big / 8 < ((size_t)1 << 30) ? big / 8 : ((size_t)1 << 30))) + ((size_t)256 << 20);      // (two blocks with acquire()'s slack)
        if (lim < ((size_t)256 << 20)) lim = (size_t)256 << 20;
        const size_t cap = cache_limit();
        return lim < cap ? lim : cap;
    }
    size_t device_limit_locked(int device) { std::lock_guard<std::mutex> lk(mu_); return device_limit(device); }
    void trim(int device, size_t keep_bytes)                          // device < 0: all devices
    {
        std::vector<DevBlock> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = (int)blocks_.size() - 1; i >= 0 && retained_ > keep_bytes; --i)
                if (device < 0 || blocks_[i].device == device) {
                    retained_ -= blocks_[i].bytes;
                    drop.push_back(blocks_[i]);
                    blocks_.erase(blocks_.begin() + i);
                }
        }
        for (auto &d : drop) (void)hipFree(d.p);
    }
    int stream(int device, hipStream_t *out)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = 0; i < (int)streams_.size(); ++i)
                if (streams_[i].first == device) { *out = streams_[i].second; streams_.erase(streams_.begin() + i); return SA_AMD_OK; }
        }
        return hip_status(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    }
    void release_stream(int device, hipStream_t s)
    {
        if (!s) return;
        std::lock_guard<std::mutex> lk(mu_);
        streams_.push_back(std::make_pair(device, s));
    }
    // a pinned host block of at least `bytes`, first-touched on NUMA node `node` (-1: no placement).  New blocks of a node
    // are allocated by a helper thread that runs on that node's CPUs (helpers.hpp): hipHostMalloc pins -- touches -- the
    // pages from the allocating thread, and the default memory policy puts them on the node it runs on.
    // Blocks are PORTABLE and MAPPED (hipHostMallocPortable | hipHostMallocMapped): a block allocated while device A was
    // current may be handed to a caller on device B of the same NUMA node, whose kernels read and write it zero-copy (the
    // small-text path) and whose DMA engines target it (staging) -- portable makes it pinned for every device's context,
    // mapped gives every device a pointer to it, so the pool needs no device key.  The smallest free block that fits is
    // taken, and only one that is not more than four times the request (1 MiB for tiny requests): the 4 KiB read-back
    // buffer of a thread does not sit on a 96 MiB small-batch chunk for the thread's lifetime.  Retained pinned memory is
    // bounded (SA_AMD_PINNED_MAX_BYTES, default 2 GiB): beyond it the blocks that have been idle longest are freed.
    static size_t pinned_size_class(size_t bytes)
    {
        if (bytes <= ((size_t)1 << 20)) { size_t c = 4096; while (c < bytes) c <<= 1; return c; }
        return (bytes + (((size_t)4 << 20) - 1)) & ~(((size_t)4 << 20) - 1);
    }
    int pinned(size_t bytes, int node, int device, PinBlock *out)
    {
        const size_t want = pinned_size_class(bytes);
        {
            std::lock_guard<std::mutex> lk(mu_);
            const size_t most = want * 4 > ((size_t)1 << 20) ? want * 4 : ((size_t)1 << 20);
            int best = -1;
            for (int i = 0; i < (int)pinned_.size(); ++i)
                if (pinned_[i].bytes >= bytes && pinned_[i].bytes <= most && pinned_[i].node == node &&
                    (best < 0 || pinned_[i].bytes < pinned_[best].bytes)) best = i;
            if (best >= 0) {
                *out = pinned_[best];
                pinned_retained_ -= out->bytes;
                pinned_.erase(pinned_.begin() + best);
                return SA_AMD_OK;
            }
        }
        void *p = nullptr;
        hipError_t e = hipErrorUnknown;
        auto alloc = [&]() {
            if (device >= 0) (void)hipSetDevice(device);
            e = hipHostMalloc(&p, want, hipHostMallocPortable | hipHostMallocMapped);
        };
        if (node >= 0) helper_pool(node).run_on_helper(alloc); else alloc();
        if (e != hipSuccess) { (void)hipGetLastError(); return hip_status(e); }
        out->p = p; out->bytes = want; out->node = node; out->stamp = 0;
        return SA_AMD_OK;
    }
    void release_pinned(const PinBlock &b)
    {
        if (!b.p) return;
        const size_t limit = (size_t)env_int("SA_AMD_PINNED_MAX_BYTES", (int64_t)2 << 30, 0, (int64_t)1 << 44);
        std::vector<PinBlock> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (b.bytes > limit) drop.push_back(b);
            else {
                while (pinned_retained_ + b.bytes > limit && !pinned_.empty()) {
                    int old = 0;
                    for (int i = 1; i < (int)pinned_.size(); ++i) if (pinned_[i].stamp < pinned_[old].stamp) old = i;
                    pinned_retained_ -= pinned_[old].bytes;
                    drop.push_back(pinned_[old]);
                    pinned_.erase(pinned_.begin() + old);
                }
                PinBlock kept = b;
                kept.stamp = ++clock_;
                pinned_.push_back(kept);
                pinned_retained_ += b.bytes;
            }
        }
        for (auto &d : drop) (void)hipHostFree(d.p);
    }
    void clear()
    {
        std::vector<DevBlock> drop; std::vector<PinBlock> pdrop; std::vector<std::pair<int, hipStream_t>> sdrop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            drop.swap(blocks_); pdrop.swap(pinned_); sdrop.swap(streams_);
            retained_ = 0; pinned_retained_ = 0;
            recent_.clear();
        }
        for (auto &d : drop) (void)hipFree(d.p);
        for (auto &d : pdrop) (void)hipHostFree(d.p);
        for (auto &s : sdrop) (void)hipStreamDestroy(s.second);
    }
};
static ResourcePool &pool()
{
    static ResourcePool *p = new ResourcePool();      // intentionally never destroyed: no HIP calls during static destruction
    return *p;
}

}  // namespace sa
