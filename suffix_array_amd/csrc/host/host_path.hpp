// host/host_path.hpp -- host pointers in, host pointers out: the contract of `saca()` (reference src/saca.rs:9-15),
// i.e. what `SuffixArray::new` / `::set` (src/sa.rs:23-33) hand over: a borrowed text and a caller-owned Vec<u32>.
//
// Device memory comes from a process-wide pool of blocks (one block = text + SA + workspace of one build), so neither
// repeated calls nor the short-lived worker threads of sa_amd_saca_batch pay hipMalloc / hipFree per call; the pool
// retains at most SA_AMD_CACHE_MAX_BYTES (default 64 GiB of the 288 GB) and sa_amd_release_cache() empties it.
// The suffix array travels back either with one hipMemcpy into the caller's pageable buffer, or -- large arrays --
// chunk by chunk through pinned staging buffers while helper threads move finished chunks into the caller's buffer
// (a freshly allocated Vec<u32>, as SuffixArray::new makes, is page-faulted in by several threads instead of one).
#pragma once
#include "pipeline.hpp"

#include <chrono>
#include <condition_variable>
#include <mutex>

namespace sa {

static int pick_device()
{
    const char *e = getenv("SA_AMD_DEVICE");
    if (!e || !*e) return -1;              // -1: keep the calling thread's current device
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    return end == e ? -1 : (int)v;
}

static size_t cache_limit()
{
    return (size_t)env_int("SA_AMD_CACHE_MAX_BYTES", (int64_t)64 << 30, 0, (int64_t)1 << 50);
}

static double wall_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- pool of device blocks, streams and pinned staging buffers (process-wide, mutex-protected) ----
struct DevBlock { int device = -1; void *p = nullptr; size_t bytes = 0; };
struct PinBlock { void *p = nullptr; size_t bytes = 0; };

class ResourcePool {
    std::mutex mu_;
    std::vector<DevBlock> blocks_;                 // free device blocks
    std::vector<std::pair<int, hipStream_t>> streams_;
    std::vector<PinBlock> pinned_;
    size_t retained_ = 0;

public:
    // a free block of `device` with at least `need` bytes (the smallest such), else a new allocation
    int acquire(int device, size_t need, DevBlock *out)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
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
        size_t want = need + need / 8;                               // slack: a slowly growing series of texts reuses the block
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
        const size_t limit = cache_limit();
        std::vector<DevBlock> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (b.bytes > limit) drop.push_back(b);
            else {
                // evict the largest blocks until the newcomer fits (it is the size the caller is working at)
                while (retained_ + b.bytes > limit && !blocks_.empty()) {
                    int big = 0;
                    for (int i = 1; i < (int)blocks_.size(); ++i) if (blocks_[i].bytes > blocks_[big].bytes) big = i;
                    retained_ -= blocks_[big].bytes;
                    drop.push_back(blocks_[big]);
                    blocks_.erase(blocks_.begin() + big);
                }
                blocks_.push_back(b);
                retained_ += b.bytes;
            }
        }
        for (auto &d : drop) (void)hipFree(d.p);
    }
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
    int pinned(size_t bytes, PinBlock *out)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = 0; i < (int)pinned_.size(); ++i)
                if (pinned_[i].bytes >= bytes) { *out = pinned_[i]; pinned_.erase(pinned_.begin() + i); return SA_AMD_OK; }
        }
        void *p = nullptr;
        const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); return hip_status(e); }
        out->p = p; out->bytes = bytes;
        return SA_AMD_OK;
    }
    void release_pinned(const PinBlock &b)
    {
        if (!b.p) return;
        std::lock_guard<std::mutex> lk(mu_);
        pinned_.push_back(b);
    }
    void clear()
    {
        std::vector<DevBlock> drop; std::vector<PinBlock> pdrop; std::vector<std::pair<int, hipStream_t>> sdrop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            drop.swap(blocks_); pdrop.swap(pinned_); sdrop.swap(streams_);
            retained_ = 0;
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

// phases of the last host-pointer call of this thread, in milliseconds (sa_amd_last_host_timing)
struct HostTiming { double acquire = 0, h2d = 0, build = 0, d2h = 0, release = 0, total = 0; int staged = 0; };
static thread_local HostTiming g_host_timing;

// ---- device -> caller's pageable buffer through pinned staging + helper threads ----
constexpr size_t STAGE_BYTES = (size_t)16 << 20;
constexpr int STAGE_COUNT = 3;

static int staged_download(void *dst_host, const void *dsrc, size_t bytes, hipStream_t st, int copy_threads, int device)
{
    PinBlock stage[STAGE_COUNT];
    hipEvent_t ev[STAGE_COUNT] = { nullptr, nullptr, nullptr };
    int rc = SA_AMD_OK;
    for (int i = 0; i < STAGE_COUNT && rc == SA_AMD_OK; ++i) rc = pool().pinned(STAGE_BYTES, &stage[i]);
    for (int i = 0; i < STAGE_COUNT && rc == SA_AMD_OK; ++i) rc = hip_status(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    const size_t nchunk = (bytes + STAGE_BYTES - 1) / STAGE_BYTES;
    auto len = [&](size_t c) { return c + 1 < nchunk ? STAGE_BYTES : bytes - c * STAGE_BYTES; };
    if (rc == SA_AMD_OK) {
        // issued: chunks whose device->stage copy has been enqueued (their event is recorded); drained[c]: helper threads done with chunk c
        std::mutex mu;
        std::condition_variable cv;
        size_t issued = 0;
        bool failed = false;
        std::vector<int> drained(nchunk, 0);
        const int T = copy_threads < 1 ? 1 : copy_threads;
        auto helper = [&](int t) {
            (void)hipSetDevice(device);
            for (size_t c = 0; c < nchunk; ++c) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return issued > c || failed; });
                    if (failed) return;
                }
                if (hipEventSynchronize(ev[c % STAGE_COUNT]) != hipSuccess) {
                    std::lock_guard<std::mutex> lk(mu); failed = true; cv.notify_all(); return;
                }
                const size_t L = len(c), per = (((L + T - 1) / T) + 4095) & ~(size_t)4095;     // whole pages per helper, the last one takes the rest
                const size_t b = (size_t)t * per, e = b + per < L ? b + per : L;
                if (b < e) memcpy((char *)dst_host + c * STAGE_BYTES + b, (const char *)stage[c % STAGE_COUNT].p + b, e - b);
                { std::lock_guard<std::mutex> lk(mu); ++drained[c]; }
                cv.notify_all();
            }
        };
        std::vector<std::thread> threads;
        try {
            for (int t = 0; t < T; ++t) threads.emplace_back(helper, t);
        } catch (...) {
            { std::lock_guard<std::mutex> lk(mu); failed = true; }
            cv.notify_all();
            rc = SA_AMD_ENOMEM;
        }
        const int started = (int)threads.size();
        for (size_t c = 0; c < nchunk && rc == SA_AMD_OK; ++c) {
            if (c >= (size_t)STAGE_COUNT) {                           // the stage's previous tenant must have been moved out
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return drained[c - STAGE_COUNT] >= started || failed; });
                if (failed) { rc = SA_AMD_EHIP; break; }
            }
            rc = hip_status(hipMemcpyAsync(stage[c % STAGE_COUNT].p, (const char *)dsrc + c * STAGE_BYTES, len(c), hipMemcpyDeviceToHost, st));
            if (rc == SA_AMD_OK) rc = hip_status(hipEventRecord(ev[c % STAGE_COUNT], st));
            if (rc != SA_AMD_OK) { std::lock_guard<std::mutex> lk(mu); failed = true; }
            else { std::lock_guard<std::mutex> lk(mu); issued = c + 1; }
            cv.notify_all();
        }
        for (auto &t : threads) t.join();
        if (rc == SA_AMD_OK && failed) rc = SA_AMD_EHIP;
    }
    for (int i = 0; i < STAGE_COUNT; ++i) {
        if (ev[i]) (void)hipEventDestroy(ev[i]);
        pool().release_pinned(stage[i]);
    }
    return rc;
}

// host buffers in, host buffers out; with_sentinel writes SA[0] = n too (saca layout)
static int build_host(const uint8_t *T, uint32_t *SA_host, int32_t n, bool with_sentinel, int device)
{
    if (n < 0 || (n > 0 && (!T || !SA_host)) || (with_sentinel && !SA_host)) return SA_AMD_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SA_AMD_ENODEVICE;
    if (device >= ndev) return SA_AMD_EINVAL;
    if (n == 0) { if (with_sentinel) SA_host[0] = 0; return SA_AMD_OK; }
    DeviceGuard guard(device);                                    // the caller's current device is restored on return
    if (guard.rc != SA_AMD_OK) return guard.rc;
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    HostTiming tm;
    const double t_begin = wall_ms();
    const size_t wb = (size_t)sa_amd_workspace_bytes(n);
    const size_t tb = align_up((size_t)n, 256), sb = align_up(((size_t)n + 1) * 4, 256);
    const size_t need = tb + sb + wb;
    DevBlock blk;
    hipStream_t st = nullptr;
    int rc = pool().stream(cur, &st);
    if (rc != SA_AMD_OK) return rc;
    rc = pool().acquire(cur, need, &blk);
    if (rc != SA_AMD_OK) { pool().release_stream(cur, st); return rc; }
    double t0 = wall_ms();
    tm.acquire = t0 - t_begin;
    uint8_t *dT = (uint8_t *)blk.p;
    uint32_t *dSA = (uint32_t *)((char *)blk.p + tb);
    void *dW = (char *)blk.p + tb + sb;
    rc = hip_status(hipMemcpyAsync(dT, T, (size_t)n, hipMemcpyHostToDevice, st));
    if (rc == SA_AMD_OK) rc = hip_status(hipStreamSynchronize(st));
    double t1 = wall_ms();
    tm.h2d = t1 - t0;
    if (rc == SA_AMD_OK) rc = build_device(dT, dSA, n, dW, (int64_t)wb, st, nullptr);
    t0 = wall_ms();
    tm.build = t0 - t1;
    if (rc == SA_AMD_OK) {
        const uint32_t *src = with_sentinel ? dSA : dSA + 1;
        const size_t out_bytes = ((size_t)n + (with_sentinel ? 1 : 0)) * 4;
        const int copy_threads = (int)env_int("SA_AMD_COPY_THREADS", 8, 0, 32);      // 0: plain hipMemcpy into the caller's buffer
        const size_t staged_min = (size_t)env_int("SA_AMD_STAGED_MIN_BYTES", (int64_t)64 << 20, 0, (int64_t)1 << 40);
        if (copy_threads > 0 && out_bytes >= staged_min) {
            tm.staged = copy_threads;
            rc = staged_download(SA_host, src, out_bytes, st, copy_threads, cur);
        } else {
            rc = hip_status(hipMemcpyAsync(SA_host, src, out_bytes, hipMemcpyDeviceToHost, st));
        }
    }
    const int rs = hip_status(hipStreamSynchronize(st));       // also drains the stream after a failure
    if (rc == SA_AMD_OK) rc = rs;
    t1 = wall_ms();
    tm.d2h = t1 - t0;
    pool().release(blk);
    pool().release_stream(cur, st);
    tm.release = wall_ms() - t1;
    tm.total = wall_ms() - t_begin;
    g_host_timing = tm;
    if (env_int("SA_AMD_VERBOSE", 0, 0, 9) >= 2)
        fprintf(stderr, "suffix_array_amd: n=%d acquire %.2f h2d %.2f build %.2f d2h %.2f (staged %d) release %.2f total %.2f ms\n", n,
                tm.acquire, tm.h2d, tm.build, tm.d2h, tm.staged, tm.release, tm.total);
    return rc;
}

}  // namespace sa
