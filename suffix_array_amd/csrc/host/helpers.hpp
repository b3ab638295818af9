// host/helpers.hpp -- persistent helper threads and NUMA placement for the host-pointer entry points (SURVEY.md 8e names
// the limiters of the 8-GPU batch: host DRAM for eight concurrent pinned copies, NUMA placement of the staging buffers).
//
// One pool of helper threads per NUMA node, created on first use and kept for the life of the process: the threads of node
// N run on N's CPUs (sched_setaffinity), so the pinned staging blocks they allocate are first-touched -- and the copies out
// of them are made -- next to the GPU's PCIe root.  The node of a device comes from
// /sys/bus/pci/devices/<bdf>/numa_node, its CPUs from /sys/devices/system/node/node<N>/cpulist.
// SA_AMD_NUMA=0 turns the placement off (one unpinned pool); SA_AMD_VERBOSE=2 prints the node chosen per call.
// A caller's work never waits for a helper to exist: parallel_for runs on the calling thread too.
#pragma once
#include "support.hpp"
#include "tuning.hpp"

#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <sched.h>
#include <unistd.h>

namespace sa {

// parses "0-63,128-191" into a cpu set; false when nothing could be read
static bool parse_cpulist(const char *s, cpu_set_t *set)
{
    CPU_ZERO(set);
    bool any = false;
    while (*s) {
        char *end = nullptr;
        const long a = strtol(s, &end, 10);
        if (end == s) break;
        long b = a;
        s = end;
        if (*s == '-') { b = strtol(s + 1, &end, 10); if (end == s + 1) break; s = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (c >= 0) { CPU_SET((int)c, set); any = true; }
        if (*s == ',') ++s; else break;
    }
    return any;
}

static bool read_small_file(const char *path, char *buf, size_t cap)
{
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const size_t got = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[got] = 0;
    return got > 0;
}

// NUMA node of a HIP device (-1: unknown, single-node host, or SA_AMD_NUMA=0); cached per device
static int device_numa_node(int device)
{
    static std::mutex mu;
    static std::map<int, int> cache;
    if (env_int("SA_AMD_NUMA", 1, 0, 1) == 0 || device < 0) return -1;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(device);
    if (it != cache.end()) return it->second;
    int node = -1;
    char bdf[64] = { 0 };
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) == hipSuccess) {
        for (char *p = bdf; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
        char path[160], buf[64];
        snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
        if (read_small_file(path, buf, sizeof(buf))) node = (int)strtol(buf, nullptr, 10);
    } else (void)hipGetLastError();
    if (node >= 0) {
        // a node whose CPU list cannot be read is of no use for placement
        char path[96], buf[4096];
        cpu_set_t set;
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        if (!read_small_file(path, buf, sizeof(buf)) || !parse_cpulist(buf, &set)) node = -1;
    }
    cache[device] = node;
    return node;
}

class HelperPool {
    struct Job {
        std::function<void(int)> fn;
        int count = 0;
        int next = 0;          // next task index to hand out
        int done = 0;
    };
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::deque<Job *> jobs_;
    std::vector<std::thread> threads_;
    int node_;
    int want_ = 0;             // threads the pool should have
    bool have_cpus_ = false;
    cpu_set_t cpus_;

    void worker()
    {
        if (have_cpus_) (void)sched_setaffinity(0, sizeof(cpus_), &cpus_);
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_work_.wait(lk, [&] { return !jobs_.empty(); });
            Job *j = jobs_.front();
            const int t = j->next++;
            if (j->next >= j->count) jobs_.pop_front();
            lk.unlock();
            j->fn(t);
            lk.lock();
            if (++j->done == j->count) cv_done_.notify_all();
        }
    }

public:
    explicit HelperPool(int node) : node_(node)
    {
        int cpus = (int)std::thread::hardware_concurrency();
        if (node >= 0) {
            char path[96], buf[4096];
            snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
            if (read_small_file(path, buf, sizeof(buf)) && parse_cpulist(buf, &cpus_)) { have_cpus_ = true; cpus = CPU_COUNT(&cpus_); }
        }
        // enough for one download of twelve slices (callers of one device download one at a time) plus a page-touching job -- PER DEVICE
        // of the node: the devices that hang off one socket (four of eight on the evidence hosts) share this pool, and one
        // sa_amd_saca_batch call drives them all at once; never more than the node has CPUs (two left to the callers)
        want_ = (int)env_int("SA_AMD_HELPER_THREADS", 16, 0, 256);
        if (want_ > 0 && node >= 0) {
            int ndev = 0, here = 0;
            if (hipGetDeviceCount(&ndev) == hipSuccess) {
                for (int d = 0; d < ndev; ++d) here += device_numa_node(d) == node ? 1 : 0;
            } else (void)hipGetLastError();
            if (here > 1) want_ *= here;
            if (want_ > 128) want_ = 128;
        }
        if (cpus > 2 && want_ > cpus - 2) want_ = cpus - 2;
        else if (cpus > 0 && want_ > cpus) want_ = cpus;
    }
    int node() const { return node_; }
    int helpers() const { return want_; }      // threads the pool has or will start (0: every job runs on its caller)

    // fn(0) .. fn(count-1), each once, on the helpers and on the calling thread; returns when all have run.
    // Tasks must not throw.  Helper threads are started on first use; if none can be started the caller does all the work.
    void parallel_for(int count, const std::function<void(int)> &fn)
    {
        if (count <= 0) return;
        if (count == 1) { fn(0); return; }
        Job job;
        job.fn = fn; job.count = count;
        {
            std::lock_guard<std::mutex> lk(mu_);
            while ((int)threads_.size() < want_) {
                try { threads_.emplace_back([this] { worker(); }); threads_.back().detach(); }
                catch (...) { want_ = (int)threads_.size(); break; }
            }
            jobs_.push_back(&job);
        }
        cv_work_.notify_all();
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            if (job.next < job.count) {                              // the caller takes tasks like a helper does
                const int t = job.next++;
                if (job.next >= job.count) {
                    for (auto it = jobs_.begin(); it != jobs_.end(); ++it) if (*it == &job) { jobs_.erase(it); break; }
                }
                lk.unlock();
                fn(t);
                lk.lock();
                ++job.done;
            } else {
                cv_done_.wait(lk, [&] { return job.done == job.count; });
                break;
            }
        }
    }
    // The same in two halves: start() hands fn(0) .. fn(count-1) to the helpers and returns; finish() takes what is left
    // on the calling thread and returns when all have run.  The handle (and what fn refers to) must outlive finish().
    struct Async { Job job; bool queued = false; };
    void start(Async &a, int count, const std::function<void(int)> &fn)
    {
        a.job.fn = fn; a.job.count = count; a.job.next = 0; a.job.done = 0; a.queued = false;
        if (count <= 0) return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            while ((int)threads_.size() < want_) {
                try { threads_.emplace_back([this] { worker(); }); threads_.back().detach(); }
                catch (...) { want_ = (int)threads_.size(); break; }
            }
            jobs_.push_back(&a.job);
            a.queued = true;
        }
        cv_work_.notify_all();
    }
    void finish(Async &a)
    {
        if (!a.queued) return;
        Job &job = a.job;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            if (job.next < job.count) {
                const int t = job.next++;
                if (job.next >= job.count) {
                    for (auto it = jobs_.begin(); it != jobs_.end(); ++it) if (*it == &job) { jobs_.erase(it); break; }
                }
                lk.unlock();
                job.fn(t);
                lk.lock();
                ++job.done;
            } else {
                cv_done_.wait(lk, [&] { return job.done == job.count; });
                break;
            }
        }
        a.queued = false;
    }
    // runs fn once on a helper thread of this pool (its CPU affinity decides where first-touched pages land);
    // on the calling thread when the pool has no helpers
    void run_on_helper(const std::function<void()> &fn)
    {
        bool have;
        {
            std::lock_guard<std::mutex> lk(mu_);
            while ((int)threads_.size() < (want_ > 0 ? 1 : 0)) {
                try { threads_.emplace_back([this] { worker(); }); threads_.back().detach(); }
                catch (...) { want_ = 0; break; }
            }
            have = !threads_.empty();
        }
        if (!have) { fn(); return; }
        Job job;
        job.fn = [&](int) { fn(); };
        job.count = 1;
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(&job);
        }
        cv_work_.notify_all();
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return job.done == job.count; });
    }
};

// the pool of a NUMA node (-1: the unpinned pool); pools live as long as the process (their threads are detached and
// sleep on a condition variable; no HIP call is made at exit)
static HelperPool &helper_pool(int node)
{
    static std::mutex mu;
    static std::map<int, HelperPool *> pools;
    std::lock_guard<std::mutex> lk(mu);
    auto it = pools.find(node);
    if (it != pools.end()) return *it->second;
    HelperPool *p = new HelperPool(node);
    pools[node] = p;
    return *p;
}

}  // namespace sa
