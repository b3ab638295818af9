// host/support.hpp -- status mapping, launch checks, per-kernel event timing, small RAII helpers.
#pragma once
#include "../kernels/common.hpp"
#include "../kernels/keys.hpp"
#include "../kernels/radix_sort.hpp"
#include "../kernels/onesweep.hpp"
#include "../kernels/rerank.hpp"
#include "../kernels/refine.hpp"
#include "../kernels/bucket_sort.hpp"
#include "../kernels/isa.hpp"
#include "../kernels/extras.hpp"
#include "../kernels/small.hpp"
#ifdef SA_AMD_DIAG
#include "../kernels/sample_sort.hpp"
#include "../kernels/radix_sort_diag.hpp"
#include "../kernels/induce_proto.hpp"
#endif
#include "../../../include/suffix_array_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <thread>
#include <vector>
#include <chrono>
#include <sys/syscall.h>
#include <unistd.h>

namespace sa {

static inline double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool debug_sync()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("SA_AMD_DEBUG_SYNC"); v = (e && *e && *e != '0') ? 1 : 0; }
    return v == 1;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            if (getenv("SA_AMD_VERBOSE"))                                                          \
                fprintf(stderr, "suffix_array_amd: %s -> %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? SA_AMD_ENOMEM : SA_AMD_EHIP;                        \
        }                                                                                          \
    } while (0)

#define LAUNCH_CHECK(st)                                                                           \
    do {                                                                                           \
        HIP_TRY(hipGetLastError());                                                                \
        if (debug_sync()) HIP_TRY(hipStreamSynchronize(st));                                       \
    } while (0)

inline int hip_status(hipError_t e) { return e == hipSuccess ? SA_AMD_OK : (e == hipErrorOutOfMemory ? SA_AMD_ENOMEM : SA_AMD_EHIP); }

// device memory that is freed on every exit path (HIP_TRY returns early)
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    int alloc(size_t bytes) { reset(); return hip_status(hipMalloc(&p, bytes ? bytes : 1)); }
    void reset() { if (p) (void)hipFree(p); p = nullptr; }
    template <typename T> T *as() const { return (T *)p; }
};

// makes `device` current for the scope and restores the caller's device afterwards (device < 0: leave it alone)
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int rc = SA_AMD_OK;
    explicit DeviceGuard(int device)
    {
        if (device < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) { rc = SA_AMD_EHIP; return; }
        if (prev != device) { rc = hip_status(hipSetDevice(device)); switched = rc == SA_AMD_OK; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// the body of every extern "C" entry point runs inside this: nothing may unwind through the C ABI
#define SA_ABI_GUARD_BEGIN try {
#define SA_ABI_GUARD_END(fallback)                                                                 \
    } catch (const std::bad_alloc &) { return (fallback) == 0 ? SA_AMD_ENOMEM : (fallback); }      \
      catch (...) { return (fallback) == 0 ? SA_AMD_EINTERNAL : (fallback); }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int bit_length(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline) ----
enum KClass { KC_BYTE_HIST = 0, KC_BUILD_KEYS, KC_UPSWEEP, KC_SPINE, KC_DOWNSWEEP, KC_RR_COUNT, KC_RR_SCAN, KC_RR_APPLY,
              KC_GATHER, KC_SCATTER, KC_LOCAL, KC_MISC, KC_UPSWEEP32, KC_DOWNSWEEP32, KC_ONESWEEP, KC_ONESWEEP32, KC_FINISH, KC_BUCKET,
#ifdef SA_AMD_DIAG
              KC_SS_COUNT, KC_SS_SCATTER, KC_SS_BUCKET,
#endif
              KC_COUNT };
static const char *const kclass_names[KC_COUNT] = { "k_byte_hist", "k_build_keys", "k_radix_upsweep", "k_spine_rows",
                                                    "k_radix_downsweep", "k_rr_count", "k_rr_scan", "k_rr_apply",
                                                    "k_gather_key2", "k_scatter_pairs", "k_group_sort", "misc",   // (k_gather_key2: the plain gathers; k_group_sort: all fused gather + sort kernels)
                                                    "k_radix_upsweep32", "k_radix_downsweep32",
                                                    "k_onesweep", "k_onesweep32",      // single-pass tile scatter, 64- / 32-bit keys (kernels/onesweep.hpp)
                                                    "k_finish_sorted",                 // one pass over the sorted keys that orders the small groups in place
                                                    "k_bucket_sort"                    // the low 16 bits of the 32-bit first stage, bucket by bucket in LDS (kernels/bucket_sort.hpp)
#ifdef SA_AMD_DIAG
                                                    , "k_ss_count", "k_ss_scatter",    // diagnostic library: sample sort of the 64-bit stage (kernels/sample_sort.hpp), the two distribution
                                                    "k_ss_bucket_sort"                 // levels and every bucket ordered in LDS
#endif
                                                    };
struct Profiler {
    bool on = false;
    uint64_t mask = ~0ull;      // kernel classes that get events (each pair costs a few microseconds of host time)
    struct Rec { int cls; hipEvent_t a, b; int64_t units; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double ms[KC_COUNT] = { 0 };
    int64_t launches[KC_COUNT] = { 0 }, units[KC_COUNT] = { 0 };
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void begin(int cls, int64_t u, hipStream_t st)
    {
        open = on && ((mask >> cls) & 1ull);
        if (!open) return;
        Rec r; r.cls = cls; r.units = u; r.a = get(); r.b = get();
        (void)hipEventRecord(r.a, st);
        recs.push_back(r);
    }
    bool open = false;
    void end(hipStream_t st) { if (open && !recs.empty()) (void)hipEventRecord(recs.back().b, st); open = false; }
    ~Profiler()      // (a worker thread of sa_amd_saca_batch that profiled must not leak its events)
    {
        if ((long)syscall(SYS_gettid) == (long)getpid()) return;      // the main thread's copy dies at process exit: no HIP calls then
        for (auto &r : recs) { if (r.a) (void)hipEventDestroy(r.a); if (r.b) (void)hipEventDestroy(r.b); }
        for (auto e : pool) if (e) (void)hipEventDestroy(e);
    }
    void resolve()   // call after the stream has been synchronised
    {
        for (auto &r : recs) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.cls] += t; launches[r.cls]++; units[r.cls] += r.units; }
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
    }
};
static thread_local Profiler g_prof;
static thread_local sa_amd_stats g_last_stats;
#define PROF(cls, units, st, launch_stmt)                                                          \
    do { g_prof.begin(cls, units, st); launch_stmt; g_prof.end(st); LAUNCH_CHECK(st); } while (0)

}  // namespace sa
