// host/host_path.hpp -- host pointers in, host pointers out: the contract of `saca()` (reference src/saca.rs:9-15),
// i.e. what `SuffixArray::new` / `::set` (src/sa.rs:23-33) hand over: a borrowed text and a caller-owned Vec<u32>.
//
// Device memory comes from a process-wide pool of blocks (one block = text + SA + workspace of one build), so neither
// repeated calls nor the short-lived worker threads of sa_amd_saca_batch pay hipMalloc / hipFree per call; the pool
// retains at most SA_AMD_CACHE_MAX_BYTES (default 128 GiB of the 288 GB) and sa_amd_release_cache() empties it.
// The suffix array travels back either with one hipMemcpy into the caller's pageable buffer, or -- large arrays --
// chunk by chunk through pinned staging buffers while helper threads move finished chunks into the caller's buffer
// (a freshly allocated Vec<u32>, as SuffixArray::new makes, is page-faulted in by several threads instead of one).
#pragma once
#include "pipeline.hpp"

#include <chrono>
#include <memory>
#include <mutex>
#include <sys/mman.h>

namespace sa {

constexpr int SA_MADV_POPULATE_WRITE = 23;      // MADV_POPULATE_WRITE (older libc headers do not define it; older kernels answer EINVAL)

static int pick_device()
{
    const char *e = getenv("SA_AMD_DEVICE");
    if (!e || !*e) return -1;              // -1: keep the calling thread's current device
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    return end == e ? -1 : (int)v;
}

static double wall_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// phases of the last host-pointer call of this thread, in milliseconds (sa_amd_last_host_timing)
struct HostTiming { double acquire = 0, h2d = 0, build = 0, d2h = 0, release = 0, total = 0; int staged = 0; double early = 0, spill = 0; };      // early: fraction of the array that travelled before the build was done
static thread_local HostTiming g_host_timing;

// ---- device -> caller's pageable buffer through pinned staging + helper threads ----
constexpr size_t STAGE_BYTES = (size_t)16 << 20;
constexpr int STAGE_COUNT = 3;

// Device -> pinned stage by the copy engine, or by a copy KERNEL that stores into the mapped block.  Measured
// (tools/realloc_dma_probe.hip, profiles/r04_realloc_dma_probe.txt): hipMemcpyAsync device -> host runs at 55 GB/s in a process
// until the process frees a large device block; from then on at 29.5 GB/s out of EVERY block, old or new, however allocated --
// and some processes start that way, depending on what the process before left behind; now and then it recovers.  A pool that
// trims idle blocks (pool.hpp), a caller whose texts grow: the download of the next build is then 9 instead of 5 ms per 256 MiB.
// 128 workgroups of plain 16-byte loads and stores reach 54 GB/s in either state (host -> device by the engine is 55 GB/s in both
// and stays) -- but they are workgroups: a batch whose downloads run beside the next text's build loses 8 % to them (8 x 512 MiB:
// 390-400 against 360 ms).  So every download MEASURES: one of its chunks goes through the engine between two events and is timed
// by the device's clock; below 40 GB/s the rest of the download (and the next one, except for its own timed chunk) is the
// kernel's, at 40 GB/s or more the engine's.  What the state was last is kept per device (g_engine_slow).
// SA_AMD_NO_KERNEL_D2H=1: always the engine; SA_AMD_KERNEL_D2H_ALWAYS=1: always the kernel (tests, A/B).
static std::atomic<int> g_engine_slow[64];         // per device: 1 = the copy engine was last seen at half rate
struct __attribute__((packed, aligned(4))) D2hWords4 { uint32_t x, y, z, w; };
__global__ __launch_bounds__(256) void k_copy_to_host(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t words)
{
    // (src is only 4-byte aligned -- the array without its sentinel starts one element in --, the stage is 16-byte aligned)
    const size_t n16 = words / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const D2hWords4 q = *(const D2hWords4 *)(src + 4 * i);
        *(uint4 *)(dst + 4 * i) = make_uint4(q.x, q.y, q.z, q.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (words & 3)) dst[n16 * 4 + threadIdx.x] = src[n16 * 4 + threadIdx.x];
}
static bool kernel_d2h() { return env_int("SA_AMD_NO_KERNEL_D2H", 0, 0, 1) == 0; }
static int copy_to_stage(void *stage, const void *dsrc, size_t bytes, hipStream_t st, int blocks)
{
    if (kernel_d2h() && (bytes & 3) == 0 && (((uintptr_t)dsrc) & 3) == 0 && bytes >= 4) {
        hipLaunchKernelGGL(k_copy_to_host, dim3((unsigned)blocks), dim3(256), 0, st, (const uint32_t *)dsrc, (uint32_t *)stage, bytes / 4);
        return hip_status(hipGetLastError());
    }
    return hip_status(hipMemcpyAsync(stage, dsrc, bytes, hipMemcpyDeviceToHost, st));
}
constexpr int D2H_BLOCKS = 128;             // workgroups of a download's copy kernel

// The calling thread keeps up to STAGE_COUNT device -> stage copies enqueued; when a chunk's event has fired, the chunk
// is cut into `copy_threads` page-aligned slices that the persistent helpers of the device's NUMA node (and the caller
// itself) move into the caller's buffer, then the stage takes the chunk STAGE_COUNT further on.  The DMA of the
// following chunks runs meanwhile.  No thread is created or joined per call (helpers.hpp).
static int staged_download(void *dst_host, const void *dsrc, size_t bytes, hipStream_t st, int copy_threads, int device, int node)
{
    PinBlock stage[STAGE_COUNT];
    hipEvent_t ev[STAGE_COUNT] = { nullptr, nullptr, nullptr };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SA_AMD_OK;
    for (int i = 0; i < STAGE_COUNT && rc == SA_AMD_OK; ++i) rc = pool().pinned(STAGE_BYTES, node, device, &stage[i]);
    for (int i = 0; i < STAGE_COUNT && rc == SA_AMD_OK; ++i) rc = hip_status(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    const size_t nchunk = (bytes + STAGE_BYTES - 1) / STAGE_BYTES;
    auto len = [&](size_t c) { return c + 1 < nchunk ? STAGE_BYTES : bytes - c * STAGE_BYTES; };
    const bool always = env_int("SA_AMD_KERNEL_D2H_ALWAYS", 0, 0, 1) != 0;
    std::atomic<int> &state = g_engine_slow[(unsigned)device & 63u];
    bool by_kernel = kernel_d2h() && (always || state.load(std::memory_order_relaxed) == 1);
    // the timed chunk: the second (a whole one; a process's very first copy into a stage block takes milliseconds of set-up)
    const bool timed = kernel_d2h() && !always && nchunk >= 3;
    const size_t timed_chunk = 1;
    if (timed && rc == SA_AMD_OK) rc = hip_status(hipEventCreate(&e0));
    if (timed && rc == SA_AMD_OK) rc = hip_status(hipEventCreate(&e1));
    auto issue = [&](size_t c) {
        const bool probe = timed && c == timed_chunk;
        int r = SA_AMD_OK;
        if (probe) r = hip_status(hipEventRecord(e0, st));
        if (r == SA_AMD_OK)
            r = (by_kernel && !probe) ? copy_to_stage(stage[c % STAGE_COUNT].p, (const char *)dsrc + c * STAGE_BYTES, len(c), st, D2H_BLOCKS)
                                      : hip_status(hipMemcpyAsync(stage[c % STAGE_COUNT].p, (const char *)dsrc + c * STAGE_BYTES, len(c), hipMemcpyDeviceToHost, st));
        if (probe && r == SA_AMD_OK) r = hip_status(hipEventRecord(e1, st));
        if (r == SA_AMD_OK) r = hip_status(hipEventRecord(ev[c % STAGE_COUNT], st));
        return r;
    };
    if (rc == SA_AMD_OK) {
        HelperPool &hp = helper_pool(node);
        const int T = copy_threads < 1 ? 1 : copy_threads;
        const bool trace = env_int("SA_AMD_VERBOSE", 0, 0, 9) >= 3;
        double t_wait = 0, t_copy = 0, t_mark = trace ? wall_ms() : 0;
        const bool started_by_kernel = by_kernel;
        for (size_t c = 0; c < nchunk && c < (size_t)STAGE_COUNT && rc == SA_AMD_OK; ++c) rc = issue(c);
        for (size_t c = 0; c < nchunk && rc == SA_AMD_OK; ++c) {
            rc = hip_status(hipEventSynchronize(ev[c % STAGE_COUNT]));
            if (trace) { const double now = wall_ms(); t_wait += now - t_mark; t_mark = now; }
            if (rc != SA_AMD_OK) break;
            if (timed && c == timed_chunk) {
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.0f) {
                    const double gbs = (double)STAGE_BYTES / (double)ms * 1e-6;
                    const int slow = gbs < 40.0 ? 1 : 0;
                    state.store(slow, std::memory_order_relaxed);
                    by_kernel = slow == 1;                        // (the chunks not yet issued follow the measurement)
                    if (trace) fprintf(stderr, "suffix_array_amd: the copy engine moved a chunk at %.1f GB/s -> the download goes on by %s\n", gbs, by_kernel ? "copy kernel" : "the engine");
                } else (void)hipGetLastError();
            }
            const size_t L = len(c), per = (((L + T - 1) / T) + 4095) & ~(size_t)4095;     // whole pages per slice, the last one takes the rest
            char *dst = (char *)dst_host + c * STAGE_BYTES;
            const char *src = (const char *)stage[c % STAGE_COUNT].p;
            hp.parallel_for(T, [=](int t) {
                const size_t b = (size_t)t * per, e = b + per < L ? b + per : L;
                if (b < e) memcpy(dst + b, src + b, e - b);
            });
            if (trace) { const double now = wall_ms(); t_copy += now - t_mark; t_mark = now; }
            if (c + STAGE_COUNT < nchunk) rc = issue(c + STAGE_COUNT);       // the stage is free again
        }
        if (trace) fprintf(stderr, "suffix_array_amd: staged download of %zu bytes (started by %s, ended by %s): %.2f ms waiting for the chunks, %.2f ms copying out of the stage (%d slices, %d helpers)\n",
                           bytes, started_by_kernel ? "copy kernel" : "the copy engine", by_kernel ? "copy kernel" : "the copy engine", t_wait, t_copy, T, hp.helpers());
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (int i = 0; i < STAGE_COUNT; ++i) {
        if (ev[i]) (void)hipEventDestroy(ev[i]);
        pool().release_pinned(stage[i]);
    }
    return rc;
}

// Callers that share a device (the worker threads of sa_amd_saca_batch, or an application that calls saca() from several
// threads) take turns PER PHASE: one upload, one build and one download at a time per device.  Left alone, threads that
// start together stay in step -- both upload, both build on half a GPU, both download on half a link (measured, two threads,
// 8 x 512 MiB: every call h2d 18 + build 32 + d2h 85 ms, 500 ms per batch) -- with the lanes one text's download runs at
// full link speed under the next one's upload and build.  SA_AMD_NO_LANES=1: no turns (A/B).  Texts below SA_AMD_LANES_MIN_N
// (32 MiB) never take turns: their builds leave most of the GPU idle and overlap (build_host).
struct DeviceLanes { std::mutex up, run, down; };
static DeviceLanes &device_lanes(int device)
{
    static DeviceLanes lanes[64];
    return lanes[(unsigned)device & 63u];
}
struct LaneTurn {
    std::mutex *m;
    LaneTurn(std::mutex &mx, bool on) : m(on ? &mx : nullptr) { if (m) m->lock(); }
    ~LaneTurn() { if (m) m->unlock(); }
    LaneTurn(const LaneTurn &) = delete;
    LaneTurn &operator=(const LaneTurn &) = delete;
};

// a pooled stream / pinned block that goes back to the pool on every path out of a scope (also when something throws:
// the entry points catch at the ABI boundary, sa_api.hip)
struct PooledStream {
    int device; hipStream_t st = nullptr;
    explicit PooledStream(int d) : device(d) {}
    ~PooledStream() { if (st) pool().release_stream(device, st); }
    PooledStream(const PooledStream &) = delete;
    PooledStream &operator=(const PooledStream &) = delete;
};
struct PooledPin {
    PinBlock b;
    ~PooledPin() { if (b.p) pool().release_pinned(b); }
    PooledPin() = default;
    PooledPin(const PooledPin &) = delete;
    PooledPin &operator=(const PooledPin &) = delete;
};

static int64_t lanes_min_n() { return env_int("SA_AMD_LANES_MIN_N", (int64_t)32 << 20, 0, (int64_t)1 << 40); }

// ---- early download: the front of the array travels while the last refinement rounds run (EarlyDownload, host/pipeline.hpp) ----
// From the moment the build says that only the slots of its tied list can still change until the build is done, a few helper
// threads pull the array chunk by chunk, front to back: device -> pinned staging block (one DMA per chunk on a copy stream of
// their own) -> the caller's buffer (memcpy by the helper that asked for the chunk; one thread moves ~28 GB/s, four keep up with
// the link).  (A plain hipMemcpy into the pageable buffer from one helper reaches the link rate too -- the runtime stages it
// itself -- but the build ran 0.5 ms slower beside it, same box; with the pinned blocks the rounds keep their pace: 8.51 against
// 8.65 ms for rounds 3-9 of C3 in the kernel traces with the copy off / on.)
// The entries that were still tied at that moment arrive with stale values: the build sends their final values behind
// (compacted, with the bitmap that says which entries they are), the helpers patch them in while the rest of the array is
// downloaded as before.  C3 (256 MiB): the copy starts ~9 ms before the build ends, 44 % of the array is there when it does.
constexpr int EARLY_PULLERS = 4;
struct EarlyPull {
    size_t chunk = STAGE_BYTES;              // bytes per copy (SA_AMD_EARLY_CHUNK_BYTES; a multiple of 4096, at most a staging block)
    std::atomic<size_t> issued{0};          // chunks handed to the pullers (started or about to start)
    std::atomic<bool> stop{false};
    std::atomic<int> rc{SA_AMD_OK};
    std::unique_ptr<std::atomic<uint8_t>[]> done;      // per chunk: it is in the caller's buffer
    char *dst = nullptr;
    const char *src = nullptr;
    size_t bytes = 0, nchunk = 0, max_chunks = 0;
    hipStream_t cst = nullptr;
    hipEvent_t after = nullptr;
    int device = -1, node = -1;
    void run(int)                           // on EARLY_PULLERS helper threads
    {
        if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); rc.store(SA_AMD_EHIP); return; }
        PinBlock stage;
        hipEvent_t ev = nullptr;
        int r = pool().pinned(STAGE_BYTES, node, device, &stage);
        if (r == SA_AMD_OK) r = hip_status(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        while (r == SA_AMD_OK) {
            const size_t c = issued.fetch_add(1);
            if (c >= max_chunks || stop.load()) break;      // (a chunk taken but not copied: the caller downloads it with the rest)
            const size_t b = c * chunk, len = b + chunk < bytes ? chunk : bytes - b;
            // (the copy engine here, whatever state it is in: copy kernels beside the last rounds of the build cost C3 more build time
            // than they saved -- build 60.9 against 50.7 ms on a stream like the build's, 55.6 with 32 workgroups on a stream of the
            // lowest priority, end to end 75.4 against 70.0 with the engine at 55 GB/s and 75.2 against 74.0 with it at 29.5)
            r = hip_status(hipMemcpyAsync(stage.p, src + b, len, hipMemcpyDeviceToHost, cst));
            if (r == SA_AMD_OK) r = hip_status(hipEventRecord(ev, cst));
            if (r == SA_AMD_OK) r = hip_status(hipEventSynchronize(ev));
            if (r != SA_AMD_OK) break;
            memcpy(dst + b, stage.p, len);
            done[c].store(1);
        }
        if (r != SA_AMD_OK) { (void)hipGetLastError(); rc.store(r); }
        if (ev) (void)hipEventDestroy(ev);
        pool().release_pinned(stage);
    }
    size_t copied() const                   // chunks of the front that have arrived (call after the pullers have been joined)
    {
        size_t c = 0;
        while (c < nchunk && c < max_chunks && done[c].load()) ++c;
        return c;
    }
};

// the final values of the entries that were still tied when the early copy began, into the part of the caller's array that has
// arrived: tile_off / bits / holes are host copies of EarlyDownload's device arrays; entries [0, upto)
static std::function<void(int)> early_patch_task(uint32_t *out, const uint32_t *tile_off, const uint32_t *bits, const uint32_t *holes, size_t upto, int threads)
{
    const size_t tiles = (upto + EARLY_TILE - 1) / EARLY_TILE;
    const size_t T = (size_t)(threads < 1 ? 1 : threads);
    const size_t per = (tiles + T - 1) / T;
    return [=](int t) {
        const size_t t0 = (size_t)t * per, t1 = t0 + per < tiles ? t0 + per : tiles;
        for (size_t tile = t0; tile < t1; ++tile) {
            size_t k = tile_off[tile];
            const size_t w0 = tile * (EARLY_TILE / 32), w1 = w0 + EARLY_TILE / 32;
            for (size_t w = w0; w < w1; ++w) {
                uint32_t b = bits[w];
                while (b) {
                    const size_t j = w * 32 + (size_t)__builtin_ctz(b);
                    if (j < upto) out[j] = holes[k];
                    ++k;
                    b &= b - 1u;
                }
            }
        }
    };
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
    const int node = device_numa_node(cur);                       // staging buffers and copy helpers live next to the GPU's PCIe root
    const double t_begin = wall_ms();
    const int small_max = (int)env_int("SA_AMD_SMALL_MAX", 8192, 0, SM_MAX_N);
    if (n <= small_max) {
        // Small texts: copy -> ONE launch -> copy.  The kernel reads the text from and writes the array to a pooled PINNED host
        // block over PCIe (zero-copy; hipHostMalloc memory is mapped into the device), so there is no device block, no
        // hipMemcpy and no read-back: 0.2 ms -> tens of microseconds for the sizes of the reference's own tests (src/tests.rs:14).
        const size_t tb = align_up((size_t)n, 256), need = tb + ((size_t)n + 1) * 4;
        PooledStream ps(cur);
        PooledPin pp;
        int rc = pool().stream(cur, &ps.st);
        if (rc != SA_AMD_OK) return rc;
        hipStream_t st = ps.st;
        rc = pool().pinned(need < ((size_t)64 << 10) ? ((size_t)64 << 10) : need, -1, cur, &pp.b);
        if (rc != SA_AMD_OK) return rc;
        PinBlock &pb = pp.b;
        void *dbase = nullptr;
        rc = hip_status(hipHostGetDevicePointer(&dbase, pb.p, 0));
        if (rc == SA_AMD_OK) {
            memcpy(pb.p, T, (size_t)n);
            if (n <= SM_LITE_SINGLE_N)
                hipLaunchKernelGGL((k_small_sa_lite), dim3(1), dim3(SM_LITE_THREADS), 0, st, (const uint8_t *)dbase, (uint32_t *)((char *)dbase + tb), (int)n,
                                   (uint32_t *)nullptr);
            else
                hipLaunchKernelGGL((k_small_sa), dim3(1), dim3(SM_THREADS), 0, st, (const uint8_t *)dbase, (uint32_t *)((char *)dbase + tb), (int)n,
                                   (uint32_t *)nullptr);
            rc = hip_status(hipGetLastError());
            const int rs = hip_status(hipStreamSynchronize(st));
            if (rc == SA_AMD_OK) rc = rs;
            if (rc == SA_AMD_OK) {
                const uint32_t *src = (const uint32_t *)((const char *)pb.p + tb);
                if (with_sentinel) memcpy(SA_host, src, ((size_t)n + 1) * 4); else memcpy(SA_host, src + 1, (size_t)n * 4);
            }
        }
        { sa_amd_stats z; memset(&z, 0, sizeof(z)); g_last_stats = z; }
        tm.total = tm.build = wall_ms() - t_begin;
        g_host_timing = tm;
        return rc;
    }
    const size_t wb = (size_t)sa_amd_workspace_bytes(n);
    const size_t tb = align_up((size_t)n, 256), sb = align_up(((size_t)n + 1) * 4, 256);
    const size_t need = tb + sb + wb;
    DevBlock blk;
    PinBlock spill;                                              // reduced-memory route: the workspace slabs the device block cannot hold
    struct SpillEnd { PinBlock &b_; ~SpillEnd() { if (b_.p) pool().release_pinned(b_); } } spill_end{ spill };
    size_t ws_dev = wb, ws_host = 0;                             // bytes of the workspace in the device block / in pinned host memory
    void *dW2 = nullptr;
    hipStream_t st = nullptr;
    int rc = pool().stream(cur, &st);
    if (rc != SA_AMD_OK) return rc;
    rc = pool().acquire(cur, need, &blk);
    if (rc == SA_AMD_ENOMEM && env_int("SA_AMD_NO_REDUCED", 0, 0, 1) == 0) {
        // The device cannot give text + array + the whole workspace (another tenant, or a text near MAX_LENGTH next to other
        // blocks).  The reference's engine needs 257 KiB beside its output, so "out of memory" is not an answer a drop-in should
        // give lightly: take what the device has, keep the text, the array and the most-used slabs there (carve() orders them by
        // need) and put the slabs that do not fit into pinned host memory, which the kernels reach over PCIe -- slow (the lists
        // and the third key buffer go first), correct.  SA_AMD_NO_REDUCED=1: SA_AMD_ENOMEM as before.
        (void)hipGetLastError();
        size_t free_b = 0, total_b = 0;
        const size_t margin = (size_t)256 << 20;
        const size_t floor_ws = carve(nullptr, n, ~(size_t)0, nullptr).bytes - (size_t)n * 32 - 64 * 8;      // everything but one value buffer, isa, the lists and the third key buffer
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > margin + tb + sb + floor_ws) {
            const size_t cap = ((free_b - margin) & ~(((size_t)2 << 20) - 1)) - tb - sb;
            const Workspace dry = carve(nullptr, n, cap, nullptr);
            rc = pool().acquire(cur, tb + sb + align_up(dry.bytes, 256), &blk);
            if (rc == SA_AMD_OK && dry.bytes2 > 0) {
                rc = pool().pinned(dry.bytes2, node, cur, &spill);
                if (rc == SA_AMD_OK) rc = hip_status(hipHostGetDevicePointer(&dW2, spill.p, 0));
                if (rc != SA_AMD_OK) { pool().release(blk); blk = DevBlock(); }
            }
            if (rc == SA_AMD_OK) { ws_dev = align_up(dry.bytes, 256); ws_host = dry.bytes2; }
        } else (void)hipGetLastError();
    }
    if (rc != SA_AMD_OK) { pool().release_stream(cur, st); return rc; }
    double t0 = wall_ms();
    tm.acquire = t0 - t_begin;
    tm.spill = (double)ws_host;
    uint8_t *dT = (uint8_t *)blk.p;
    uint32_t *dSA = (uint32_t *)((char *)blk.p + tb);
    void *dW = (char *)blk.p + tb + sb;
    DeviceLanes &lanes = device_lanes(cur);
    // (texts below SA_AMD_LANES_MIN_N bytes take no turns: a mid-size build is bound by launches and read-backs, several of them
    // in flight on one device overlap -- 128 x 1 MiB of English through one batch call 195 -> 66 ms with four host threads)
    const bool turns = env_int("SA_AMD_NO_LANES", 0, 0, 1) == 0 && n >= lanes_min_n();
    // The caller's array is usually FRESH memory (`vec![0; n + 1]`, reference src/sa.rs:24: zero pages that are mapped on first
    // write): the helpers touch its pages -- read a byte, write it back: the contents stay whatever they were -- while the GPU
    // builds, so the download later copies into mapped memory (C3, fresh buffer: d2h 26-31 -> 20 ms).  SA_AMD_NO_PREFAULT=1: off.
    const size_t out_bytes_all = ((size_t)n + (with_sentinel ? 1 : 0)) * 4;
    const int copy_threads = (int)env_int("SA_AMD_COPY_THREADS", 12, 0, 32);      // 0: plain hipMemcpy into the caller's buffer
    const size_t staged_min = (size_t)env_int("SA_AMD_STAGED_MIN_BYTES", (int64_t)64 << 20, 0, (int64_t)1 << 40);
    const bool staged = copy_threads > 0 && out_bytes_all >= staged_min;
    std::atomic<bool> prefault_stop(false);                     // (declared before the handle: the helpers read it until finish())
    std::atomic<size_t> prefault_next(0);
    HelperPool::Async prefault;
    HelperPool &hp = helper_pool(node);
    struct PrefaultEnd {                                         // (also when something below throws: the helpers hold a pointer to the handle)
        HelperPool &pool_; HelperPool::Async &h_; std::atomic<size_t> &next_; std::atomic<bool> &stop_;
        ~PrefaultEnd() { stop_.store(true); next_.store(~(size_t)0 >> 1); pool_.finish(h_); }      // (no unit is started any more)
    } prefault_end{ hp, prefault, prefault_next, prefault_stop };
    {
        LaneTurn turn(lanes.up, turns);
        rc = hip_status(hipMemcpyAsync(dT, T, (size_t)n, hipMemcpyHostToDevice, st));
        if (rc == SA_AMD_OK) rc = hip_status(hipStreamSynchronize(st));
    }
    double t1 = wall_ms();
    tm.h2d = t1 - t0;
    // (started behind the upload: the runtime's staging of the pageable text and the page touching would share the same cores)
    if (staged && env_int("SA_AMD_NO_PREFAULT", 0, 0, 1) == 0) {
        // In the order the download will fill the array, a staging chunk's worth (16 MiB) at a time, helper t taking the units
        // t, t + T, ...: when the build is done before the whole array is mapped (a 10 ms build of 512 MiB of random bytes against
        // 25-70 ms for the 2 GiB of its array) the helpers stop after the unit they are in and the download starts at once --
        // into the mapped front part at full speed, page-faulting the rest in as it goes, which costs less than waiting for it.
        // When the build ends first (random bytes: 4.6 ms of build against 20 ms of page mapping for a 1 GiB array), all but a few
        // helpers stop after the 2 MiB piece they are in -- the download needs them -- and the few go on mapping AHEAD of the
        // download, which would otherwise fault every page in from its copy threads (SA_AMD_PREFAULT_KEEP, default 3; 0: all stop).
        char *dst = (char *)SA_host;
        const size_t units = (out_bytes_all + STAGE_BYTES - 1) / STAGE_BYTES;
        const int keep = (int)env_int("SA_AMD_PREFAULT_KEEP", 3, 0, 32);
        std::atomic<bool> *stop = &prefault_stop;
        std::atomic<size_t> *next = &prefault_next;
        hp.start(prefault, copy_threads, [=](int t) {
            auto touch = [](uintptr_t from, uintptr_t to) {
                while (from < to) {
                    volatile char *q = (volatile char *)from;
                    const char c = *q;
                    *q = c;
                    from = (from + 4096) & ~(uintptr_t)4095;      // first byte of the next page
                }
            };
            // (2 MiB at a time inside a unit: the build waits for the piece a helper is in when it ends -- with whole 16 MiB
            // units a 0.8 ms build of a 16 MiB text took 1.5 ms into a fresh buffer)
            constexpr size_t PIECE = (size_t)2 << 20;
            auto over = [=]() { return t >= keep && stop->load(std::memory_order_relaxed); };
            for (;;) {
                if (over()) break;
                const size_t u = next->fetch_add(1, std::memory_order_relaxed);      // units in the order the download fills them
                if (u >= units) break;
                const size_t ub = u * STAGE_BYTES, ue = ub + STAGE_BYTES < out_bytes_all ? ub + STAGE_BYTES : out_bytes_all;
                for (size_t b = ub; b < ue && !over(); b += PIECE) {
                    const size_t e = b + PIECE < ue ? b + PIECE : ue;
                    const uintptr_t a = (uintptr_t)(dst + b), end = (uintptr_t)(dst + e);
                    // whole pages: one madvise call maps them writable without a trap per page (Linux >= 5.14; contents untouched);
                    // anything it refuses, and the partial pages at the ends, are touched byte by byte
                    const uintptr_t pa = (a + 4095) & ~(uintptr_t)4095, pe = end & ~(uintptr_t)4095;
                    if (pe > pa && madvise((void *)pa, pe - pa, SA_MADV_POPULATE_WRITE) == 0) {
                        touch(a, pa < end ? pa : end);
                        touch(pe > a ? pe : a, end);
                    } else touch(a, end);
                }
            }
        });
    }
    // early download (large arrays): see EarlyPull above.  SA_AMD_EARLY_DIV = d: the copy starts when at most n / d suffixes are
    // still tied (0: never); the chunks it takes before the build ends are bounded so that what has to be sent behind them fits
    // into the part of the caller's array that is still free.
    const int64_t early_div = env_int("SA_AMD_EARLY_DIV", 4, 0, 1 << 20);
    const size_t early_min = (size_t)env_int("SA_AMD_EARLY_MIN_BYTES", (int64_t)128 << 20, 0, (int64_t)1 << 40);
    const uint32_t *dsrc = with_sentinel ? dSA : dSA + 1;
    EarlyDownload early;
    EarlyPull pull;
    HelperPool::Async pull_job;
    hipStream_t cst = nullptr;
    pull.chunk = (size_t)env_int("SA_AMD_EARLY_CHUNK_BYTES", (int64_t)STAGE_BYTES, 65536, (int64_t)STAGE_BYTES) & ~(size_t)4095;
    const int64_t early_wait = env_int("SA_AMD_EARLY_WAIT_CHUNKS", 0, 0, 1 << 20);     // tests: the build waits until so many chunks have been copied
    // (the pullers are helper threads: a pool without helpers would run them only when the build is over)
    // (not on the reduced-memory route: the values sent behind would come out of a slab that may live in host memory)
    const bool early_on = staged && early_div > 0 && out_bytes_all >= early_min && out_bytes_all >= 4 * pull.chunk && hp.helpers() > 0 && ws_host == 0;
    if (early_on && rc == SA_AMD_OK && pool().stream(cur, &cst) == SA_AMD_OK) {
        early.off = with_sentinel ? 1 : 0;
        early.threshold = (int64_t)n / early_div;
        pull.dst = (char *)SA_host; pull.src = (const char *)dsrc; pull.bytes = out_bytes_all;
        pull.nchunk = (out_bytes_all + pull.chunk - 1) / pull.chunk;
        pull.done.reset(new std::atomic<uint8_t>[pull.nchunk]);
        for (size_t c = 0; c < pull.nchunk; ++c) pull.done[c].store(0);
        pull.cst = cst; pull.device = cur; pull.node = node;
        EarlyPull *pp = &pull;
        HelperPool *hpp = &hp;
        HelperPool::Async *pj = &pull_job;
        EarlyDownload *ep = &early;
        early.start = [pp, hpp, pj, ep](hipEvent_t ev) {
            // what is sent behind the early chunks -- tile offsets, bitmap, at most min(tied at the snapshot, entries taken) final
            // values -- lands in the end of the caller's array: the most chunks K with K chunks + all that <= the array
            size_t k = 0;
            for (;;) {
                const size_t b = (k + 1) * pp->chunk, e = b / 4;
                const size_t holes = (size_t)ep->m_snap < e ? (size_t)ep->m_snap : e;
                if (b + b / 32 + b / 2048 + holes * 4 + 16384 > pp->bytes) break;
                ++k;
            }
            pp->max_chunks = k;
            pp->after = ev;
            if (hipStreamWaitEvent(pp->cst, ev, 0) != hipSuccess) { (void)hipGetLastError(); return; }      // (no copy before the marks and SA[0] are in place)
            hpp->start(*pj, EARLY_PULLERS, [pp](int t) { pp->run(t); });
        };
        early.stop = [pp, early_wait]() -> int64_t {
            if (early_wait > 0) {
                const size_t want = (size_t)early_wait < pp->max_chunks ? (size_t)early_wait : pp->max_chunks;
                const double t_give_up = wall_ms() + 2000.0;
                for (;;) {
                    size_t have = 0;
                    while (have < want && pp->done[have].load()) ++have;
                    if (have >= want || pp->rc.load() != SA_AMD_OK || wall_ms() > t_give_up) break;
                    std::this_thread::yield();
                }
            }
            pp->stop.store(true);
            size_t k = pp->issued.load();
            if (k > pp->max_chunks) k = pp->max_chunks;
            const size_t b = k * pp->chunk;
            return (int64_t)((b < pp->bytes ? b : pp->bytes) / 4);
        };
    }
    struct PullEnd {                                             // (the puller holds pointers to this frame: joined on every path)
        HelperPool &pool_; HelperPool::Async &h_; EarlyPull &p_;
        ~PullEnd() { p_.stop.store(true); pool_.finish(h_); }
    } pull_end{ hp, pull_job, pull };
    if (rc == SA_AMD_OK) {
        LaneTurn turn(lanes.run, turns);
        rc = build_device(dT, dSA, n, dW, (int64_t)ws_dev, st, nullptr, early.start ? &early : nullptr, dW2, (int64_t)ws_host);
    }
    // (usually done by now; otherwise no further unit is started -- SA_AMD_PREFAULT_WAIT=1: every page is mapped first, A/B)
    if (env_int("SA_AMD_PREFAULT_WAIT", 0, 0, 1) == 0) prefault_stop.store(true, std::memory_order_relaxed);
    else hp.finish(prefault);                                    // (otherwise joined when the download is over: PrefaultEnd)
    pull.stop.store(true);
    t0 = wall_ms();
    tm.build = t0 - t1;
    HelperPool::Async patch_job;
    struct PatchEnd {                                            // (the patch tasks hold pointers into the caller's array)
        HelperPool &pool_; HelperPool::Async &h_;
        ~PatchEnd() { pool_.finish(h_); }
    } patch_end{ hp, patch_job };
    if (rc == SA_AMD_OK) {
        LaneTurn turn(lanes.down, turns);
        const uint32_t *src = dsrc;
        const size_t out_bytes = out_bytes_all;
        size_t have = 0;                                        // bytes of the array that are (or are about to be) in the caller's buffer
        size_t zone_at = out_bytes;                             // where the values sent behind have landed: free again once the patch is done
        if (early.started && early.covered > 0) {
            // the values sent behind: [tile offsets | bitmap | final values] land in the END of the caller's array (free until the
            // rest of the download gets there); the helpers patch the front part from them while the rest of the array travels
            const size_t covered = (size_t)early.covered, got = covered * 4 < out_bytes ? covered * 4 : out_bytes;
            const size_t tiles_c = (covered + EARLY_TILE - 1) / EARLY_TILE;
            const size_t off_b = align_up((tiles_c + 1) * 4, 256), bits_b = align_up(tiles_c * (EARLY_TILE / 8), 256), holes_b = (size_t)early.holes * 4;
            const size_t aux = off_b + bits_b + holes_b;
            if (got + aux + 8192 <= out_bytes) {
                char *zone = (char *)SA_host + ((out_bytes - aux) & ~(size_t)4095);
                rc = hip_status(hipMemcpyAsync(zone, early.d_tile_off, (tiles_c + 1) * 4, hipMemcpyDeviceToHost, st));
                if (rc == SA_AMD_OK) rc = hip_status(hipMemcpyAsync(zone + off_b, early.d_bits, tiles_c * (EARLY_TILE / 8), hipMemcpyDeviceToHost, st));
                if (rc == SA_AMD_OK && holes_b) rc = hip_status(hipMemcpyAsync(zone + off_b + bits_b, early.d_holes, holes_b, hipMemcpyDeviceToHost, st));
                if (rc == SA_AMD_OK) rc = hip_status(hipStreamSynchronize(st));
                hp.finish(pull_job);                             // (the chunk that was in flight when the build ended has arrived by now)
                if (rc == SA_AMD_OK && pull.rc.load() != SA_AMD_OK) rc = pull.rc.load();
                // chunks the puller took but did not copy (it saw the stop first) are downloaded with the rest
                const size_t copied_b = pull.copied() * pull.chunk < out_bytes ? pull.copied() * pull.chunk : out_bytes;
                if (rc == SA_AMD_OK && copied_b > 0) {
                    const int pt = copy_threads > 8 ? 8 : copy_threads;
                    hp.start(patch_job, pt, early_patch_task(SA_host, (const uint32_t *)zone, (const uint32_t *)(zone + off_b),
                                                             (const uint32_t *)(zone + off_b + bits_b), copied_b / 4, pt));
                    have = copied_b;
                    zone_at = (size_t)(zone - (char *)SA_host);
                    tm.early = (double)copied_b / (double)out_bytes;
                }
            }
            // (else: the early chunks are simply downloaded again with everything else)
        }
        hp.finish(pull_job);
        if (rc == SA_AMD_OK && pull.rc.load() != SA_AMD_OK) rc = pull.rc.load();
        auto fetch = [&](size_t from, size_t to) {
            if (rc != SA_AMD_OK || from >= to) return;
            if (staged) {
                tm.staged = copy_threads;
                rc = staged_download((char *)SA_host + from, (const char *)src + from, to - from, st, copy_threads, cur, node);
            } else {
                rc = hip_status(hipMemcpyAsync((char *)SA_host + from, (const char *)src + from, to - from, hipMemcpyDeviceToHost, st));
                if (rc == SA_AMD_OK) rc = hip_status(hipStreamSynchronize(st));      // (the turn ends when the copy has)
            }
        };
        const size_t split = zone_at < out_bytes ? (zone_at > have ? zone_at & ~(STAGE_BYTES - 1) : have) : out_bytes;
        fetch(have, split < have ? have : split);
        hp.finish(patch_job);                                    // the landing zone is free from here on
        fetch(split < have ? have : split, out_bytes);
    }
    hp.finish(pull_job);
    if (early.ev) { (void)hipEventDestroy(early.ev); early.ev = nullptr; }
    if (cst) { (void)hipStreamSynchronize(cst); pool().release_stream(cur, cst); }
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
        fprintf(stderr, "suffix_array_amd: n=%d device %d numa node %d acquire %.2f h2d %.2f build %.2f d2h %.2f (staged %d, %.0f %% of the array sent before the build was done) release %.2f total %.2f ms%s\n", n,
                cur, node, tm.acquire, tm.h2d, tm.build, tm.d2h, tm.staged, tm.early * 100.0, tm.release, tm.total,
                tm.spill > 0 ? " -- REDUCED-MEMORY route: part of the workspace in pinned host memory" : "");
    return rc;
}

// Many SMALL texts of one device (sa_amd_saca_batch): the texts of up to SA_AMD_SMALL_MAX bytes are packed into pooled pinned
// blocks (zero-copy, like the single small call above) and built by ONE launch per block of k_small_sa_batch, one workgroup
// per text -- 256 texts at a time on the chip, one synchronisation per chunk instead of one per text.  A chunk holds up to
// SMALL_BATCH_BYTES of texts + arrays + descriptors.  status[i] of the items handled is set; returns the first failure.
constexpr size_t SMALL_BATCH_BYTES = (size_t)96 << 20;
constexpr int SMALL_BATCH_TEXTS = 1 << 16;

static int build_host_small_batch(const uint8_t *const *T, uint32_t *const *SA, const int32_t *n, const int *items, size_t count,
                                  int device, int32_t *status)
{
    if (count == 0) return SA_AMD_OK;
    DeviceGuard guard(device);
    if (guard.rc != SA_AMD_OK) { for (size_t k = 0; k < count; ++k) status[items[k]] = guard.rc; return guard.rc; }
    int cur = 0;
    int rc = hip_status(hipGetDevice(&cur));
    PooledStream ps(cur);
    if (rc == SA_AMD_OK) { ps.device = cur; rc = pool().stream(cur, &ps.st); }
    if (rc != SA_AMD_OK) { for (size_t k = 0; k < count; ++k) status[items[k]] = rc; return rc; }
    hipStream_t st = ps.st;
    int first = SA_AMD_OK;
    HelperPool &hp = helper_pool(device_numa_node(cur));
    const int copy_slices = (int)env_int("SA_AMD_COPY_THREADS", 12, 0, 32) > 1 ? (int)env_int("SA_AMD_COPY_THREADS", 12, 0, 32) : 1;
    size_t k0 = 0;
    while (k0 < count) {
        // the chunk [k0, k1): descriptors first, then the texts (16-byte slots), then the arrays (4 (n + 1) bytes each, 16-byte slots)
        size_t k1 = k0, tbytes = 0, sbytes = 0;
        while (k1 < count && k1 - k0 < (size_t)SMALL_BATCH_TEXTS) {
            const size_t nn = (size_t)n[items[k1]];
            const size_t tb = align_up(nn, 16), sb = align_up((nn + 1) * 4, 16);
            if (k1 > k0 && (k1 - k0 + 1) * 16 + tbytes + tb + sbytes + sb > SMALL_BATCH_BYTES) break;
            tbytes += tb; sbytes += sb; ++k1;
        }
        const size_t cnt = k1 - k0, dbytes = align_up(cnt * 16, 256), need = dbytes + align_up(tbytes, 256) + sbytes;
        std::vector<uint32_t> slot_of(cnt);
        PooledPin pp;                                     // (back to the pool at the end of the chunk, also when a helper job throws)
        PinBlock &pb = pp.b;
        int rcc = pool().pinned(need < ((size_t)64 << 10) ? ((size_t)64 << 10) : need, -1, cur, &pb);
        void *dbase = nullptr;
        if (rcc == SA_AMD_OK) rcc = hip_status(hipHostGetDevicePointer(&dbase, pb.p, 0));
        if (rcc == SA_AMD_OK) {
            uint4 *desc = (uint4 *)pb.p;
            char *tpart = (char *)pb.p + dbytes, *spart = tpart + align_up(tbytes, 256);
            // descriptors: the texts of up to SM_LITE_N bytes first (the light kernel's: eight workgroups per CU), then the others
            size_t n_lite = 0;
            for (size_t k = k0; k < k1; ++k) n_lite += n[items[k]] <= SM_LITE_N ? 1 : 0;
            size_t to = 0, so = 0, li = 0, fi = n_lite;
            for (size_t k = k0; k < k1; ++k) {
                const size_t nn = (size_t)n[items[k]];
                const size_t d = nn <= (size_t)SM_LITE_N ? li++ : fi++;
                slot_of[k - k0] = (uint32_t)d;
                desc[d] = make_uint4((unsigned)to, (unsigned)so, (unsigned)nn, 0u);
                to += align_up(nn, 16); so += align_up((nn + 1) * 4, 16);
            }
            const uint32_t *slot = slot_of.data();
            // the copies in and out are split over the node's helpers when there is enough to copy (4 096 texts of 4 KiB: 16 MiB in,
            // 64 MiB out -- one thread's memcpy was three quarters of the call)
            const int slices = (tbytes + sbytes) >= ((size_t)4 << 20) ? copy_slices : 1;
            const size_t per = (cnt + (size_t)slices - 1) / (size_t)slices;
            hp.parallel_for(slices, [=](int t) {
                const size_t b = k0 + (size_t)t * per, e = b + per < k1 ? b + per : k1;
                for (size_t k = b; k < e; ++k) { const uint4 d = desc[slot[k - k0]]; memcpy(tpart + d.x, T[items[k]], (size_t)d.z); }
            });
            const uint8_t *d_texts = (const uint8_t *)dbase + dbytes;
            uint8_t *d_arrays = (uint8_t *)dbase + dbytes + align_up(tbytes, 256);
            if (n_lite > 0)
                hipLaunchKernelGGL((k_small_sa_batch<SM_LITE_N, SM_LITE_THREADS>), dim3((unsigned)n_lite), dim3(SM_LITE_THREADS), 0, st,
                                   d_texts, d_arrays, (const uint4 *)dbase);
            rcc = hip_status(hipGetLastError());
            if (cnt > n_lite && rcc == SA_AMD_OK) {
                hipLaunchKernelGGL((k_small_sa_batch<SM_MAX_N, SM_THREADS>), dim3((unsigned)(cnt - n_lite)), dim3(SM_THREADS), 0, st,
                                   d_texts, d_arrays, (const uint4 *)dbase + n_lite);
                rcc = hip_status(hipGetLastError());
            }
            const int rs = hip_status(hipStreamSynchronize(st));
            if (rcc == SA_AMD_OK) rcc = rs;
            if (rcc == SA_AMD_OK)
                hp.parallel_for(slices, [=](int t) {
                    const size_t b = k0 + (size_t)t * per, e = b + per < k1 ? b + per : k1;
                    for (size_t k = b; k < e; ++k) { const uint4 d = desc[slot[k - k0]]; memcpy(SA[items[k]], spart + d.y, ((size_t)d.z + 1) * 4); }
                });
        }
        for (size_t k = k0; k < k1; ++k) status[items[k]] = rcc;
        if (first == SA_AMD_OK) first = rcc;
        k0 = k1;
    }
    return first;
}

}  // namespace sa
