// sa_api.hip -- the C ABI (include/suffix_array_amd.h) of the MI355X-native suffix-array construction engine.
// Replaces the body of `saca()` (reference src/saca.rs:9-15) and the C engine behind
// `cdivsufsort::sort_in_place` (src/saca.rs:14).
//
// There is deliberately no CPU fallback: every entry point runs the HIP kernels of kernels/*.hpp or returns an
// error code.  Nothing unwinds through the ABI (SA_ABI_GUARD_*), device buffers are RAII (DevBuf), and an entry point
// that switches the HIP device restores the caller's device before it returns (DeviceGuard).
//
// One translation unit, two products (csrc/Makefile):
//   libsuffix_array_amd.so        this file as is: bit-exact under ANY environment
//   libsuffix_array_amd_diag.so   -DSA_AMD_DIAG: adds the timing ablations (wrong orders on purpose), phase stamps and
//                                 the primitive test hooks of sa_diag.inc -- used by tools/ and the primitive tests only
#include "host/support.hpp"
#include "host/tuning.hpp"
#include "host/pipeline.hpp"
#include "host/host_path.hpp"

extern "C" {

#define SA_EXPORT __attribute__((visibility("default")))

SA_EXPORT int32_t sa_amd_max_length(void) { return SA_AMD_MAX_LENGTH; }

SA_EXPORT int32_t sa_amd_divsufsort(const uint8_t *T, int32_t *SA, int32_t n)
{
    SA_ABI_GUARD_BEGIN
    return sa::build_host(T, (uint32_t *)SA, n, false, sa::pick_device());
    SA_ABI_GUARD_END(0)
}

SA_EXPORT int32_t sa_amd_saca_u8(const uint8_t *T, uint32_t *SA, int32_t n)
{
    SA_ABI_GUARD_BEGIN
    return sa::build_host(T, SA, n, true, sa::pick_device());
    SA_ABI_GUARD_END(0)
}

SA_EXPORT int32_t sa_amd_saca_batch(const uint8_t *const *T, uint32_t *const *SA, const int32_t *n, const int32_t *device,
                                    int32_t count, int32_t *status)
{
    if (count < 0 || (count > 0 && (!T || !SA || !n))) return SA_AMD_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SA_AMD_ENODEVICE;
    SA_ABI_GUARD_BEGIN
    std::vector<int32_t> st((size_t)count, SA_AMD_OK);
    std::vector<std::vector<int>> per_dev((size_t)ndev), small_dev((size_t)ndev);
    // texts of up to SA_AMD_SMALL_MAX bytes (the one-workgroup kernel's) are built together, one launch per device and chunk
    // (host/host_path.hpp, build_host_small_batch) -- unless a device has a single one, which takes the single-call path
    const int small_max = (int)sa::env_int("SA_AMD_SMALL_MAX", 8192, 0, sa::SM_MAX_N);
    for (int i = 0; i < count; ++i) {
        const int d = device ? device[i] : i % ndev;
        if (d < 0 || d >= ndev) { st[(size_t)i] = SA_AMD_EINVAL; continue; }
        if (n[i] > 0 && n[i] <= small_max && T[i] && SA[i]) small_dev[(size_t)d].push_back(i);
        else per_dev[(size_t)d].push_back(i);
    }
    for (int d = 0; d < ndev; ++d)
        if (small_dev[(size_t)d].size() == 1) { per_dev[(size_t)d].push_back(small_dev[(size_t)d][0]); small_dev[(size_t)d].clear(); }
    std::vector<std::atomic<int>> small_taken((size_t)ndev);
    for (auto &a : small_taken) a.store(0);
    // Two host threads per device (SA_AMD_BATCH_THREADS, 1..16), each with its own stream and device block, take the
    // device's items in turn: while one waits for its 4(n+1)-byte copy back over PCIe the other uploads and computes,
    // so the link and the GPU overlap instead of alternating.
    // (SA_AMD_BATCH_THREADS = 0, the default: two per device; six for a device whose texts are all below SA_AMD_LANES_MIN_N, twelve
    // when they are all below 8 MiB -- those builds leave most of the GPU idle and take no turns, host/host_path.hpp; measured,
    // one GPU: 128 x 1 MiB of English 196 ms with two threads and turns, 67 with eight, 58 with twelve; 16 x 16 MiB 75 / 65 / 71)
    const int per_env = (int)sa::env_int("SA_AMD_BATCH_THREADS", 0, 0, 16);
    const int64_t lanes_min = sa::lanes_min_n();
    std::vector<int> per_of((size_t)ndev, 2);
    for (int d = 0; d < ndev; ++d) {
        int64_t largest = 0;
        for (int i : per_dev[(size_t)d]) largest = n[i] > largest ? n[i] : largest;
        per_of[(size_t)d] = per_env > 0 ? per_env : (largest < lanes_min ? (largest < ((int64_t)8 << 20) ? 12 : 6) : 2);
    }
    std::vector<std::atomic<size_t>> next((size_t)ndev);
    for (auto &a : next) a.store(0);
    auto work = [&](int d) {
        if (!small_dev[(size_t)d].empty() && small_taken[(size_t)d].exchange(1) == 0) {
            // (the first worker of the device to get here; the others go on with the large texts meanwhile)
            try { (void)sa::build_host_small_batch(T, SA, n, small_dev[(size_t)d].data(), small_dev[(size_t)d].size(), d, st.data()); }
            catch (const std::bad_alloc &) { for (int i : small_dev[(size_t)d]) st[(size_t)i] = SA_AMD_ENOMEM; }
            catch (...) { for (int i : small_dev[(size_t)d]) st[(size_t)i] = SA_AMD_EINTERNAL; }
        }
        for (;;) {
            const size_t q = next[(size_t)d].fetch_add(1);
            if (q >= per_dev[(size_t)d].size()) break;
            const int i = per_dev[(size_t)d][q];
            int32_t rc;
            try { rc = sa::build_host(T[i], SA[i], n[i], true, d); }
            catch (const std::bad_alloc &) { rc = SA_AMD_ENOMEM; }
            catch (...) { rc = SA_AMD_EINTERNAL; }
            st[(size_t)i] = rc;
        }
    };
    std::vector<std::thread> workers;
    bool spawn_failed = false;
    for (int d = 0; d < ndev && !spawn_failed; ++d) {
        const size_t items = per_dev[(size_t)d].size() + (small_dev[(size_t)d].empty() ? 0 : 1);
        for (int k = 0; k < per_of[(size_t)d] && (size_t)k < items; ++k) {
            try { workers.emplace_back(work, d); }
            catch (...) { spawn_failed = true; break; }         // (std::system_error: no more threads)
        }
    }
    for (auto &t : workers) t.join();
    if (spawn_failed)
        for (int d = 0; d < ndev; ++d) work(d);                  // whatever the started workers left is done here, serially
    int32_t first = SA_AMD_OK;
    for (int i = 0; i < count; ++i) {
        if (status) status[i] = st[(size_t)i];
        if (first == SA_AMD_OK && st[(size_t)i] != SA_AMD_OK) first = st[(size_t)i];
    }
    return first;
    SA_ABI_GUARD_END(0)
}

SA_EXPORT int64_t sa_amd_workspace_bytes(int32_t n)
{
    if (n < 0) return -1;
    return (int64_t)sa::carve(nullptr, n).bytes;
}

SA_EXPORT int32_t sa_amd_saca_device(const uint8_t *dT, uint32_t *dSA, int32_t n, void *dWork, int64_t work_bytes,
                                     void *stream, sa_amd_stats *stats)
{
    if (n < 0 || !dSA || (n > 0 && (!dT || !dWork))) return SA_AMD_EINVAL;
    if (n > 0 && (((uintptr_t)dWork) & 255u)) return SA_AMD_EINVAL;       // the carved slabs are read with 16-byte vector loads
    SA_ABI_GUARD_BEGIN
    return sa::build_device(dT, dSA, n, dWork, work_bytes, (hipStream_t)stream, stats);
    SA_ABI_GUARD_END(0)
}

// ---- next rows: bucket table and integrity check on the device-resident arrays ----

SA_EXPORT int32_t sa_amd_bucket_table_device(const uint8_t *dT, const uint32_t *dSA, int32_t n, uint32_t *dBkt, void *stream)
{
    SA_ABI_GUARD_BEGIN
    if (n < 0 || !dBkt || (n > 0 && !dT)) return SA_AMD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dSA) {
        // a device-resident index has the sorted order at hand: one binary search per bucket edge (~0.08 ms whatever n)
        hipLaunchKernelGGL(sa::k_bucket_table, dim3((sa::BKT_LEN + 255) / 256), dim3(256), 0, st, dT, dSA, (int64_t)n, dBkt);
    } else {
        // as the reference builds it (src/sa.rs:96-116): bigram counts of the text + prefix sum, no suffix array needed.  The
        // counts live in the first 65 536 words of dBkt itself (k_bigram_scan reads them all before it writes)
        if (hipMemsetAsync(dBkt, 0, (size_t)65536 * 4, st) != hipSuccess) return SA_AMD_EHIP;
        int64_t pairs = sa::ceil_div((int64_t)n, sa::BG_MIN_CHUNK);
        if (pairs > sa::BG_MAX_PAIRS) pairs = sa::BG_MAX_PAIRS;
        if (n >= 2)
            hipLaunchKernelGGL(sa::k_bigram_hist, dim3((unsigned)(2 * pairs)), dim3(sa::BG_THREADS), 0, st, dT, (int64_t)n, (int)pairs, dBkt);
        hipLaunchKernelGGL(sa::k_bigram_scan, dim3(1), dim3(sa::BGS_THREADS), 0, st, (const uint32_t *)dBkt, dT, (int64_t)n, dBkt);
    }
    if (hipGetLastError() != hipSuccess) return SA_AMD_EHIP;
    return hipStreamSynchronize(st) == hipSuccess ? SA_AMD_OK : SA_AMD_EHIP;
    SA_ABI_GUARD_END(0)
}

// layout of the larger work block (fast form): flags | rank | four pair buffers | spine + digit totals | granules + error word
struct CiLayout { size_t flags, rank, alt, alt_elems, spine, status, err, starts, bitmap, bitmap_bytes, bytes; };
static CiLayout ci_layout(int32_t n)
{
    CiLayout L;
    const size_t N1 = (size_t)n + 1;
    size_t off = 0;
    auto take = [&](size_t b) { const size_t o = off; off = sa::align_up(off + b, 256); return o; };
    L.flags = take(256);
    L.rank = take(N1 * 4);
    L.alt_elems = (N1 + 67) & ~(size_t)3;
    L.alt = take(4 * L.alt_elems * 4);
    L.spine = take(((size_t)sa::RADIX * sa::SORT_MAX_WG + sa::RADIX) * 4);
    L.status = take(((size_t)sa::ceil_div((int64_t)N1, sa::OS_MIN_TILE) + 1) * sa::RADIX * 8);
    L.err = take(256);
    L.starts = take(257 * 4);
    L.bitmap_bytes = ((N1 + 31) / 32 + 1) * 4;
    L.bitmap = take(L.bitmap_bytes);
    L.bytes = off;
    return L;
}

SA_EXPORT int64_t sa_amd_check_integrity_work_bytes(int32_t n)
{
    if (n < 0) return -1;
    return (int64_t)ci_layout(n).bytes;
}

SA_EXPORT int32_t sa_amd_check_integrity_device(const uint8_t *dT, int32_t n, const uint32_t *dSA, void *dWork,
                                                int64_t work_bytes, void *stream)
{
    if (n < 0 || !dSA || !dWork || (n > 0 && !dT)) return SA_AMD_EINVAL;
    if (work_bytes < ((int64_t)n + 1) * 4 + 256) return SA_AMD_EINVAL;
    SA_ABI_GUARD_BEGIN
    using namespace sa;
    hipStream_t st = (hipStream_t)stream;
    uint32_t *flags = (uint32_t *)dWork;
    uint32_t *rank = (uint32_t *)((char *)dWork + 256);
    HIP_TRY(hipMemsetAsync(flags, 0, 4, st));
    int64_t blocks = ((int64_t)n + 1 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const CiLayout L = ci_layout(n);
    if (work_bytes >= (int64_t)L.bytes && (((uintptr_t)dWork) & 255u) == 0 && (((uintptr_t)dSA) & 15u) == 0 && n >= 2) {
        // ---- streaming form: range check, binned inverse permutation, one random rank line per slot ----
        hipLaunchKernelGGL(k_ci_range, dim3((unsigned)blocks), dim3(256), 0, st, dSA, (int64_t)n, flags);
        HIP_TRY(hipGetLastError());
        uint32_t f = 0;
        { const int rcw = read_words(&f, flags, 4, st); if (rcw) return rcw; }
        if (f & 1u) return SA_AMD_ERANGE;
        if (f & 2u) return 0;
        char *base = (char *)dWork;
        Workspace w;
        memset(&w, 0, sizeof(w));
        w.isa = (uint32_t *)(base + L.rank);
        w.spine = (uint32_t *)(base + L.spine);
        w.digit_tot = w.spine + (size_t)RADIX * SORT_MAX_WG;
        w.os_status = (unsigned long long *)(base + L.status);
        w.os_err = (uint32_t *)(base + L.err);
        w.ss.spine = w.spine; w.ss.digit_tot = w.digit_tot; w.ss.status = w.os_status; w.ss.err = w.os_err;
        HIP_TRY(hipMemsetAsync(w.os_err, 0, 16, st));
        uint32_t *alt = (uint32_t *)(base + L.alt);
        const Tuning tn = Tuning::from_env(N_SORT_VARIANTS, N_SORT32_VARIANTS, N_OS_SHAPES64, N_OS_SHAPES32);
        sa_amd_stats local;
        memset(&local, 0, sizeof(local));
        // pairs (SA[i], i), i = 0 .. n, binned by the suffix position; the scatter skips the empty suffix (value n)
        const int rcs = scatter_binned((uint32_t *)dSA, nullptr, alt, alt + L.alt_elems, (int64_t)n + 1, (int64_t)n, w, st, &local, tn, true,
                                       alt + 2 * L.alt_elems, alt + 3 * L.alt_elems);
        if (rcs) return rcs;
        // first bytes: boundaries proposed from the array, proved in text order (streaming); then the slot-order check
        uint32_t *starts = (uint32_t *)(base + L.starts), *bitmap = (uint32_t *)(base + L.bitmap);
        HIP_TRY(hipMemsetAsync(bitmap, 0, L.bitmap_bytes, st));
        hipLaunchKernelGGL(k_ci_starts, dim3(1), dim3(512), 0, st, dT, dSA, (int64_t)n, starts, bitmap);
        int64_t fblocks = ceil_div((int64_t)n, 256 * 16);
        if (fblocks > 16384) fblocks = 16384;
        hipLaunchKernelGGL(k_ci_first_bytes, dim3((unsigned)fblocks), dim3(256), 0, st, dT, (int64_t)n, (const uint32_t *)w.isa, (const uint32_t *)starts, flags);
        const int64_t cblocks = ceil_div((int64_t)n, (int64_t)CI_THREADS * CI_ITEMS);
        hipLaunchKernelGGL(k_ci_check_shared, dim3((unsigned)cblocks), dim3(CI_THREADS), 0, st, dSA, (int64_t)n, (const uint32_t *)w.isa,
                           (const uint32_t *)bitmap, flags);
        HIP_TRY(hipGetLastError());
        uint32_t words[2] = { 0, 0 };
        { const int rcw = read_words(&words[0], flags, 4, st); if (rcw) return rcw; }
        { const int rcw = read_words(&words[1], w.os_err, 4, st); if (rcw) return rcw; }
        if (words[1]) return SA_AMD_EINTERNAL;
        return (words[0] & 2u) ? 0 : 1;
    }
    // ---- small work block (4 (n + 1) + 256 bytes): random-store inverse, three rank reads per slot ----
    hipLaunchKernelGGL(sa::k_ci_scatter, dim3((unsigned)blocks), dim3(256), 0, st, dSA, (int64_t)n, rank, flags);
    hipLaunchKernelGGL(sa::k_ci_check, dim3((unsigned)blocks), dim3(256), 0, st, dT, dSA, (int64_t)n, (const uint32_t *)rank, flags);
    if (hipGetLastError() != hipSuccess) return SA_AMD_EHIP;
    uint32_t f = 0;
    if (hipMemcpyAsync(&f, flags, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return SA_AMD_EHIP;
    if (hipStreamSynchronize(st) != hipSuccess) return SA_AMD_EHIP;
    if (f & 1u) return SA_AMD_ERANGE;
    return (f & 2u) ? 0 : 1;
    SA_ABI_GUARD_END(0)
}

// enable_buckets on host buffers (reference src/sa.rs:89-119): the text goes up, 257 KiB come back; nothing else is needed --
// the reference builds the table from the text alone.  Device block and stream come from the process-wide pool.
static int32_t bucket_table_host(const uint8_t *T, int32_t n, uint32_t *bkt)
{
    using namespace sa;
    if (n < 0 || !bkt || (n > 0 && !T)) return SA_AMD_EINVAL;
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    DeviceGuard guard(pick_device());
    if (guard.rc != SA_AMD_OK) return guard.rc;
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    const size_t tb = align_up((size_t)n + 16, 256);
    DevBlock blk;
    hipStream_t st = nullptr;
    int32_t rc = pool().stream(cur, &st);
    if (rc != SA_AMD_OK) return rc;
    rc = pool().acquire(cur, tb + (size_t)BKT_LEN * 4, &blk);
    if (rc != SA_AMD_OK) { pool().release_stream(cur, st); return rc; }
    uint8_t *dT = (uint8_t *)blk.p;
    uint32_t *dB = (uint32_t *)((char *)blk.p + tb);
    if (n > 0) rc = hip_status(hipMemcpyAsync(dT, T, (size_t)n, hipMemcpyHostToDevice, st));
    if (rc == SA_AMD_OK) rc = sa_amd_bucket_table_device(dT, nullptr, n, dB, st);
    if (rc == SA_AMD_OK) rc = hip_status(hipMemcpyAsync(bkt, dB, (size_t)BKT_LEN * 4, hipMemcpyDeviceToHost, st));
    const int32_t rs = hip_status(hipStreamSynchronize(st));       // (also drains the stream after a failure)
    if (rc == SA_AMD_OK) rc = rs;
    pool().release(blk);
    pool().release_stream(cur, st);
    return rc;
}

// host buffers; which = 2: integrity check, 3: build SA (into SA, n + 1 entries) then bucket table
static int32_t extras_host(const uint8_t *T, int32_t n, uint32_t *SA, int64_t sa_len, uint32_t *bkt, int which)
{
    using namespace sa;
    if (n < 0 || !SA || (n > 0 && !T)) return SA_AMD_EINVAL;
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    if (which == 2 && sa_len != (int64_t)n + 1) return 0;          // reference src/sa.rs:73-75: false
    DeviceGuard guard(pick_device());
    if (guard.rc != SA_AMD_OK) return guard.rc;
    DevBuf dT, dSA, dB, dW;
    int32_t rc;
    const size_t N = (size_t)n;
    if ((rc = dT.alloc(N))) return rc;
    if ((rc = dSA.alloc((N + 1) * 4))) return rc;
    if (N) HIP_TRY(hipMemcpy(dT.p, T, N, hipMemcpyHostToDevice));
    if (which == 3) {
        const int64_t wb = sa_amd_workspace_bytes(n);
        if ((rc = dW.alloc((size_t)wb))) return rc;
        if ((rc = build_device(dT.as<uint8_t>(), dSA.as<uint32_t>(), n, dW.p, wb, nullptr, nullptr))) return rc;
        HIP_TRY(hipMemcpy(SA, dSA.p, (N + 1) * 4, hipMemcpyDeviceToHost));
        // the text is in HBM already: the table from its bigrams, as the reference counts them
        if ((rc = dB.alloc((size_t)BKT_LEN * 4))) return rc;
        if ((rc = sa_amd_bucket_table_device(dT.as<uint8_t>(), nullptr, n, dB.as<uint32_t>(), nullptr))) return rc;
        HIP_TRY(hipMemcpy(bkt, dB.p, (size_t)BKT_LEN * 4, hipMemcpyDeviceToHost));
        return SA_AMD_OK;
    }
    HIP_TRY(hipMemcpy(dSA.p, SA, (N + 1) * 4, hipMemcpyHostToDevice));
    int64_t wb = sa_amd_check_integrity_work_bytes(n);              // the streaming form; the small block if that much is not to be had
    rc = dW.alloc((size_t)wb);
    if (rc == SA_AMD_ENOMEM) { (void)hipGetLastError(); wb = ((int64_t)n + 1) * 4 + 256; rc = dW.alloc((size_t)wb); }
    if (rc) return rc;
    return sa_amd_check_integrity_device(dT.as<uint8_t>(), n, dSA.as<uint32_t>(), dW.p, wb, nullptr);
}

SA_EXPORT int32_t sa_amd_bucket_table(const uint8_t *T, int32_t n, const uint32_t *SA, uint32_t *bkt)
{
    if (!bkt) return SA_AMD_EINVAL;
    SA_ABI_GUARD_BEGIN
    (void)SA;                       // (the reference builds the table from the text alone, src/sa.rs:96-116)
    return bucket_table_host(T, n, bkt);
    SA_ABI_GUARD_END(0)
}

SA_EXPORT int32_t sa_amd_saca_u8_buckets(const uint8_t *T, uint32_t *SA, int32_t n, uint32_t *bkt)
{
    if (!bkt) return SA_AMD_EINVAL;
    SA_ABI_GUARD_BEGIN
    return extras_host(T, n, SA, (int64_t)n + 1, bkt, 3);
    SA_ABI_GUARD_END(0)
}

SA_EXPORT int32_t sa_amd_check_integrity(const uint8_t *T, int32_t n, const uint32_t *SA, int64_t sa_len)
{
    SA_ABI_GUARD_BEGIN
    return extras_host(T, n, (uint32_t *)SA, sa_len, nullptr, 2);
    SA_ABI_GUARD_END(0)
}

// ---- device-resident index: text + suffix array kept in HBM for bucket table, integrity check and batched search ----

struct sa_amd_index {
    int device;
    int32_t n;
    uint8_t *dT;
    uint32_t *dSA;
    uint32_t *dBkt;       // bucket table once sa_amd_index_buckets has built it (narrows the searches, src/sa.rs:123-161)
};

SA_EXPORT int32_t sa_amd_index_create(const uint8_t *T, int32_t n, const uint32_t *SA, sa_amd_index **out)
{
    using namespace sa;
    if (!out || n < 0 || (n > 0 && !T)) return SA_AMD_EINVAL;
    *out = nullptr;
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    SA_ABI_GUARD_BEGIN
    DeviceGuard guard(pick_device());
    if (guard.rc != SA_AMD_OK) return guard.rc;
    DevBuf dT, dSA;
    int32_t rc;
    const size_t N = (size_t)n;
    if ((rc = dT.alloc(N))) return rc;
    if ((rc = dSA.alloc((N + 1) * 4))) return rc;
    if (N) HIP_TRY(hipMemcpy(dT.p, T, N, hipMemcpyHostToDevice));
    if (SA) HIP_TRY(hipMemcpy(dSA.p, SA, (N + 1) * 4, hipMemcpyHostToDevice));
    else {                                                       // SuffixArray::new on the device
        DevBuf dW;
        const int64_t wb = sa_amd_workspace_bytes(n);
        if ((rc = dW.alloc((size_t)wb))) return rc;
        if ((rc = build_device(dT.as<uint8_t>(), dSA.as<uint32_t>(), n, dW.p, wb, nullptr, nullptr))) return rc;
    }
    sa_amd_index *ix = new (std::nothrow) sa_amd_index();
    if (!ix) return SA_AMD_ENOMEM;
    ix->n = n; ix->device = 0; ix->dBkt = nullptr;
    (void)hipGetDevice(&ix->device);
    ix->dT = dT.as<uint8_t>(); ix->dSA = dSA.as<uint32_t>();
    dT.p = nullptr; dSA.p = nullptr;                             // ownership moves to the index
    *out = ix;
    return SA_AMD_OK;
    SA_ABI_GUARD_END(0)
}

SA_EXPORT void sa_amd_index_destroy(sa_amd_index *ix)      // (frees and deletes: nothing that throws)
{
    if (!ix) return;
    if (ix->dT) (void)hipFree(ix->dT);
    if (ix->dSA) (void)hipFree(ix->dSA);
    if (ix->dBkt) (void)hipFree(ix->dBkt);
    delete ix;
}

SA_EXPORT int32_t sa_amd_index_sa(const sa_amd_index *ix, uint32_t *SA_out)
{
    if (!ix || !SA_out) return SA_AMD_EINVAL;
    sa::DeviceGuard guard(ix->device);
    if (guard.rc != SA_AMD_OK) return guard.rc;
    return hipMemcpy(SA_out, ix->dSA, ((size_t)ix->n + 1) * 4, hipMemcpyDeviceToHost) == hipSuccess ? SA_AMD_OK : SA_AMD_EHIP;
}

SA_EXPORT int32_t sa_amd_index_buckets(sa_amd_index *ix, uint32_t *bkt)
{
    SA_ABI_GUARD_BEGIN
    if (!ix || !bkt) return SA_AMD_EINVAL;
    sa::DeviceGuard guard(ix->device);
    if (guard.rc != SA_AMD_OK) return guard.rc;
    if (!ix->dBkt) {
        uint32_t *dB = nullptr;
        if (hipMalloc((void **)&dB, (size_t)sa::BKT_LEN * 4) != hipSuccess) { (void)hipGetLastError(); return SA_AMD_ENOMEM; }
        const int32_t rc = sa_amd_bucket_table_device(ix->dT, ix->dSA, ix->n, dB, nullptr);
        if (rc != SA_AMD_OK) { (void)hipFree(dB); return rc; }
        ix->dBkt = dB;                                           // kept: later searches start from the pattern's bucket
    }
    return hipMemcpy(bkt, ix->dBkt, (size_t)sa::BKT_LEN * 4, hipMemcpyDeviceToHost) == hipSuccess ? SA_AMD_OK : SA_AMD_EHIP;
    SA_ABI_GUARD_END(0)
}

SA_EXPORT int32_t sa_amd_index_check_integrity(const sa_amd_index *ix)
{
    SA_ABI_GUARD_BEGIN
    if (!ix) return SA_AMD_EINVAL;
    sa::DeviceGuard guard(ix->device);
    if (guard.rc != SA_AMD_OK) return guard.rc;
    // the work block of the streaming form comes from the process-wide pool (a 5.5 GB hipMalloc / hipFree per call would
    // cost more than the check); the small block if that much is not to be had
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return SA_AMD_EHIP;
    sa::DevBlock blk;
    int64_t wb = sa_amd_check_integrity_work_bytes(ix->n);
    int32_t rc = sa::pool().acquire(cur, (size_t)wb, &blk);
    if (rc == SA_AMD_ENOMEM) { wb = ((int64_t)ix->n + 1) * 4 + 256; rc = sa::pool().acquire(cur, (size_t)wb, &blk); }
    if (rc) return rc;
    rc = sa_amd_check_integrity_device(ix->dT, ix->n, ix->dSA, blk.p, wb, nullptr);
    sa::pool().release(blk);
    return rc;
    SA_ABI_GUARD_END(0)
}

SA_EXPORT int32_t sa_amd_index_search(const sa_amd_index *ix, const uint8_t *pat_data, const int64_t *pat_off, int32_t count,
                                      uint8_t *contains, uint32_t *range_lo, uint32_t *range_hi, uint32_t *lcp_start,
                                      uint32_t *lcp_len)
{
    SA_ABI_GUARD_BEGIN
    using namespace sa;
    if (!ix || count < 0 || (count > 0 && !pat_off)) return SA_AMD_EINVAL;
    if (count == 0) return SA_AMD_OK;
    const int64_t total = pat_off[count];
    if (total < 0 || (total > 0 && !pat_data)) return SA_AMD_EINVAL;
    for (int32_t i = 0; i < count; ++i) if (pat_off[i + 1] < pat_off[i] || pat_off[i] < 0) return SA_AMD_EINVAL;
    DeviceGuard guard(ix->device);
    if (guard.rc != SA_AMD_OK) return guard.rc;
    const size_t C = (size_t)count;
    DevBuf dP, dO, dC, dR;
    int32_t rc;
    if ((rc = dP.alloc((size_t)total))) return rc;
    if ((rc = dO.alloc((C + 1) * 8))) return rc;
    if ((rc = dC.alloc(C))) return rc;
    if ((rc = dR.alloc(C * 4 * 4))) return rc;
    if (total) HIP_TRY(hipMemcpy(dP.p, pat_data, (size_t)total, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dO.p, pat_off, (C + 1) * 8, hipMemcpyHostToDevice));
    uint32_t *R = dR.as<uint32_t>();
    const int64_t threads = (int64_t)count * WAVE;
    hipLaunchKernelGGL(k_search_batch, dim3((unsigned)ceil_div(threads, SEARCH_THREADS)), dim3(SEARCH_THREADS), 0, nullptr,
                       (const uint8_t *)ix->dT, (const uint32_t *)ix->dSA, (int64_t)ix->n, dP.as<const uint8_t>(),
                       dO.as<const int64_t>(), count, dC.as<uint8_t>(), R, R + C, R + 2 * C, R + 3 * C, (const uint32_t *)ix->dBkt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (contains) HIP_TRY(hipMemcpy(contains, dC.p, C, hipMemcpyDeviceToHost));
    if (range_lo) HIP_TRY(hipMemcpy(range_lo, R, C * 4, hipMemcpyDeviceToHost));
    if (range_hi) HIP_TRY(hipMemcpy(range_hi, R + C, C * 4, hipMemcpyDeviceToHost));
    if (lcp_start) HIP_TRY(hipMemcpy(lcp_start, R + 2 * C, C * 4, hipMemcpyDeviceToHost));
    if (lcp_len) HIP_TRY(hipMemcpy(lcp_len, R + 3 * C, C * 4, hipMemcpyDeviceToHost));
    return SA_AMD_OK;
    SA_ABI_GUARD_END(0)
}

// ---- packed format (reference src/packed_sa.rs); byte layout: u32 magic "SA4x" LE, u32 length, u64 data length
//      (bincode's Vec<u8> prefix), data ----

static int sa_bits_of(uint32_t length)          // reference src/packed_sa.rs:127-129
{
    const uint32_t v = length ? length - 1 : 0;
    return v ? sa::bit_length(v) : 0;
}

SA_EXPORT int64_t sa_amd_pack_bound(int64_t length)
{
    if (length < 0 || length > 0xffffffffLL) return -1;
    const int bits = sa_bits_of((uint32_t)length);
    return 16 + (int64_t)((length + 127) / 128) * bits * 16;
}

SA_EXPORT int32_t sa_amd_pack(const uint32_t *SA, int64_t length, uint8_t *out, int64_t capacity, int64_t *out_len)
{
    SA_ABI_GUARD_BEGIN
    using namespace sa;
    if (!SA || !out || !out_len || length < 1 || length > 0xffffffffLL) return SA_AMD_EINVAL;
    if (capacity < sa_amd_pack_bound(length)) return SA_AMD_EINVAL;
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    const int bits = sa_bits_of((uint32_t)length);
    const int64_t blocks = (length + 127) / 128;
    const int64_t words = blocks * bits * 4;
    int64_t data_len = 0;
    if (bits > 0) {
        DeviceGuard guard(pick_device());
        if (guard.rc != SA_AMD_OK) return guard.rc;
        DevBuf dS, dO;
        int32_t rc;
        if ((rc = dS.alloc((size_t)length * 4))) return rc;
        if ((rc = dO.alloc((size_t)words * 4))) return rc;
        HIP_TRY(hipMemcpy(dS.p, SA, (size_t)length * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_pack4x, dim3((unsigned)ceil_div(words, 256)), dim3(256), 0, nullptr, dS.as<const uint32_t>(), length, bits,
                           dO.as<uint32_t>(), words);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out + 16, dO.p, (size_t)words * 4, hipMemcpyDeviceToHost));
        data_len = words * 4;
        if (length % 128) {                                   // a partial last block loses its trailing zero bytes (src/packed_sa.rs:41-45)
            const int64_t last = (blocks - 1) * bits * 16;
            while (data_len > last && out[16 + data_len - 1] == 0) --data_len;
        }
    }
    const uint32_t magic = 2016690515u, len32 = (uint32_t)length;   // src/packed_sa.rs:7
    const uint64_t dl = (uint64_t)data_len;
    memcpy(out, &magic, 4); memcpy(out + 4, &len32, 4); memcpy(out + 8, &dl, 8);
    *out_len = 16 + data_len;
    return SA_AMD_OK;
    SA_ABI_GUARD_END(0)
}

SA_EXPORT int32_t sa_amd_unpack(const uint8_t *bytes, int64_t nbytes, uint32_t *SA, int64_t capacity, int64_t *length)
{
    SA_ABI_GUARD_BEGIN
    using namespace sa;
    if (!bytes || !length || nbytes < 16) return SA_AMD_EINVAL;
    uint32_t magic, len32; uint64_t dl;
    memcpy(&magic, bytes, 4); memcpy(&len32, bytes + 4, 4); memcpy(&dl, bytes + 8, 8);
    if (magic != 2016690515u || dl != (uint64_t)(nbytes - 16)) return SA_AMD_EINVAL;       // InvalidData in the reference
    *length = len32;
    if (!SA || capacity < (int64_t)len32) return SA_AMD_EINVAL;
    const int bits = sa_bits_of(len32);
    const int64_t blocks = ((int64_t)len32 + 127) / 128;
    const int64_t full = blocks * bits * 16;
    // every block but the last is stored whole, and only a PARTIAL last block is right-trimmed (src/packed_sa.rs:36-46):
    // anything shorter is a truncated file, not an array with missing zeros
    const int64_t min_dl = (len32 % 128) ? (blocks - 1) * bits * 16 : full;
    if ((int64_t)dl > full || (int64_t)dl < min_dl) return SA_AMD_EINVAL;
    if (len32 == 0) return SA_AMD_OK;
    if (bits == 0) { SA[0] = 0; return SA_AMD_OK; }           // length 1: the reference's unpack loop does not terminate here
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    DeviceGuard guard(pick_device());
    if (guard.rc != SA_AMD_OK) return guard.rc;
    const int64_t in_words = ((int64_t)dl + 3) / 4;
    DevBuf dI, dS;
    int32_t rc;
    if ((rc = dI.alloc((size_t)(in_words ? in_words : 1) * 4))) return rc;
    if ((rc = dS.alloc((size_t)len32 * 4))) return rc;
    HIP_TRY(hipMemset(dI.p, 0, (size_t)(in_words ? in_words : 1) * 4));
    if (dl) HIP_TRY(hipMemcpy(dI.p, bytes + 16, (size_t)dl, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_unpack4x, dim3((unsigned)ceil_div((int64_t)len32, 256)), dim3(256), 0, nullptr, dI.as<const uint32_t>(), in_words,
                       (int64_t)len32, bits, dS.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(SA, dS.p, (size_t)len32 * 4, hipMemcpyDeviceToHost));
    return SA_AMD_OK;
    SA_ABI_GUARD_END(0)
}

SA_EXPORT void sa_amd_release_cache(void)
{
    try { sa::pool().clear(); } catch (...) { }
}

SA_EXPORT void sa_amd_last_stats(sa_amd_stats *out)
{
    if (out) *out = sa::g_last_stats;
}

SA_EXPORT int32_t sa_amd_last_host_timing(double *ms, int32_t capacity)
{
    const sa::HostTiming &t = sa::g_host_timing;
    const double v[9] = { t.acquire, t.h2d, t.build, t.d2h, t.release, t.total, (double)t.staged, t.early, t.spill };
    for (int i = 0; i < capacity && i < 9; ++i) ms[i] = v[i];
    return 9;
}

SA_EXPORT int32_t sa_amd_device_count(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return 0;
    return ndev;
}

SA_EXPORT int32_t sa_amd_device_pci_bus_id(int32_t device, char *buf, int32_t capacity)
{
    if (!buf || capacity < 16) return SA_AMD_EINVAL;
    buf[0] = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SA_AMD_ENODEVICE;
    if (device < 0 || device >= ndev) return SA_AMD_EINVAL;
    if (hipDeviceGetPCIBusId(buf, capacity, device) != hipSuccess) { (void)hipGetLastError(); buf[0] = 0; return SA_AMD_EHIP; }
    return SA_AMD_OK;
}

SA_EXPORT const char *sa_amd_strerror(int32_t code)
{
    switch (code) {
    case SA_AMD_OK: return "ok";
    case SA_AMD_EINVAL: return "invalid argument";
    case SA_AMD_ENOMEM: return "out of memory";
    case SA_AMD_EHIP: return "HIP runtime error";
    case SA_AMD_ENODEVICE: return "no HIP device";
    case SA_AMD_EINTERNAL: return "internal error: refinement did not converge";
    case SA_AMD_ERANGE: return "suffix offset out of range";
    default: return "unknown error";
    }
}

SA_EXPORT void sa_amd_profile_begin_classes(uint64_t class_mask)
{
    sa::Profiler &p = sa::g_prof;
    p.on = true;
    p.mask = class_mask;
    for (int i = 0; i < sa::KC_COUNT; ++i) { p.ms[i] = 0; p.launches[i] = 0; p.units[i] = 0; }
}

SA_EXPORT void sa_amd_profile_begin(void) { sa_amd_profile_begin_classes(~0ull); }

SA_EXPORT int32_t sa_amd_profile_end(double *ms, int64_t *launches, int64_t *units, int32_t capacity)
{
    sa::Profiler &p = sa::g_prof;
    p.on = false;
    const int cnt = capacity < sa::KC_COUNT ? capacity : sa::KC_COUNT;
    for (int i = 0; i < cnt; ++i) {
        if (ms) ms[i] = p.ms[i];
        if (launches) launches[i] = p.launches[i];
        if (units) units[i] = p.units[i];
    }
    return sa::KC_COUNT;
}

SA_EXPORT const char *sa_amd_profile_kernel_name(int32_t i)
{
    return (i >= 0 && i < sa::KC_COUNT) ? sa::kclass_names[i] : "";
}

#ifdef SA_AMD_DIAG
SA_EXPORT const char *sa_amd_version(void) { return "suffix_array_amd 0.2.0 (gfx950) DIAGNOSTIC BUILD"; }
#include "sa_diag.inc"
#else
SA_EXPORT const char *sa_amd_version(void) { return "suffix_array_amd 0.2.0 (gfx950)"; }
#endif

}  // extern "C"
