// pcie_probe.hip -- which way of moving a suffix array (4(n+1) bytes) from HBM into a caller's PAGEABLE buffer, and a
// text (n bytes) the other way, is fastest on this host?  Informs the staging design of the host-pointer entry points
// (sa_amd_saca_u8 / sa_amd_divsufsort; reference contract src/saca.rs:9-15: caller-owned slices in and out).
//   build:  hipcc -O2 --offload-arch=gfx950 -o /tmp/pcie_probe tools/pcie_probe.hip -lpthread
//   run:    /tmp/pcie_probe [MiB of SA = 1024]
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// parallel memcpy with T plain threads (created per call: what a library without a pool would pay)
static void par_memcpy(char *dst, const char *src, size_t bytes, int T)
{
    if (T <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes + T - 1) / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; ++t) {
        const size_t b = (size_t)t * per, e = b + per < bytes ? b + per : bytes;
        if (b >= e) break;
        th.emplace_back([=]() { memcpy(dst + b, src + b, e - b); });
    }
    for (auto &x : th) x.join();
}

// D2H through two pinned staging buffers: copy chunk c to stage[c & 1] on the stream, then T threads move it out
static double staged_d2h(char *dst, const char *dsrc, size_t bytes, size_t chunk, int T, char *stage[2], hipStream_t st)
{
    hipEvent_t ev[2];
    CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    const double t0 = now_ms();
    const size_t nchunk = (bytes + chunk - 1) / chunk;
    auto len = [&](size_t c) { return c + 1 < nchunk ? chunk : bytes - c * chunk; };
    CK(hipMemcpyAsync(stage[0], dsrc, len(0), hipMemcpyDeviceToHost, st));
    CK(hipEventRecord(ev[0], st));
    for (size_t c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) {
            CK(hipMemcpyAsync(stage[(c + 1) & 1], dsrc + (c + 1) * chunk, len(c + 1), hipMemcpyDeviceToHost, st));
            CK(hipEventRecord(ev[(c + 1) & 1], st));
        }
        CK(hipEventSynchronize(ev[c & 1]));
        par_memcpy(dst + c * chunk, stage[c & 1], len(c), T);
        // (stage[c & 1] is reused by chunk c + 2, issued in the next iteration: after this memcpy)
    }
    const double t1 = now_ms();
    CK(hipEventDestroy(ev[0])); CK(hipEventDestroy(ev[1]));
    return t1 - t0;
}

static double staged_h2d(char *ddst, const char *src, size_t bytes, size_t chunk, int T, char *stage[2], hipStream_t st)
{
    hipEvent_t ev[2];
    CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    const double t0 = now_ms();
    const size_t nchunk = (bytes + chunk - 1) / chunk;
    auto len = [&](size_t c) { return c + 1 < nchunk ? chunk : bytes - c * chunk; };
    for (size_t c = 0; c < nchunk; ++c) {
        if (c >= 2) CK(hipEventSynchronize(ev[c & 1]));          // the copy that last used this stage is done
        par_memcpy(stage[c & 1], src + c * chunk, len(c), T);
        CK(hipMemcpyAsync(ddst + c * chunk, stage[c & 1], len(c), hipMemcpyHostToDevice, st));
        CK(hipEventRecord(ev[c & 1], st));
    }
    CK(hipStreamSynchronize(st));
    const double t1 = now_ms();
    CK(hipEventDestroy(ev[0])); CK(hipEventDestroy(ev[1]));
    return t1 - t0;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 1024;
    const size_t bytes = mib << 20;
    printf("pcie_probe: %zu MiB, hardware threads %u\n", mib, std::thread::hardware_concurrency());
    char *d = nullptr;
    CK(hipMalloc((void **)&d, bytes));
    CK(hipMemset(d, 0x5a, bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    char *pageable = (char *)malloc(bytes);
    double t0 = now_ms();
    memset(pageable, 1, bytes);
    printf("first touch (memset) of %zu MiB pageable: %.1f ms\n", mib, now_ms() - t0);
    char *pinned = nullptr;
    t0 = now_ms();
    CK(hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault));
    printf("hipHostMalloc %zu MiB: %.1f ms\n", mib, now_ms() - t0);
    auto report = [&](const char *what, double ms) { printf("%-58s %8.2f ms  %7.2f GB/s\n", what, ms, bytes / ms / 1e6); fflush(stdout); };
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now_ms(); CK(hipMemcpy(pageable, d, bytes, hipMemcpyDeviceToHost)); report("D2H hipMemcpy -> pageable", now_ms() - t0);
        t0 = now_ms(); CK(hipMemcpy(pinned, d, bytes, hipMemcpyDeviceToHost)); report("D2H hipMemcpy -> pinned", now_ms() - t0);
        t0 = now_ms(); CK(hipMemcpy(d, pageable, bytes, hipMemcpyHostToDevice)); report("H2D hipMemcpy <- pageable", now_ms() - t0);
        t0 = now_ms(); CK(hipMemcpy(d, pinned, bytes, hipMemcpyHostToDevice)); report("H2D hipMemcpy <- pinned", now_ms() - t0);
    }
    {
        t0 = now_ms();
        CK(hipHostRegister(pageable, bytes, hipHostRegisterDefault));
        const double t1 = now_ms();
        CK(hipMemcpy(pageable, d, bytes, hipMemcpyDeviceToHost));
        const double t2 = now_ms();
        CK(hipHostUnregister(pageable));
        const double t3 = now_ms();
        printf("hipHostRegister %.1f ms + D2H %.1f ms (%.1f GB/s) + unregister %.1f ms = %.1f ms\n", t1 - t0, t2 - t1,
               bytes / (t2 - t1) / 1e6, t3 - t2, t3 - t0);
    }
    for (int T : { 1, 2, 4, 8, 16 }) {
        t0 = now_ms(); par_memcpy(pageable, pinned, bytes, T);
        char buf[96]; snprintf(buf, sizeof buf, "host memcpy pinned -> pageable, %d threads", T); report(buf, now_ms() - t0);
    }
    for (size_t cm : { (size_t)8, (size_t)32, (size_t)128 }) {
        const size_t chunk = cm << 20;
        char *stage[2];
        CK(hipHostMalloc((void **)&stage[0], chunk, hipHostMallocDefault));
        CK(hipHostMalloc((void **)&stage[1], chunk, hipHostMallocDefault));
        for (int T : { 1, 2, 4, 8 }) {
            char buf[96];
            double ms = staged_d2h(pageable, d, bytes, chunk, T, stage, st);
            snprintf(buf, sizeof buf, "D2H staged, chunk %zu MiB, %d copy threads", cm, T); report(buf, ms);
            ms = staged_h2d(d, pageable, bytes, chunk, T, stage, st);
            snprintf(buf, sizeof buf, "H2D staged, chunk %zu MiB, %d copy threads", cm, T); report(buf, ms);
        }
        CK(hipHostFree(stage[0])); CK(hipHostFree(stage[1]));
    }
    // allocation cost the per-call path pays today
    for (size_t gib : { (size_t)1, (size_t)4, (size_t)14 }) {
        void *p = nullptr;
        t0 = now_ms();
        if (hipMalloc(&p, gib << 30) != hipSuccess) { printf("hipMalloc %zu GiB failed\n", gib); continue; }
        const double t1 = now_ms();
        CK(hipFree(p));
        printf("hipMalloc %zu GiB: %.2f ms, hipFree: %.2f ms\n", gib, t1 - t0, now_ms() - t1);
    }
    return 0;
}
