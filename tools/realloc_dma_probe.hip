// Is a device -> pinned-host copy slower out of a block that was hipMalloc'ed AFTER another large block was freed?
// (seen in the engine: after the pool had dropped and re-allocated its device block, the staged download ran at 28 instead of
// 52 GB/s)   hipcc --offload-arch=gfx950 -O2 -o tools/bin/realloc_dma_probe tools/realloc_dma_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int measure(const char *what, char *d, size_t off, hipStream_t st, void **stage)
{
    const size_t chunk = (size_t)16 << 20, bytes = (size_t)256 << 20;
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        const double t0 = now_ms();
        for (size_t c = 0; c < bytes / chunk; ++c) CK(hipMemcpyAsync(stage[c % 3], d + off + c * chunk, chunk, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        const double dt = now_ms() - t0;
        if (dt < best) best = dt;
    }
    printf("%-64s %6.2f ms  %5.1f GB/s\n", what, best, 256.0 / 1024.0 * 1.073741824 / best * 1e3);
    return 0;
}
__global__ void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static int measure_kernel(const char *what, char *d, hipStream_t st, void **stage, int blocks)
{
    const size_t chunk = (size_t)16 << 20, bytes = (size_t)256 << 20;
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        const double t0 = now_ms();
        for (size_t c = 0; c < bytes / chunk; ++c)
            hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, st, (const uint4 *)(d + c * chunk), (uint4 *)stage[c % 3], chunk / 16);
        CK(hipStreamSynchronize(st));
        const double dt = now_ms() - t0;
        if (dt < best) best = dt;
    }
    printf("%-58s %4d WG %6.2f ms  %5.1f GB/s\n", what, blocks, best, 0.268435456 / best * 1e3);
    return 0;
}
int main(int argc, char **argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    void *stage[3];
    for (int i = 0; i < 3; ++i) CK(hipHostMalloc(&stage[i], (size_t)16 << 20, hipHostMallocPortable | hipHostMallocMapped));
    const size_t big = (size_t)3700 << 20, small = (size_t)64 << 20;
    char *a; CK(hipMalloc(&a, big)); CK(hipMemsetAsync(a, 1, big, st));
    if (measure("first block of the process (3.6 GiB)", a, 0, st, stage)) return 1;
    if (variant == 1) {            // a second block while the first is alive, no hipFree at all
        char *b; CK(hipMalloc(&b, big)); CK(hipMemsetAsync(b, 3, big, st));
        if (measure("v1: second block, the first still alive (no hipFree so far)", b, 0, st, stage)) return 1;
        if (measure("v1: the first block again", a, 0, st, stage)) return 1;
        return 0;
    }
    if (variant == 2) {            // free + allocate, nothing in between
        CK(hipStreamSynchronize(st)); CK(hipFree(a));
        char *b; CK(hipMalloc(&b, big)); CK(hipMemsetAsync(b, 3, big, st));
        if (measure("v2: block allocated right after the first was freed", b, 0, st, stage)) return 1;
        return 0;
    }
    if (variant == 3) {            // a small block allocated and freed: does any hipFree do it?
        char *s0; CK(hipMalloc(&s0, small)); CK(hipMemsetAsync(s0, 2, small, st)); CK(hipStreamSynchronize(st)); CK(hipFree(s0));
        if (measure("v3: the first block after a 64 MiB block was allocated and freed", a, 0, st, stage)) return 1;
        char *b; CK(hipMalloc(&b, big)); CK(hipMemsetAsync(b, 3, big, st));
        if (measure("v3: a second big block", b, 0, st, stage)) return 1;
        return 0;
    }
    if (variant == 5) {            // the same with a copy KERNEL into the mapped stage blocks
        for (int blocks : { 16, 64, 256, 1024 }) if (measure_kernel("v5: kernel copy, first block", a, st, stage, blocks)) return 1;
        CK(hipStreamSynchronize(st)); CK(hipFree(a));
        char *b; CK(hipMalloc(&b, big)); CK(hipMemsetAsync(b, 3, big, st));
        if (measure("v5: DMA, block allocated after the first was freed", b, 0, st, stage)) return 1;
        for (int blocks : { 16, 64, 256, 1024 }) if (measure_kernel("v5: kernel copy, block after free", b, st, stage, blocks)) return 1;
        return 0;
    }
    if (variant == 6) {            // in the slow state: do other ways of allocating device memory give blocks the engine reads at full rate?
        CK(hipStreamSynchronize(st)); CK(hipFree(a));
        char *b; CK(hipMalloc(&b, big)); CK(hipMemsetAsync(b, 3, big, st));
        if (measure("v6: hipMalloc after free", b, 0, st, stage)) return 1;
        {
            void *m = nullptr;
            if (hipMallocAsync(&m, big, st) == hipSuccess) {
                CK(hipMemsetAsync(m, 3, big, st));
                if (measure("v6: hipMallocAsync (stream-ordered pool)", (char *)m, 0, st, stage)) return 1;
                CK(hipFreeAsync(m, st)); CK(hipStreamSynchronize(st));
            } else { (void)hipGetLastError(); printf("v6: hipMallocAsync not available\n"); }
        }
        {
            void *m = nullptr;
            if (hipExtMallocWithFlags(&m, big, hipDeviceMallocUncached) == hipSuccess) {
                CK(hipMemsetAsync(m, 3, big, st));
                if (measure("v6: hipExtMallocWithFlags(uncached)", (char *)m, 0, st, stage)) return 1;
                CK(hipFree(m));
            } else { (void)hipGetLastError(); printf("v6: hipExtMallocWithFlags(uncached) failed\n"); }
        }
        {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0;
            if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran) {
                const size_t sz = (big + gran - 1) / gran * gran;
                hipMemGenericAllocationHandle_t h;
                void *va = nullptr;
                if (hipMemCreate(&h, sz, &prop, 0) == hipSuccess && hipMemAddressReserve(&va, sz, gran, nullptr, 0) == hipSuccess && hipMemMap(va, sz, 0, h, 0) == hipSuccess) {
                    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
                    CK(hipMemSetAccess(va, sz, &acc, 1));
                    CK(hipMemsetAsync(va, 3, big, st));
                    printf("v6: virtual memory API, granularity %zu\n", gran);
                    if (measure("v6: hipMemCreate + hipMemMap", (char *)va, 0, st, stage)) return 1;
                } else { (void)hipGetLastError(); printf("v6: virtual memory API failed\n"); }
            } else { (void)hipGetLastError(); printf("v6: no allocation granularity\n"); }
        }
        // and many smaller blocks instead of one large one
        {
            char *parts[16]; const size_t psz = (size_t)256 << 20;
            for (int i = 0; i < 16; ++i) { CK(hipMalloc(&parts[i], psz)); CK(hipMemsetAsync(parts[i], 1, psz, st)); }
            if (measure("v6: a 256 MiB hipMalloc block", parts[7], 0, st, stage)) return 1;
        }
        return 0;
    }
    if (variant == 7) {            // is the rate a property of the block or of the process?
        char *b; CK(hipMalloc(&b, big)); CK(hipMemsetAsync(b, 3, big, st));
        if (measure("v7: second block, the first alive", b, 0, st, stage)) return 1;
        CK(hipStreamSynchronize(st)); CK(hipFree(b));
        if (measure("v7: the FIRST block after the second was freed", a, 0, st, stage)) return 1;
        char *c; CK(hipMalloc(&c, big)); CK(hipMemsetAsync(c, 3, big, st));
        if (measure("v7: a third block", c, 0, st, stage)) return 1;
        if (measure("v7: the first block again", a, 0, st, stage)) return 1;
        CK(hipStreamSynchronize(st)); CK(hipFree(c));
        if (measure("v7: the first block after the third was freed", a, 0, st, stage)) return 1;
        char *d; CK(hipMalloc(&d, big)); CK(hipMemsetAsync(d, 3, big, st));
        if (measure("v7: a fourth block", d, 0, st, stage)) return 1;
        if (measure("v7: the first block again", a, 0, st, stage)) return 1;
        return 0;
    }
    if (variant == 4) {            // new pinned stage blocks after the free: is it the host side?
        CK(hipStreamSynchronize(st)); CK(hipFree(a));
        char *b; CK(hipMalloc(&b, big)); CK(hipMemsetAsync(b, 3, big, st));
        if (measure("v4: block after free, old stage blocks", b, 0, st, stage)) return 1;
        void *stage2[3];
        for (int i = 0; i < 3; ++i) CK(hipHostMalloc(&stage2[i], (size_t)16 << 20, hipHostMallocPortable | hipHostMallocMapped));
        if (measure("v4: block after free, NEW stage blocks", b, 0, st, stage2)) return 1;
        hipStream_t st2; CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        if (measure("v4: block after free, new stage blocks, new stream", b, 0, st2, stage2)) return 1;
        // host -> device for comparison
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            const double t0 = now_ms();
            for (size_t c = 0; c < 16; ++c) CK(hipMemcpyAsync(b + c * ((size_t)16 << 20), stage2[c % 3], (size_t)16 << 20, hipMemcpyHostToDevice, st2));
            CK(hipStreamSynchronize(st2));
            const double dt = now_ms() - t0; if (dt < best) best = dt;
        }
        printf("%-64s %6.2f ms  %5.1f GB/s\n", "v4: host -> device into that block", best, 0.268435456 / best * 1e3);
        return 0;
    }
    return 0;
}
