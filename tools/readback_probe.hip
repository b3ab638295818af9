// What does one blocking read-back of a few words cost?  (a) kernel -> hipMemcpyAsync into pinned memory -> hipStreamSynchronize
// (read_words of host/pipeline.hpp), (b) kernel -> a second tiny kernel that stores the words and then a sequence number into
// MAPPED pinned host memory -> the host spins on the sequence number.  hipcc --offload-arch=gfx950 -O2 -o tools/bin/readback_probe tools/readback_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_work(uint32_t *w, uint32_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) w[0] = v; }
__global__ void k_post(const uint32_t *src, volatile uint32_t *dst, int words, uint32_t seq)
{
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[1 + i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store((uint32_t *)dst, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t *d; CK(hipMalloc(&d, 4096));
    uint32_t *pin; CK(hipHostMalloc(&pin, 4096, hipHostMallocDefault));
    uint32_t *map; CK(hipHostMalloc(&map, 8192, hipHostMallocMapped | hipHostMallocCoherent));
    uint32_t *dmap; CK(hipHostGetDevicePointer((void **)&dmap, map, 0));
    map[0] = 0;
    const int N = 2000;
    for (int words : { 1, 64, 576 }) {
        // (a)
        for (int warm = 0; warm < 2; ++warm) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, (uint32_t)i);
                CK(hipMemcpyAsync(pin, d, words * 4, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                if (pin[0] != (uint32_t)i) { printf("mismatch a\n"); return 1; }
            }
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (warm) printf("%4d words  (a) kernel + memcpyAsync + streamSynchronize: %6.2f us per round trip\n", words, us);
        }
        // (b)
        uint32_t seq = map[0];
        for (int warm = 0; warm < 2; ++warm) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                ++seq;
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, (uint32_t)i);
                hipLaunchKernelGGL(k_post, dim3(1), dim3(256), 0, st, (const uint32_t *)d, (volatile uint32_t *)dmap, words, seq);
                while (__atomic_load_n((volatile uint32_t *)map, __ATOMIC_ACQUIRE) != seq) { }
                if (map[1] != (uint32_t)i) { printf("mismatch b\n"); return 1; }
            }
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (warm) printf("%4d words  (b) kernel + post kernel into mapped memory + host spin:  %6.2f us per round trip\n", words, us);
        }
        // (c) the kernel alone + sync (what the work itself costs)
        {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, (uint32_t)i); CK(hipStreamSynchronize(st)); }
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            printf("%4d words  (c) kernel + streamSynchronize only:                   %6.2f us\n", words, us);
        }
    }
    return 0;
}
