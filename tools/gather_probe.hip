// gather_probe.hip -- what does ONE random 4-byte look-up into a table far larger than the caches cost on MI355X, by the
// flavour of the load and of the allocation?  k_group_sort (the tie-breaking rounds behind sa_amd_saca_u8, the arithmetic
// that replaces cdivsufsort::sort_in_place, reference src/saca.rs:14) is bound by exactly this: rank[SA[i] + h] look-ups
// into a 1 GiB array.  rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum shows that a plain look-up is a 128-byte request.
//   flavour 0  plain global_load_dword
//           1  __builtin_nontemporal_load                       (nt)
//           2  relaxed atomic load, agent scope                 (sc1)
//           3  relaxed atomic load, system scope                (sc0 sc1)
//           4-7  global_load_dword with sc0 sc1 nt / sc0 / sc1 / sc0 sc1   (inline asm)
//   allocation 0 hipMalloc, 1 hipExtMallocWithFlags(hipDeviceMallocUncached), 2 hipDeviceMallocFinegrained
//   build:  hipcc -O3 --offload-arch=gfx950 -o tools/bin/gather_probe tools/gather_probe.hip
//   run:    tools/bin/gather_probe [log2 table words = 28] [log2 look-ups = 28] [sorted window: 0 = uniform]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ inline uint32_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}

template <int F>
__device__ __forceinline__ uint32_t look(const uint32_t *p)
{
    if constexpr (F == 0) return *p;
    else if constexpr (F == 1) return __builtin_nontemporal_load(p);
    else if constexpr (F == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (F == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else {
        uint32_t v;                                       // (issued only: the caller waits once for all eight)
        if constexpr (F == 4) asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "=&v"(v) : "v"(p) : "memory");
        else if constexpr (F == 5) asm volatile("global_load_dword %0, %1, off sc0" : "=&v"(v) : "v"(p) : "memory");
        else if constexpr (F == 6) asm volatile("global_load_dword %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
        else asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=&v"(v) : "v"(p) : "memory");
        return v;
    }
}

// every thread: 8 independent look-ups, all in flight at once
template <int F>
__global__ __launch_bounds__(256) void k_gather(const uint32_t *__restrict__ tab, uint32_t mask, uint32_t *__restrict__ out, uint64_t n)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t idx[8], v[8], acc = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) idx[r] = mix(t * 8 + r) & mask;
    if (t * 8 >= n) return;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = look<F>(tab + idx[r]);
    if constexpr (F >= 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
#pragma unroll
    for (int r = 0; r < 8; ++r) acc += v[r];
    out[t] = acc;
}

template <int F>
static void run(const char *name, const uint32_t *tab, uint32_t mask, uint32_t *out, uint64_t n)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)((n / 8 + 255) / 256);
    k_gather<F><<<grid, 256>>>(tab, mask, out, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 3; ++i) k_gather<F><<<grid, 256>>>(tab, mask, out, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 3;
    printf("  %-44s %8.3f ms  %6.1f G look-ups/s\n", name, ms, n / ms * 1e-6);
}

int main(int argc, char **argv)
{
    const int lt = argc > 1 ? atoi(argv[1]) : 28, ln = argc > 2 ? atoi(argv[2]) : 28;
    const uint64_t words = 1ull << lt, n = 1ull << ln;
    uint32_t *out;
    CK(hipMalloc(&out, n / 8 * 4));
    for (int alloc = 0; alloc < 3; ++alloc) {
        uint32_t *tab = nullptr;
        hipError_t e = alloc == 0 ? hipMalloc(&tab, words * 4)
                                  : hipExtMallocWithFlags((void **)&tab, words * 4, alloc == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (e != hipSuccess) { printf("allocation %d: %s\n", alloc, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipMemset(tab, 1, words * 4));
        CK(hipDeviceSynchronize());
        printf("table of 2^%d words (%s), 2^%d look-ups\n", lt, alloc == 0 ? "hipMalloc" : alloc == 1 ? "hipDeviceMallocUncached" : "hipDeviceMallocFinegrained", ln);
        run<0>("plain", tab, (uint32_t)(words - 1), out, n);
        run<1>("nontemporal", tab, (uint32_t)(words - 1), out, n);
        run<2>("atomic relaxed, agent scope", tab, (uint32_t)(words - 1), out, n);
        run<3>("atomic relaxed, system scope", tab, (uint32_t)(words - 1), out, n);
        run<4>("asm sc0 sc1 nt", tab, (uint32_t)(words - 1), out, n);
        run<5>("asm sc0", tab, (uint32_t)(words - 1), out, n);
        run<6>("asm sc1", tab, (uint32_t)(words - 1), out, n);
        run<7>("asm sc0 sc1", tab, (uint32_t)(words - 1), out, n);
        CK(hipFree(tab));
    }
    return 0;
}
