// sa_extras.hpp -- the "next" rows of SURVEY.md section 8f that sit directly on either side of the
// construction path, as HIP kernels working on the device-resident text and suffix array:
//   k_bucket_table     reference src/sa.rs:89-119 (enable_buckets): right edge of every 1-/2-byte
//                      prefix bucket of the SA, 65 793 entries
//   k_ci_*             reference src/sa.rs:72-84 (check_integrity) in its linear-time form
#pragma once
#include "sa_kernels.hpp"

namespace sa {

constexpr int BKT_LEN = 256 * 257 + 1;     // reference src/sa.rs:95

// class of the suffix starting at p, in the layout of reference src/sa.rs:94:
// [$; (0,$), (0,0) ... (0,255); ...; (255,$), (255,0) ... (255,255)]
__device__ __forceinline__ uint32_t bucket_class(const uint8_t *__restrict__ T, int64_t n, int64_t p)
{
    if (p >= n) return 0u;                                            // the empty suffix, src/sa.rs:98
    const uint32_t c0 = T[p];
    if (p + 1 >= n) return c0 * 257u + 1u;                            // last byte alone, src/sa.rs:106-108
    return c0 * 257u + (uint32_t)T[p + 1] + 2u;                       // src/sa.rs:103
}

// The reference counts bigrams and prefix-sums them (src/sa.rs:100-116), so bkt[b] = number of
// suffixes whose class is <= b.  The SA is sorted and the class is monotone along it, so that
// number is an upper bound found by binary search: 65 793 searches of <= 32 probes each replace a
// 65 793-bin histogram over the whole text (which does not fit LDS as 32-bit counters).
__global__ __launch_bounds__(256) void k_bucket_table(const uint8_t *__restrict__ T, const uint32_t *__restrict__ SA,
                                                       int64_t n, uint32_t *__restrict__ bkt)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= BKT_LEN) return;
    int64_t lo = 0, hi = n + 1;                     // first slot whose class is > b
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (bucket_class(T, n, (int64_t)SA[mid]) <= (uint32_t)b) lo = mid + 1; else hi = mid;
    }
    bkt[b] = (uint32_t)lo;
}

// ---- check_integrity, linear-time form ----
// pass 1: range check + inverse permutation scatter; pass 2: the scatter must read back (catches
// duplicates) and every adjacent pair must be strictly increasing: T[a] < T[b], or equal first
// bytes and rank[a+1] < rank[b+1] (rank of the empty suffix is 0).
// flags: bit 0 = some entry out of range (the reference panics there), bit 1 = not a suffix array
__global__ __launch_bounds__(256) void k_ci_scatter(const uint32_t *__restrict__ SA, int64_t n, uint32_t *__restrict__ rank,
                                                     uint32_t *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= n; i += stride) {
        const uint32_t v = SA[i];
        if ((int64_t)v > n) atomicOr(flags, 1u);
        else rank[v] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void k_ci_check(const uint8_t *__restrict__ T, const uint32_t *__restrict__ SA, int64_t n,
                                                   const uint32_t *__restrict__ rank, uint32_t *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= n; i += stride) {
        const uint32_t b = SA[i];
        if ((int64_t)b > n) continue;                       // already flagged
        if (rank[b] != (uint32_t)i) { bad = true; continue; }   // duplicate value somewhere
        if (i == 0) { if ((int64_t)b != n) bad = true; continue; }   // the empty suffix must come first
        if (i == 1) continue;                               // anything non-empty is greater than the empty suffix
        const uint32_t a = SA[i - 1];
        if ((int64_t)a >= n || (int64_t)b >= n) { bad = true; continue; }   // n may only sit in slot 0
        const uint8_t ca = T[a], cb = T[b];
        if (ca < cb) continue;
        if (ca > cb || rank[a + 1] >= rank[b + 1]) bad = true;
    }
    if (__any(bad) && lane_id() == 0) atomicOr(flags, 2u);
}

}  // namespace sa
