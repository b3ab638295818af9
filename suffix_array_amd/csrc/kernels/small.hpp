// kernels/small.hpp -- the whole construction of a SMALL text in one launch of one workgroup.
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
//
// The reference's own test domain is n < 4096 (src/tests.rs:14) and a caller of `SuffixArray::new` (src/sa.rs:23-27) may
// well index many short strings.  The general pipeline needs ~15 launches and as many 4-byte read-backs whatever the size
// (0.2 ms for 16 bytes); a text of up to SM_MAX_N bytes fits one CU's LDS, so this kernel does everything there:
//   text -> LDS; key of suffix i = its first 7 symbols as 9-bit fields (byte + 1, 0 past the end: the shorter suffix is the
//   smaller, reference src/sa.rs:76-82); bitonic sort of (key, suffix) in LDS; ranks from the group heads; while some
//   suffixes are still tied: key = (rank[i], rank[i + h]) (0 past the end), sort again, h doubles (Manber-Myers in LDS,
//   at most log2(n) rounds -- random bytes need none);  SA[0] = n (reference src/saca.rs:13), SA[1 + p] = suffix at place p.
// There is no CPU path below a size threshold: this IS the small-n path, on the GPU.
#pragma once
#include "common.hpp"

namespace sa {

constexpr int SM_MAX_N = 8192;
constexpr int SM_THREADS = 1024;
constexpr int SM_PER = SM_MAX_N / SM_THREADS;      // places per thread in the rank scan

// G consecutive steps of a bitonic stage (stage size k; compare distances J, J / 2, ..., J >> (G - 1)) without a barrier in
// between: a thread takes the 2^G elements whose indices differ only in those G bits -- every partner of every step is
// its own -- so a sort of 8192 pairs takes 35 barriers instead of 91.
template <int G>
__device__ __forceinline__ void sm_bitonic_steps(unsigned long long *key, uint16_t *idx, int N2, int k, int J, int tid)
{
    constexpr int E = 1 << G;
    const int p0 = __ffs(J >> (G - 1)) - 1;               // lowest of the G bits
    for (int t = tid; t < (N2 >> G); t += SM_THREADS) {
        const int base = ((t >> p0) << (p0 + G)) | (t & ((1 << p0) - 1));
        const bool up = (base & k) == 0;
        unsigned long long kk[E];
        uint16_t ii[E];
#pragma unroll
        for (int m = 0; m < E; ++m) { kk[m] = key[base | (m << p0)]; ii[m] = idx[base | (m << p0)]; }
#pragma unroll
        for (int s = G - 1; s >= 0; --s) {
#pragma unroll
            for (int m = 0; m < E; ++m) {
                if (m & (1 << s)) continue;
                const int a = m, b = m | (1 << s);
                if ((kk[a] > kk[b]) == up && kk[a] != kk[b]) {
                    const unsigned long long tk = kk[a]; kk[a] = kk[b]; kk[b] = tk;
                    const uint16_t ti = ii[a]; ii[a] = ii[b]; ii[b] = ti;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < E; ++m) { key[base | (m << p0)] = kk[m]; idx[base | (m << p0)] = ii[m]; }
    }
}

__global__ __launch_bounds__(SM_THREADS) void k_small_sa(const uint8_t *__restrict__ T, uint32_t *__restrict__ SA, int n,
                                                         uint32_t *__restrict__ rounds_out)
{
    __shared__ unsigned long long key[SM_MAX_N];
    __shared__ uint16_t idx[SM_MAX_N];
    __shared__ uint16_t rnk[SM_MAX_N + 2];
    __shared__ uint8_t txt[SM_MAX_N + 8];
    __shared__ uint32_t scan_lds[SM_THREADS / WAVE + 1];
    __shared__ uint32_t s_groups;
    const int tid = threadIdx.x;
    int N2 = 2;
    while (N2 < n) N2 <<= 1;

    for (int i = tid; i < n + 8; i += SM_THREADS) txt[i] = i < n ? T[i] : (uint8_t)0;
    __syncthreads();
    for (int i = tid; i < N2; i += SM_THREADS) {
        unsigned long long k = ~0ull;                     // padding sorts behind every suffix
        if (i < n) {
            k = 0;
#pragma unroll
            for (int j = 0; j < 7; ++j) k = (k << 9) | (unsigned long long)(i + j < n ? (unsigned)txt[i + j] + 1u : 0u);
        }
        key[i] = k;
        idx[i] = (uint16_t)i;
    }
    __syncthreads();

    int h = 7;
    uint32_t rounds = 0;
    for (;;) {
        // ---- bitonic sort of (key, suffix), ascending; up to three compare-exchange steps per barrier (sm_bitonic_steps) ----
        for (int k = 2; k <= N2; k <<= 1) {
            int j = k >> 1;
            while (j > 0) {
                const int left = 32 - __clz(j);            // steps left in this stage: j, j / 2, ..., 1
                if (left >= 3) { sm_bitonic_steps<3>(key, idx, N2, k, j, tid); j >>= 3; }
                else if (left == 2) { sm_bitonic_steps<2>(key, idx, N2, k, j, tid); j >>= 2; }
                else { sm_bitonic_steps<1>(key, idx, N2, k, j, tid); j >>= 1; }
                __syncthreads();
            }
        }
        // ---- ranks: a group = run of equal keys, rank = place of its first member + 1 (0 = past the end of the text) ----
        uint32_t last_head = 0, heads = 0;                // (last_head: place + 1 of the latest group start at or before my places)
        uint32_t mine[SM_PER];
#pragma unroll
        for (int r = 0; r < SM_PER; ++r) {
            const int p = tid * SM_PER + r;
            const bool head = p < n && (p == 0 || key[p] != key[p - 1]);
            if (head) { last_head = (uint32_t)p + 1u; ++heads; }
            mine[r] = last_head;
        }
        uint32_t all_max, all_heads;
        const uint32_t before = block_incl_max<SM_THREADS>(last_head, scan_lds, &all_max);      // includes my own places
        (void)all_max;
        // the group start in front of my first place is the inclusive max of the threads before me: shift by one thread
        __shared__ uint32_t s_shift[SM_THREADS + 1];
        s_shift[tid + 1] = before;
        if (tid == 0) s_shift[0] = 0;
        __syncthreads();
        const uint32_t carry = s_shift[tid];
        (void)block_excl_sum<SM_THREADS>(heads, scan_lds, &all_heads);
        if (tid == 0) s_groups = all_heads;
#pragma unroll
        for (int r = 0; r < SM_PER; ++r) {
            const int p = tid * SM_PER + r;
            if (p < n) rnk[idx[p]] = (uint16_t)(mine[r] ? mine[r] : carry);
        }
        __syncthreads();
        if ((int)s_groups == n || h >= n) break;          // every suffix has a place of its own (h >= n cannot leave ties)
        // ---- next round: (rank[i], rank[i + h]) ----
        for (int i = tid; i < N2; i += SM_THREADS) {
            unsigned long long k = ~0ull;
            if (i < n) k = ((unsigned long long)rnk[i] << 16) | (unsigned long long)(i + h < n ? rnk[i + h] : (uint16_t)0);
            key[i] = k;
            idx[i] = (uint16_t)i;
        }
        __syncthreads();
        h *= 2;
        ++rounds;
    }
    for (int p = tid; p < n; p += SM_THREADS) SA[1 + p] = (uint32_t)idx[p];
    if (tid == 0) { SA[0] = (uint32_t)n; if (rounds_out) *rounds_out = rounds; }
}

}  // namespace sa
