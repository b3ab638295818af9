// kernels/small.hpp -- the whole construction of a SMALL text in one launch of one workgroup.
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
//
// The reference's own test domain is n < 4096 (src/tests.rs:14) and a caller of `SuffixArray::new` (src/sa.rs:23-27) may
// well index many short strings.  The general pipeline needs ~15 launches and as many 4-byte read-backs whatever the size
// (0.2 ms for 16 bytes); a text of up to SM_MAX_N bytes fits one CU's LDS, so this kernel does everything there:
//   text -> LDS; key of suffix i = its first 6 symbols as a base-257 number (byte + 1, 0 past the end: the shorter suffix is the
//   smaller, reference src/sa.rs:76-82) with i in the low 13 bits; bitonic sort of the 8-byte elements in LDS; ranks from the group heads; while some
//   suffixes are still tied: key = (rank[i], rank[i + h], rank[i + 2h] [, rank[i + 3h]]) (0 past the end), sort again, h
//   triples / quadruples (Manber-Myers in LDS with wider keys, at most log3(n) rounds -- random bytes need none);  SA[0] = n (reference src/saca.rs:13), SA[1 + p] = suffix at place p.
// There is no CPU path below a size threshold: this IS the small-n path, on the GPU.
#pragma once
#include "common.hpp"

namespace sa {

constexpr int SM_MAX_N = 8192;
constexpr int SM_THREADS = 1024;
constexpr int SM_LITE_N = 1024;                    // the light shape: texts of up to this many bytes, SM_LITE_THREADS threads, 15 KB of LDS
constexpr int SM_LITE_THREADS = 256;
constexpr int SM_LITE_SINGLE_N = 512;             // a single call takes the light shape up to here (measured: 16 B 21.0 -> 19.5 us, 256 B 25.7 -> 24.3, 1 KiB 30.7 -> 33.8)
constexpr int SM_PER = SM_MAX_N / SM_THREADS;      // places per thread in the rank scan
constexpr int SM_LIGHT = 48;                       // largest tied group that is ordered by counting instead of a full sort
// a suffix number rides in the low bits of its key: 12 bits up to 4096 suffixes, else 13; a rank (0 ..= n) takes one bit more

// Where element e of the sorted array lives in LDS.  The bitonic steps below read 8-byte elements at strides of 8 and 64
// elements across the lanes of a wave (a lane owns the elements that differ in bits 0-2, or 3-5); in a linear layout those
// are 8- and 4-way bank conflicts in most steps.  Folding bits 5-7 into bits 0-4 of the index makes both patterns -- and
// the unit-stride one -- hit 32 different 8-byte bank pairs per half wave (a bijection inside every aligned block of 256).
__device__ __forceinline__ int sm_phys(int e) { return e ^ ((e >> 5) & 7) ^ (((e >> 6) & 3) << 3); }

// G consecutive steps of a bitonic stage (stage size k; compare distances J, J / 2, ..., J >> (G - 1)) without a barrier in
// between: a thread takes the 2^G elements whose indices differ only in those G bits -- every partner of every step is
// its own -- so a sort of 8192 elements takes 35 barriers instead of 91.  Keys are distinct (the suffix number is part of
// them), padding is all ones.
template <int G, int THREADS>
__device__ __forceinline__ void sm_bitonic_steps(unsigned long long *key, int N2, int k, int J, int tid)
{
    constexpr int E = 1 << G;
    const int p0 = __ffs(J >> (G - 1)) - 1;               // lowest of the G bits
    for (int t = tid; t < (N2 >> G); t += THREADS) {
        const int base = ((t >> p0) << (p0 + G)) | (t & ((1 << p0) - 1));
        const bool up = (base & k) == 0;
        unsigned long long kk[E];
#pragma unroll
        for (int m = 0; m < E; ++m) kk[m] = key[sm_phys(base | (m << p0))];
#pragma unroll
        for (int s = G - 1; s >= 0; --s) {
#pragma unroll
            for (int m = 0; m < E; ++m) {
                if (m & (1 << s)) continue;
                const int a = m, b = m | (1 << s);
                if ((kk[a] > kk[b]) == up) { const unsigned long long tk = kk[a]; kk[a] = kk[b]; kk[b] = tk; }
            }
        }
#pragma unroll
        for (int m = 0; m < E; ++m) key[sm_phys(base | (m << p0))] = kk[m];
    }
}

// one workgroup of THREADS threads builds the array of one text of up to MAXN bytes (the body of the kernels below)
template <int MAXN, int THREADS>
__device__ __forceinline__ void small_sa_block(const uint8_t *__restrict__ T, uint32_t *__restrict__ SA, int n,
                                               uint32_t *__restrict__ rounds_out)
{
    __shared__ unsigned long long key[MAXN];
    __shared__ uint16_t rnk[MAXN + 2];
    __shared__ uint16_t grp[MAXN + 2];                // counting rounds: the first place of the tied group a place belongs to
    __shared__ uint8_t txt[MAXN + 8];
    __shared__ uint32_t scan_lds[THREADS / WAVE + 1];
    __shared__ uint32_t s_shift[THREADS + 1];
    __shared__ uint32_t s_groups;
    constexpr int PER = MAXN / THREADS;                 // places per thread in the rank scan
    const int tid = threadIdx.x;
    int N2 = 2;
    while (N2 < n) N2 <<= 1;
    const int ib = n <= 4096 ? 12 : 13, rb = ib + 1;
    const int comps = n <= 4096 ? 4 : 3;                  // ranks per key in the later rounds: 4 x 13 + 12 = 64, 3 x 14 + 13 = 55 bits
    const unsigned long long idx_mask = (1ull << ib) - 1;

    for (int i = tid; i < n + 8; i += THREADS) txt[i] = i < n ? T[i] : (uint8_t)0;
    __syncthreads();
    // first key: the suffix's first 6 symbols as one base-257 number (symbol = byte + 1, 0 past the end), 49 bits
    for (int i = tid; i < N2; i += THREADS) {
        unsigned long long k = ~0ull;                     // padding sorts behind every suffix
        if (i < n) {
            k = 0;
#pragma unroll
            for (int j = 0; j < 6; ++j) k = k * 257ull + (unsigned long long)(i + j < n ? (unsigned)txt[i + j] + 1u : 0u);
            k = (k << ib) | (unsigned long long)i;
        }
        key[sm_phys(i)] = k;
    }
    __syncthreads();

    int h = 6;
    uint32_t rounds = 0;
    bool light = false;                                   // the last round ordered the groups by counting: no sort this time
    for (;;) {
        // ---- bitonic sort, ascending; up to three compare-exchange steps per barrier, the groups of three aligned from
        //      the bottom (bits 0-2, 3-5, 6-8, ...: the access patterns sm_phys is made for) ----
        for (int k = 2; k <= N2 && !light; k <<= 1) {
            int j = k >> 1;
            while (j > 0) {
                const int left = 32 - __clz(j);            // steps left in this stage: j, j / 2, ..., 1
                const int g = left % 3 ? left % 3 : 3;
                if (g == 3) sm_bitonic_steps<3, THREADS>(key, N2, k, j, tid);
                else if (g == 2) sm_bitonic_steps<2, THREADS>(key, N2, k, j, tid);
                else sm_bitonic_steps<1, THREADS>(key, N2, k, j, tid);
                j >>= g;
                __syncthreads();
            }
        }
        // ---- ranks: a group = run of equal keys, rank = place of its first member + 1 (0 = past the end of the text) ----
        uint32_t last_head = 0, heads = 0;                // (last_head: place + 1 of the latest group start at or before my places)
        uint32_t mine[PER];
        uint16_t who[PER];
        unsigned long long prev = tid ? key[sm_phys(tid * PER - 1)] >> ib : 0ull;
        uint16_t prev_g = light && tid ? grp[tid * PER - 1] : (uint16_t)0;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int p = tid * PER + r;
            const unsigned long long e = p < N2 ? key[sm_phys(p)] : ~0ull;
            const uint16_t g = light && p < n ? grp[p] : (uint16_t)0;
            const bool head = p < n && (p == 0 || (e >> ib) != prev || g != prev_g);
            prev = e >> ib;
            prev_g = g;
            who[r] = (uint16_t)(e & idx_mask);
            if (head) { last_head = (uint32_t)p + 1u; ++heads; }
            mine[r] = last_head;
        }
        uint32_t all_max, all_heads;
        const uint32_t before = block_incl_max<THREADS>(last_head, scan_lds, &all_max);      // includes my own places
        (void)all_max;
        // the group start in front of my first place is the inclusive max of the threads before me: shift by one thread
        s_shift[tid + 1] = before;
        if (tid == 0) s_shift[0] = 0;
        __syncthreads();
        const uint32_t carry = s_shift[tid];
        (void)block_excl_sum<THREADS>(heads, scan_lds, &all_heads);
        if (tid == 0) s_groups = all_heads;
        uint32_t my_off = 0;                              // largest distance of one of my places from its group's first place
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int p = tid * PER + r;
            if (!mine[r]) mine[r] = carry;
            if (p < n) {
                rnk[who[r]] = (uint16_t)mine[r];
                my_off = max(my_off, (uint32_t)p + 1u - mine[r]);
            }
        }
        uint32_t max_off;
        (void)block_incl_max<THREADS>(my_off, scan_lds, &max_off);      // (its barriers also publish rnk and s_groups)
        if ((int)s_groups == n || h >= n) break;          // every suffix has a place of its own (h >= n cannot leave ties)
        if (max_off < (uint32_t)SM_LIGHT) {
            // ---- counting round: every tied group is short.  The suffixes stay at their places except inside a group, where
            //      a member's new place is the group's first place + the number of members with a smaller key
            //      (rank[i + h], rank[i + 2h] [, rank[i + 3h]], i) -- no sort of the whole array ----
            unsigned long long sub[PER];
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int p = tid * PER + r;
                if (p < n) {
                    const int i = who[r];
                    unsigned long long k = 0;
                    for (int m = 1; m < comps; ++m) k = (k << rb) | (unsigned long long)(i + m * h < n ? rnk[i + m * h] : (uint16_t)0);
                    sub[r] = (k << ib) | (unsigned long long)i;
                    key[sm_phys(p)] = sub[r];
                    grp[p] = (uint16_t)(mine[r] - 1u);
                }
            }
            if (tid == 0) grp[n] = (uint16_t)0xFFFF;       // ends the last group's walk
            __syncthreads();
            uint16_t dest[PER];
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int p = tid * PER + r;
                if (p < n) {
                    const uint16_t g = (uint16_t)(mine[r] - 1u);
                    uint32_t below = 0;
                    for (int q = g; grp[q] == g; ++q) below += key[sm_phys(q)] < sub[r] ? 1u : 0u;
                    dest[r] = (uint16_t)(g + below);
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int p = tid * PER + r;
                if (p < n) key[sm_phys(dest[r])] = sub[r];
            }
            __syncthreads();
            light = true;
            h *= comps;
            ++rounds;
            continue;
        }
        // ---- next round: (rank[i], rank[i + h], rank[i + 2h] [, rank[i + 3h]]) -- the compared length triples / quadruples ----
        for (int i = tid; i < N2; i += THREADS) {
            unsigned long long k = ~0ull;                  // (a real key is smaller: its leading rank is at most n = 2^ib)
            if (i < n) {
                k = 0;
                for (int m = 0; m < comps; ++m) k = (k << rb) | (unsigned long long)(i + m * h < n ? rnk[i + m * h] : (uint16_t)0);
                k = (k << ib) | (unsigned long long)i;
            }
            key[sm_phys(i)] = k;
        }
        __syncthreads();
        light = false;
        h *= comps;
        ++rounds;
    }
    for (int p = tid; p < n; p += THREADS) SA[1 + p] = (uint32_t)(key[sm_phys(p)] & idx_mask);
    if (tid == 0) { SA[0] = (uint32_t)n; if (rounds_out) *rounds_out = rounds; }
}

__global__ __launch_bounds__(SM_THREADS) void k_small_sa(const uint8_t *__restrict__ T, uint32_t *__restrict__ SA, int n,
                                                         uint32_t *__restrict__ rounds_out)
{
    small_sa_block<SM_MAX_N, SM_THREADS>(T, SA, n, rounds_out);
}

// MANY small texts in one launch, one workgroup each (sa_amd_saca_batch: a caller that indexes thousands of short strings --
// the reference's own test domain, src/tests.rs:13-17 -- pays one launch and one synchronisation for all of them, and the
// chip works on 256 texts at a time).  desc[b] = { byte offset of text b, byte offset of its array (a multiple of 4), n, - }
// relative to the two bases; the texts need no alignment (byte loads).
// the light shape for a single text of up to SM_LITE_N bytes (four waves to start and to wait for instead of sixteen)
__global__ __launch_bounds__(SM_LITE_THREADS) void k_small_sa_lite(const uint8_t *__restrict__ T, uint32_t *__restrict__ SA, int n,
                                                       uint32_t *__restrict__ rounds_out)
{
    small_sa_block<SM_LITE_N, SM_LITE_THREADS>(T, SA, n, rounds_out);
}

// MAXN / THREADS: the shape; the host gives the texts of up to SM_LITE_N bytes to the light one (256 threads, 15 KB of LDS:
// eight workgroups per CU instead of one) and launches it over its own descriptor list.
template <int MAXN, int THREADS>
__global__ __launch_bounds__(THREADS) void k_small_sa_batch(const uint8_t *__restrict__ tbase, uint8_t *__restrict__ sbase,
                                                            const uint4 *__restrict__ desc)
{
    const uint4 d = desc[blockIdx.x];
    small_sa_block<MAXN, THREADS>(tbase + d.x, (uint32_t *)(sbase + d.y), (int)d.z, nullptr);
}

}  // namespace sa
