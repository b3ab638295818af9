// kernels/extras.hpp -- the "next" rows of SURVEY.md section 8f that sit directly on either side of the
// construction path, as HIP kernels working on the device-resident text and suffix array:
//   k_bucket_table     reference src/sa.rs:89-119 (enable_buckets): right edge of every 1-/2-byte
//                      prefix bucket of the SA, 65 793 entries, read off a device-resident suffix array
//   k_bigram_hist/scan the same table from the text alone, as the reference builds it (src/sa.rs:96-116)
//   k_ci_*             reference src/sa.rs:72-84 (check_integrity) in its linear-time form
#pragma once
#include "common.hpp"

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

// ---- the bucket table as the reference builds it: from the TEXT alone (src/sa.rs:96-116, no suffix array) ----
// 65 536 bigram bins do not fit one CU's LDS as 32-bit counters (256 KiB against 160), and 16-bit halves of a packed word
// cannot be spilled safely (a carry out of the low half is visible in the high half until it is taken back).  So the bins are
// split in two halves by the low bit of the SECOND byte (the top bit of the first would leave half the workgroups idle on
// ASCII text) and a PAIR of workgroups reads the same chunks of the text at the same time, each counting its half in 128 KiB of
// LDS: one LDS atomic per bigram, the text read once from HBM and once more from the cache.  The non-zero bins of a workgroup
// are then added to the global table (hist: 65 536 words, zeroed by the caller).
constexpr int BG_THREADS = 1024;
constexpr int BG_BINS = 32768;
constexpr int BG_MAX_PAIRS = 128;                 // 256 workgroups: one per CU (128 KiB of LDS each)
constexpr int64_t BG_MIN_CHUNK = 1 << 18;         // text bytes per pair below which fewer pairs are launched

__device__ __forceinline__ void bigram_count(uint32_t *h, uint32_t part, uint32_t c0, uint32_t c1)
{
    if ((c1 & 1u) == part) atomicAdd(&h[(c0 << 7) | (c1 >> 1)], 1u);
}

__global__ __launch_bounds__(BG_THREADS) void k_bigram_hist(const uint8_t *__restrict__ T, int64_t n, int pairs, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[BG_BINS];
    for (int i = threadIdx.x; i < BG_BINS; i += BG_THREADS) h[i] = 0;
    __syncthreads();
    const uint32_t part = blockIdx.x & 1u;
    const int pair = (int)(blockIdx.x >> 1);
    const int64_t nb = n - 1;                                             // bigrams: positions 0 .. n - 2
    // positions in front of the first 16-byte aligned address, then whole groups of 16 positions (a uint4 + the byte behind
    // it), then the rest: head and rest byte by byte, by the first pair
    int64_t head = (int64_t)((16u - (uint32_t)((uintptr_t)T & 15u)) & 15u);
    if (head > nb) head = nb > 0 ? nb : 0;
    const int64_t groups = nb > head ? (nb - head) / 16 : 0;
    const int64_t rest0 = head + groups * 16;
    if (pair == 0) {
        for (int64_t p = threadIdx.x; p < head; p += BG_THREADS) bigram_count(h, part, T[p], T[p + 1]);
        for (int64_t p = rest0 + threadIdx.x; p < nb; p += BG_THREADS) bigram_count(h, part, T[p], T[p + 1]);
    }
    const uint4 *__restrict__ V = (const uint4 *)(T + head);
    for (int64_t g = (int64_t)pair * BG_THREADS + threadIdx.x; g < groups; g += (int64_t)pairs * BG_THREADS) {
        const uint4 v = V[g];
        const uint32_t nx = T[head + g * 16 + 16];                        // (g < groups: position head + 16 g + 15 has a successor)
        const uint32_t w[5] = { v.x, v.y, v.z, v.w, nx };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t x = w[k], y = w[k + 1];
            bigram_count(h, part, x & 255u, (x >> 8) & 255u);
            bigram_count(h, part, (x >> 8) & 255u, (x >> 16) & 255u);
            bigram_count(h, part, (x >> 16) & 255u, x >> 24);
            bigram_count(h, part, x >> 24, y & 255u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BG_BINS; i += BG_THREADS) {
        const uint32_t c = h[i];
        if (c) atomicAdd(&hist[(((uint32_t)i >> 7) << 8) | (((uint32_t)i & 127u) << 1) | part], c);
    }
}

// bigram counts -> the table of reference src/sa.rs:94-116: entry 0 = the empty suffix (src/sa.rs:98), entry c0 * 257 + 1 =
// the suffix that is the lone last byte c0 (src/sa.rs:106-108), entry c0 * 257 + c1 + 2 = bigram (c0, c1) (src/sa.rs:103), then
// the inclusive prefix sum (src/sa.rs:112-116).  ONE workgroup; hist and bkt may be the same memory (hist in the first 65 536
// words of bkt): every count is read before the barrier of the block scan, every entry written behind it.
constexpr int BGS_THREADS = 1024;
constexpr int BGS_ITEMS = 65;
static_assert(BGS_THREADS * BGS_ITEMS >= BKT_LEN, "one pass over the table");
__global__ __launch_bounds__(BGS_THREADS) void k_bigram_scan(const uint32_t *hist, const uint8_t *__restrict__ T, int64_t n, uint32_t *bkt)
{
    __shared__ uint32_t lds[BGS_THREADS / WAVE + 1];
    const uint32_t last = n > 0 ? (uint32_t)T[n - 1] : 0xffffffffu;
    uint32_t v[BGS_ITEMS];
    uint32_t sum = 0;
    const uint32_t e0 = threadIdx.x * BGS_ITEMS;
#pragma unroll
    for (int k = 0; k < BGS_ITEMS; ++k) {
        const uint32_t e = e0 + (uint32_t)k;
        uint32_t c = 0;
        if (e == 0) c = 1;
        else if (e < (uint32_t)BKT_LEN) {
            const uint32_t c0 = (e - 1u) / 257u, r = (e - 1u) % 257u;
            c = r == 0 ? (c0 == last ? 1u : 0u) : hist[c0 * 256u + r - 1u];
        }
        sum += c;
        v[k] = sum;
    }
    uint32_t total;
    const uint32_t base = block_excl_sum<BGS_THREADS>(sum, lds, &total);
#pragma unroll
    for (int k = 0; k < BGS_ITEMS; ++k) {
        const uint32_t e = e0 + (uint32_t)k;
        if (e < (uint32_t)BKT_LEN) bkt[e] = base + v[k];
    }
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

// ---- the same check at streaming cost (round 3) ----
// The inverse permutation comes from the engine's binned scatter (two radix passes on the top bits of the suffix position,
// then windows of the rank array assembled in LDS: host/pipeline.hpp, scatter_binned) instead of n random 4-byte stores
// that each cost a read-modify-write of an HBM burst; the check pass fetches ONE random line of the rank array per slot
// -- rank[b] and rank[b + 1] are neighbours, and slot i's rank[b + 1] is slot i + 1's rank[a + 1], handed over by a wave
// shuffle, like T[b] -- instead of three.
__global__ __launch_bounds__(256) void k_ci_range(const uint32_t *__restrict__ SA, int64_t n, uint32_t *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    bool over = false, bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= n; i += stride) {
        const uint32_t v = SA[i];
        over |= (int64_t)v > n;
        bad |= ((int64_t)v == n) != (i == 0);             // the empty suffix sits in slot 0 and nowhere else
    }
    if (__any(over) && lane_id() == 0) atomicOr(flags, 1u);
    if (__any(bad) && lane_id() == 0) atomicOr(flags, 2u);
}

// Where the first bytes change along the array.  The claimed array is not trusted: the 257 binary searches (first slot
// whose suffix starts with a byte >= c) only PROPOSE boundaries, a running maximum makes them monotone, and
// k_ci_first_bytes then proves them: if rank is a bijection onto the slots 1 .. n and every suffix v sits inside the slot range
// of its own first byte, the ranges (disjoint, covering) hold exactly the suffixes of their byte -- the first bytes are
// non-decreasing along the array -- and two neighbouring slots have equal first bytes iff no boundary lies between them.
__global__ __launch_bounds__(512) void k_ci_starts(const uint8_t *__restrict__ T, const uint32_t *__restrict__ SA, int64_t n,
                                                    uint32_t *__restrict__ starts, uint32_t *__restrict__ bitmap)
{
    __shared__ uint32_t lds[512 / WAVE + 1];
    const int c = threadIdx.x;
    uint32_t s = 0;
    if (c <= 256) {
        int64_t lo = 1, hi = n + 1;
        if (c == 256) lo = n + 1;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            const uint32_t v = SA[mid];
            const uint32_t cv = (int64_t)v < n ? (uint32_t)T[v] : 0u;       // (an entry out of place was flagged by k_ci_range)
            if (cv < (uint32_t)c) lo = mid + 1; else hi = mid;
        }
        s = (uint32_t)lo;
    }
    uint32_t all;
    const uint32_t m = block_incl_max<512>(c <= 256 ? s : 0u, lds, &all);
    if (c <= 256) {
        starts[c] = c == 256 ? (uint32_t)(n + 1) : m;
        if (c < 256 && (int64_t)m <= n) atomicOr(&bitmap[m >> 5], 1u << (m & 31u));
    }
}

// text order, streaming: suffix v must sit in the slot range of its first byte
__global__ __launch_bounds__(256) void k_ci_first_bytes(const uint8_t *__restrict__ T, int64_t n, const uint32_t *__restrict__ rank,
                                                         const uint32_t *__restrict__ starts, uint32_t *__restrict__ flags)
{
    __shared__ uint32_t st[257];
    for (int i = threadIdx.x; i < 257; i += 256) st[i] = starts[i];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    bool bad = false;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += stride) {
        const uint32_t c = T[v], r = rank[v];
        bad |= r < st[c] || r >= st[c + 1];
    }
    if (__any(bad) && lane_id() == 0) atomicOr(flags, 2u);
}

constexpr int CI_THREADS = 256;
constexpr int CI_ITEMS = 8;
struct __attribute__((packed, aligned(4))) CiPair { uint32_t x, y; };      // two neighbouring ranks, fetched by one 4-byte aligned 8-byte load
// slot order: rank[v] = slot of suffix v for v < n (written by the binned scatter; the slab has n + 1 entries), the empty
// suffix (v = n) has rank 0.  Slot i: rank[SA[i]] == i (the scatter really is the inverse: no value twice), and for a
// neighbour with the same first byte rank[a + 1] < rank[b + 1].  ONE gather per slot: rank[b] and rank[b + 1] together.
__global__ __launch_bounds__(CI_THREADS) void k_ci_check_shared(const uint32_t *__restrict__ SA, int64_t n, const uint32_t *__restrict__ rank,
                                                                  const uint32_t *__restrict__ bitmap, uint32_t *__restrict__ flags)
{
    const int64_t base = (int64_t)blockIdx.x * (CI_THREADS * CI_ITEMS) + 1;       // slots 1 .. n
    const int l = lane_id();
    bool bad = false;
    uint32_t b[CI_ITEMS], rb[CI_ITEMS], rb1[CI_ITEMS];
#pragma unroll
    for (int k = 0; k < CI_ITEMS; ++k) {
        const int64_t i = base + k * CI_THREADS + threadIdx.x;
        b[k] = i <= n ? SA[i] : 0xffffffffu;
    }
#pragma unroll
    for (int k = 0; k < CI_ITEMS; ++k) {
        rb[k] = 0; rb1[k] = 0;
        if ((int64_t)b[k] < n) {                            // (values >= n in these slots were flagged by k_ci_range)
            const CiPair q = *(const CiPair *)(rank + b[k]);
            rb[k] = q.x;
            rb1[k] = (int64_t)b[k] + 1 < n ? q.y : 0u;      // (the suffix behind the last byte is the empty one: rank 0)
        }
    }
#pragma unroll
    for (int k = 0; k < CI_ITEMS; ++k) {
        const int64_t i = base + k * CI_THREADS + threadIdx.x;
        // rank[a + 1] of the slot in front of mine: my left neighbour's rb1, or (first lane of a wave) fetched here
        uint32_t ra1 = (uint32_t)__shfl_up((int)rb1[k], 1, WAVE);
        if (l == 0 && i >= 2 && i <= n) {
            const uint32_t a = SA[i - 1];
            ra1 = ((int64_t)a + 1 < n) ? rank[a + 1] : 0u;
        }
        if (i > n || (int64_t)b[k] >= n) continue;          // (past the end / flagged already)
        if (rb[k] != (uint32_t)i) { bad = true; continue; }      // not the inverse: some value is there twice
        const bool new_byte = (bitmap[i >> 5] >> (i & 31)) & 1u;  // a first-byte boundary between slots i - 1 and i
        if (i == 1 || new_byte) continue;                   // (k_ci_first_bytes proves the order of the first bytes)
        if (ra1 >= rb1[k]) bad = true;
    }
    if (__any(bad) && l == 0) atomicOr(flags, 2u);
}

}  // namespace sa

// ------------------------------------------------------------------------------------------
// Batched search: contains / search_all / search_lcp of reference
// src/sa.rs:164-253 for MANY patterns at once, one wave per pattern.  The reference's API is one
// pattern per call (latency-bound binary search, CPU-appropriate); the batched form is what a GPU
// can help with.  Comparisons are done 64 bytes at a time: lane l compares byte c + l of the
// pattern with byte c + l of the suffix, a ballot finds the first difference.
// ------------------------------------------------------------------------------------------
namespace sa {

struct SuffixCmp { int ord; uint32_t lcp; };   // ord: -1 suffix < pattern, 0 equal, +1 suffix > pattern

// Rust slice ordering of s[p..] against pat (lexicographic, a proper prefix is smaller) and their lcp
__device__ __forceinline__ SuffixCmp wave_compare(const uint8_t *__restrict__ T, int64_t n, int64_t p,
                                                  const uint8_t *__restrict__ pat, int64_t plen)
{
    const int l = lane_id();
    const int64_t slen = n - p;
    const int64_t common = slen < plen ? slen : plen;
    for (int64_t c = 0; c < common; c += WAVE) {
        const int64_t i = c + l;
        const bool in = i < common;
        const uint8_t a = in ? T[p + i] : 0, b = in ? pat[i] : 0;
        const uint64_t diff = __ballot(in && a != b);
        if (diff) {
            const int first = __ffsll((unsigned long long)diff) - 1;
            const int av = __shfl((int)a, first, WAVE), bv = __shfl((int)b, first, WAVE);
            SuffixCmp r; r.ord = av < bv ? -1 : 1; r.lcp = (uint32_t)(c + first);
            return r;
        }
    }
    SuffixCmp r; r.lcp = (uint32_t)common;
    r.ord = slen < plen ? -1 : (slen > plen ? 1 : 0);
    return r;
}

constexpr int SEARCH_THREADS = 256;

__global__ __launch_bounds__(SEARCH_THREADS) void k_search_batch(
    const uint8_t *__restrict__ T, const uint32_t *__restrict__ SA, int64_t n, const uint8_t *__restrict__ pat_data,
    const int64_t *__restrict__ pat_off, int32_t count, uint8_t *__restrict__ contains, uint32_t *__restrict__ range_lo,
    uint32_t *__restrict__ range_hi, uint32_t *__restrict__ lcp_start, uint32_t *__restrict__ lcp_len,
    const uint32_t *__restrict__ bkt)          // bucket table (enable_buckets) or nullptr
{
    const int q = (int)((blockIdx.x * (int64_t)SEARCH_THREADS + threadIdx.x) / WAVE);
    if (q >= count) return;                                   // whole waves leave together
    const uint8_t *pat = pat_data + pat_off[q];
    const int64_t plen = pat_off[q + 1] - pat_off[q];
    const int64_t len = n + 1;                                // sa.len()
    // get_bucket (reference src/sa.rs:123-144): with a bucket table the binary searches start from the SA range of the
    // pattern's first one or two bytes.  Every suffix below that range is smaller than the pattern and every one above
    // it larger, so the insertion point and the match range are the same as over the whole array.
    int64_t lo = 0, hi = len;
    if (bkt && plen > 1) {
        const int idx = (int)pat[0] * 257 + (int)pat[1] + 2;
        lo = bkt[idx - 1]; hi = bkt[idx];
    } else if (bkt && plen == 1) {
        lo = bkt[(int)pat[0] * 257]; hi = bkt[(int)pat[0] * 257 + 257];
    }
    const int64_t bucket_hi = hi;
    const bool empty_bucket = bkt != nullptr && plen > 0 && lo == hi;      // (search_lcp below: reference src/sa.rs:211-222)
    // search_all, first loop (reference src/sa.rs:182-190): first i with !(pat > s[sa[i]..])
    while (lo < hi) {
        const int64_t m = lo + (hi - lo) / 2;
        if (wave_compare(T, n, (int64_t)SA[m], pat, plen).ord < 0) lo = m + 1; else hi = m;
    }
    const int64_t i = lo;
    // second loop (src/sa.rs:192-201): first j >= i whose suffix does not start with pat
    int64_t lo2 = i, hi2 = bucket_hi;
    while (lo2 < hi2) {
        const int64_t m = lo2 + (hi2 - lo2) / 2;
        if ((int64_t)wave_compare(T, n, (int64_t)SA[m], pat, plen).lcp == plen) lo2 = m + 1; else hi2 = m;
    }
    const int64_t j = lo2;
    // search_lcp without buckets (src/sa.rs:207-253): i is also the insertion point of pat
    uint32_t ls = (uint32_t)n, ll = 0;
    if (empty_bucket) {
        // With a bucket table the reference searches the pattern's (c0, c1) bucket only; when that is empty no suffix shares
        // two bytes with the pattern and it answers with the FIRST suffix of the top-level bucket of c0 (one common byte),
        // or s.len()..s.len() when that is empty too (src/sa.rs:211-222) -- not with the neighbour of the insertion point.
        const int64_t tlo = bkt[(int)pat[0] * 257], thi = bkt[(int)pat[0] * 257 + 257];
        if (thi > tlo) { ls = SA[tlo]; ll = 1; }
    } else {
        SuffixCmp cb; cb.ord = 1; cb.lcp = 0;
        if (i < len) cb = wave_compare(T, n, (int64_t)SA[i], pat, plen);
        if (i < len && cb.ord == 0) { ls = SA[i]; ll = (uint32_t)(n - (int64_t)SA[i]); }            // Ok(i): start..s.len()
        else if (i > 0 && i < len) {
            const SuffixCmp ca = wave_compare(T, n, (int64_t)SA[i - 1], pat, plen);
            if (ca.lcp > cb.lcp) { ls = SA[i - 1]; ll = ca.lcp; } else { ls = SA[i]; ll = cb.lcp; }
        } else if (i == 0) { ls = SA[0]; ll = cb.lcp; }
        else { const SuffixCmp ca = wave_compare(T, n, (int64_t)SA[i - 1], pat, plen); ls = SA[i - 1]; ll = ca.lcp; }
    }
    if (lane_id() == 0) {
        if (contains) contains[q] = (uint8_t)(j > i);         // src/sa.rs:164-170: some suffix starts with pat
        if (range_lo) range_lo[q] = (uint32_t)i;
        if (range_hi) range_hi[q] = (uint32_t)j;
        if (lcp_start) lcp_start[q] = ls;
        if (lcp_len) lcp_len[q] = ll;
    }
}

}  // namespace sa

// ------------------------------------------------------------------------------------------
// Packed on-disk format, reference src/packed_sa.rs: fixed-width bit packing
// of the SA in blocks of 128 integers with `bitpacking::BitPacker4x`, bits = ceil(log2(length))
// (src/packed_sa.rs:127-129).  PARITY UNPINNED AT BYTE LEVEL: the block layout belongs to the
// external `bitpacking 0.8` crate, whose source is not under /root/reference; it is restated here
// from its published description (SIMD-BP128 "vertical" layout): a block is 32 rows of 4 lanes,
// integer 4 i + c is row i of lane c; every lane is an independent little-endian bit stream of
// 32 * bits bits (row i at bit offset i * bits); the output is `bits` 16-byte registers, register j
// holding the j-th 32-bit word of lanes 0..3.  The reference's own test only pins the round trip
// (src/tests.rs:61-76), which tests/ reproduces.
// ------------------------------------------------------------------------------------------
namespace sa {

// one thread per output 32-bit word; block b, register j, lane c -> word index (b * bits + j) * 4 + c
__global__ __launch_bounds__(256) void k_pack4x(const uint32_t *__restrict__ SA, int64_t len, int bits,
                                                 uint32_t *__restrict__ out, int64_t out_words)
{
    const int64_t wi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= out_words) return;
    const int c = (int)(wi & 3);
    const int64_t reg = wi >> 2;
    const int64_t blk = reg / bits;
    const int j = (int)(reg % bits);
    const int first_bit = 32 * j;                         // bit range [32 j, 32 j + 32) of lane c's stream
    uint32_t word = 0;
    for (int i = first_bit / bits; i < 32 && i * bits < first_bit + 32; ++i) {
        const int64_t src = blk * 128 + 4 * i + c;
        const uint64_t v = src < len ? (uint64_t)SA[src] : 0ull;     // the last block is zero padded (src/packed_sa.rs:37-39)
        const int sh = i * bits - first_bit;              // where row i starts relative to this word
        word |= sh >= 0 ? (uint32_t)(v << sh) : (uint32_t)(v >> (-sh));
    }
    out[wi] = word;
}

// one thread per integer
__global__ __launch_bounds__(256) void k_unpack4x(const uint32_t *__restrict__ in, int64_t in_words, int64_t len, int bits,
                                                   uint32_t *__restrict__ SA)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= len) return;
    const int64_t blk = idx / 128;
    const int r = (int)(idx % 128), i = r >> 2, c = r & 3;
    const int bit0 = i * bits, j = bit0 >> 5, off = bit0 & 31;
    const int64_t w0 = (blk * bits + j) * 4 + c;
    const uint64_t lo = w0 < in_words ? in[w0] : 0u;      // trimmed trailing zero bytes read as zero
    const uint64_t hi = (off + bits > 32 && w0 + 4 < in_words) ? in[w0 + 4] : 0u;
    const uint64_t both = lo | (hi << 32);
    SA[idx] = (uint32_t)((both >> off) & ((bits >= 32) ? 0xffffffffull : ((1ull << bits) - 1ull)));
}

}  // namespace sa
