// kernels/refine.hpp -- one refinement round: secondary keys (text symbols, low key bits, ranks) and the in-LDS group sort.
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
#pragma once
#include "common.hpp"
#include "keys.hpp"
#include "rerank.hpp"
#include "radix_sort.hpp"

namespace sa {

// ------------------------------------------------------------------------------------------
// Secondary key of prefix doubling.  For suffix v with offset h:
//   v + h <  n : n + ISA[v + h]     (rank of the suffix h symbols further on; ranks start at 1)
//   v + h >= n : n - 1 - v          (text ended inside the compared prefix: the shorter suffix,
//                                    i.e. the larger v, is smaller; all below every real rank)
// key = (group head << key2_bits) | key2, so one sort by key refines every group at once
// (text_key2<KS_RANK> below; the sparse variant without an ISA: sparse_key2).

// Text-keyed refinement round (used while MANY suffixes are still tied): instead of ranks -- which
// would need the ISA, n random 4-byte writes -- the secondary key is the next `s` symbols of the text
// itself, T[v+h .. v+h+s), packed like the initial keys into `kb` bits below the group head.  The
// order after the round is by h + s symbols; depth grows additively, but no rank array exists yet.
// The round that finishes a top-32-bit initial sort is the same thing with the low `kb` bits of the
// suffix's own 64-bit key as the secondary key (KS_LOWKEY).
// KS_RANK is the secondary key of a prefix-doubling round with a full ISA (step 7, dense): the rank of the suffix h
// symbols further on, n + ISA[v + h], or n - 1 - v when the text ends inside the compared prefix.
// KS_SPARSE is the same key when few suffixes are tied and no full ISA exists: the rank is looked up (sparse_rank below).
// KS_PRE (k_group_sort only): the secondary keys have been gathered into the key array already (the sparse look-up is a long
// chain of dependent loads per suffix: one thread per suffix in its own kernel, not eight per thread in the sort).
enum { KS_TEXT = 0, KS_LOWKEY = 1, KS_RANK = 2, KS_SPARSE = 3, KS_PRE = 4, KS_CHASE = 5 };   // KS_CHASE: KS_RANK keys, several look-ups per launch (k_group_sort)
struct KeySrc {
    int mode;          // KS_TEXT / KS_LOWKEY / KS_RANK / KS_SPARSE
    int64_t h;         // KS_TEXT, KS_RANK, KS_SPARSE: symbols already sorted (offset of the secondary key)
    int s;             // KS_TEXT: symbols per secondary key
    int kb;            // bits of the secondary key
    const uint32_t *isa;   // KS_RANK, KS_SPARSE
    // KS_SPARSE only:
    const uint32_t *has_isa;        // bit q: ISA[q] is valid (q has been in the tied list of a doubling round)
    const uint64_t *sorted_keys;    // the initial 64-bit keys in SA order, or
    const uint32_t *sorted_top32;   // (two-stage initial sort) only their top 32 bits; then sorted_keys is unused
    const uint32_t *sa;             // SA[1..]
    int64_t depth;                  // symbols the initial sort and the text-keyed rounds have ordered
    int top_shift;
    int iters;                      // KS_RANK: rank look-ups per round (1 = plain doubling), see k_group_sort
    int net_min;                    // k_group_sort: a tile that owns a group of more members than this orders its groups by a bitonic network (0: never)
};

// Sparse rank lookup (few tied suffixes): no ISA is built.  rank(q) of suffix q under the current order:
//  - q has been in the tied list of a doubling round: ISA[q] (has_isa bit set by k_rr_apply);
//  - otherwise its rank is still what the initial sort and the text-keyed rounds gave it.  The sorted
//    initial keys give the slot range [lo, hi) of the suffixes sharing q's first k symbols (binary
//    search); text-keyed rounds have ordered that range by the symbols k .. depth-1, so a second binary
//    search on those symbols (read from the text) finds the first slot of q's group.  rank = slot + 1,
//    the value a dense ISA scatter would have stored.
// sorted_top32 != nullptr: the first stage sorted only the top 32 key bits (no 64-bit sorted keys exist);
// level 1 then searches those, level 2 compares ALL symbols 0 .. depth-1 through the text.
__device__ __forceinline__ uint64_t sparse_key2(const uint8_t *__restrict__ T, const uint8_t *lcode, const KeyParams &P, int64_t n,
                                                const KeySrc &K, uint32_t v, bool aligned8)
{
    const int64_t p = (int64_t)v + K.h;
    if (p >= n) return (uint64_t)(n - 1 - (int64_t)v);
    if ((K.has_isa[p >> 5] >> (p & 31)) & 1u) return (uint64_t)n + (uint64_t)K.isa[p];
    const uint64_t kq = text_key(T, lcode, P, n, p, P.k, aligned8);
    int64_t lo = 0, hi = n, a = 0;              // [lo, a): slots whose (top) key equals q's
    if (K.sorted_top32) {
        const uint32_t kt = (uint32_t)(kq >> K.top_shift);
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (K.sorted_top32[mid] < kt) lo = mid + 1; else hi = mid; }
        a = lo; int64_t b = n;
        while (a < b) { const int64_t mid = (a + b) >> 1; if (K.sorted_top32[mid] <= kt) a = mid + 1; else b = mid; }
    } else {
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (K.sorted_keys[mid] < kq) lo = mid + 1; else hi = mid; }
        a = lo; int64_t b = n;
        if (K.depth > P.k) while (a < b) { const int64_t mid = (a + b) >> 1; if (K.sorted_keys[mid] <= kq) a = mid + 1; else b = mid; }
    }
    const int64_t from = K.sorted_top32 ? 0 : P.k;  // symbols already decided by level 1
    if (K.depth > from && a - lo > 1) {
        int64_t l2 = lo, h2 = a;                 // inside [lo, a): first slot not smaller on symbols from .. depth-1
        while (l2 < h2) {
            const int64_t mid = (l2 + h2) >> 1;
            const int64_t sfx = (int64_t)K.sa[mid];
            bool less = false;                   // suffix at mid < q on those symbols?
            for (int64_t i = from; i < K.depth; ++i) {
                const uint64_t ca = code_at(T, lcode, n, sfx + i), cb = code_at(T, lcode, n, p + i);
                if (ca != cb) { less = ca < cb; break; }
            }
            if (less) l2 = mid + 1; else h2 = mid;
        }
        lo = l2;
    }
    return (uint64_t)n + (uint64_t)lo + 1u;
}

template <int MODE>
__device__ __forceinline__ uint64_t text_key2(const uint8_t *__restrict__ T, const uint8_t *lcode, const KeyParams &P, int64_t n,
                                              const KeySrc &K, uint32_t v, bool aligned8)
{
    if (MODE == KS_PRE) return 0;
    if (MODE == KS_SPARSE) return sparse_key2(T, lcode, P, n, K, v, aligned8);
    if (MODE == KS_RANK || MODE == KS_CHASE) {
        const int64_t p = (int64_t)v + K.h;
        return p < n ? (uint64_t)n + (uint64_t)K.isa[p] : (uint64_t)(n - 1 - (int64_t)v);
    }
    if (MODE == KS_TEXT) return text_key(T, lcode, P, n, (int64_t)v + K.h, K.s, aligned8);
    if (P.bits > 0 && K.kb % P.bits == 0) {
        // bit-field keys: the low bits ARE the last kb / bits symbols of the key
        const int ns = K.kb / P.bits;
        return text_key(T, lcode, P, n, (int64_t)v + (P.k - ns), ns, aligned8);
    }
    return text_key(T, lcode, P, n, (int64_t)v, P.k, aligned8) & ((1ull << K.kb) - 1ull);
}

// plain gather (the tied list then goes through the global radix sort): keys[j] = (group head << kb) | key2
template <int MODE>
__global__ __launch_bounds__(GK_THREADS) void k_gather_textkey(const uint32_t *__restrict__ V, const uint32_t *__restrict__ G,
                                                                const uint8_t *__restrict__ T, KeyParams P, int64_t m, int64_t n,
                                                                KeySrc K, uint64_t *__restrict__ keys)
{
    __shared__ uint8_t lcode[256];
    lcode[threadIdx.x] = P.code[threadIdx.x];
    __syncthreads();
    const bool aligned8 = (((uintptr_t)T) & 7) == 0;
    const int64_t stride = (int64_t)gridDim.x * GK_THREADS;
    int64_t j = (int64_t)blockIdx.x * GK_THREADS + threadIdx.x;
    if (MODE != KS_SPARSE) {
        // four members per thread and step: their look-ups (one random 64-byte sector each) are in flight together -- with
        // one at a time the kernel ran at the latency of a memory access, not at the rate of the memory system
        // (Fibonacci word, 268 M rank look-ups: 6.8 ms)
        constexpr int B = 4;
        for (; j + (B - 1) * stride < m; j += B * stride) {
            uint32_t v[B], g[B];
            uint64_t k2[B];
#pragma unroll
            for (int q = 0; q < B; ++q) { v[q] = V[j + q * stride]; g[q] = G[j + q * stride]; }
#pragma unroll
            for (int q = 0; q < B; ++q) k2[q] = text_key2<MODE>(T, lcode, P, n, K, v[q], aligned8);
#pragma unroll
            for (int q = 0; q < B; ++q) keys[j + q * stride] = ((uint64_t)g[q] << K.kb) | k2[q];
        }
    }
    for (; j < m; j += stride)
        keys[j] = ((uint64_t)G[j] << K.kb) | text_key2<MODE>(T, lcode, P, n, K, V[j], aligned8);
}

// ------------------------------------------------------------------------------------------
// k_group_sort: gather of the secondary keys FUSED with the refinement of the small groups.
// After the initial sort the tied suffixes sit in millions of tiny groups (mean size 5-10 on
// English-like text), so a global 8-pass radix sort of (group head, key2) mostly re-establishes
// an order it already has.  The tied list is in slot order with every group contiguous, and the
// offset of an element inside its group is its slot minus the group-head slot, so group starts
// are known without a scan.  One workgroup takes GS_TILE consecutive list elements:
//   1. every thread loads (V, G, U) of its elements and gathers their secondary keys from the
//      text (the random accesses of the round; all of a thread's loads are in flight together);
//      keys and a group-start bitmap go to LDS;
//   2. a group that lies completely inside the tile and has at most GS_CAP members is OWNED: every
//      member counts the members that order before it (key, then list position) -- an LDS
//      broadcast read per step, wave cost = its largest group -- which is its place in the group;
//   3. (key, suffix) pairs are permuted through LDS and stored coalesced.  Members of groups that
//      are not owned keep their place and are flagged; they go through the global radix sort.
// Algorithmic traffic per element: 12 B read + the text gather, 13 B written.
// ------------------------------------------------------------------------------------------
constexpr int GS_THREADS = 256;
constexpr int GS_ITEMS = 8;
constexpr int GS_TILE = GS_THREADS * GS_ITEMS;
constexpr int GS_WORDS = GS_TILE / 64;
constexpr int GS_CAP = 1024;        // upper bound of the run-time group-size cap
constexpr int GB_LIST_MIN = 256;    // k_group_sort_big: a group that runs on beyond its tile is listed when it has this many members inside it

template <int MODE>
__global__ __launch_bounds__(GS_THREADS) void k_group_sort(const uint32_t *Vin, const uint32_t *__restrict__ G,
                                                            const uint32_t *__restrict__ U, const uint8_t *__restrict__ T, KeyParams P,
                                                            int64_t m, int64_t n, KeySrc K, uint64_t *keys,
                                                            uint32_t *Vout, uint8_t *__restrict__ bigflag, int cap,   // Vout may be Vin
                                                            uint32_t *__restrict__ big_heads = nullptr,               // list positions of the first members of groups
                                                            uint32_t *__restrict__ n_big_heads = nullptr)             // that may be k_group_sort_big's (nullptr: not listed)
{
    __shared__ uint64_t s_key[GS_TILE];
    __shared__ uint32_t s_val[GS_TILE];
    __shared__ uint64_t s_head[GS_WORDS + 1];
    __shared__ int s_nextH[GS_WORDS + 2];          // first group start in words >= w (-1: none): a group's end is one look-up
    __shared__ uint8_t lcode[256];
    lcode[threadIdx.x] = P.code[threadIdx.x];
    __syncthreads();
    const bool aligned8 = (((uintptr_t)T) & 7) == 0;
    const int64_t base = (int64_t)blockIdx.x * GS_TILE;
    const int t = threadIdx.x;
#ifdef SA_AMD_DIAG
    const bool stamping = g_gs_stamp_on != 0 && t == 0;
    unsigned long long t_prev = stamping ? __builtin_amdgcn_s_memtime() : 0ull;
    auto stamp = [&](int phase) {
        if (stamping) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            atomicAdd(&g_phase_cycles[8 + phase], now - t_prev);
            t_prev = now;
        }
    };
#else
    auto stamp = [](int) {};
#endif
    uint32_t v[GS_ITEMS], g[GS_ITEMS], u[GS_ITEMS];
    uint64_t key[GS_ITEMS];
#pragma unroll
    for (int r = 0; r < GS_ITEMS; ++r) {
        const int64_t j = base + r * GS_THREADS + t;
        const bool valid = j < m;
        v[r] = valid ? Vin[j] : 0u;
        g[r] = valid ? G[j] : 0u;
        u[r] = valid ? U[j] : 0u;
    }
    stamp(0);      // list loads issued
#pragma unroll
    for (int r = 0; r < GS_ITEMS; ++r) {
        const int64_t j = base + r * GS_THREADS + t;
        if (MODE == KS_PRE) key[r] = j < m ? (keys[j] & ((1ull << K.kb) - 1ull)) : 0ull;
        else key[r] = j < m ? text_key2<MODE>(T, lcode, P, n, K, v[r], aligned8) : 0ull;
    }
    stamp(1);      // secondary keys gathered (includes the wait for the list loads)
    // keys of at most 53 bits are ranked as (key << 11 | tile position): unique, so one compare per member
    static_assert(GS_TILE <= 2048, "11 bits of tile position");
    const bool packed = K.kb <= 53;
#pragma unroll
    for (int r = 0; r < GS_ITEMS; ++r) {
        const int jl = r * GS_THREADS + t;
        const bool head = (base + jl >= m) || u[r] == g[r];       // past the end counts as a group start
        const uint64_t hb = __ballot(head);
        if (lane_id() == 0) s_head[jl >> 6] = hb;
        // plain rank keys (n + rank or n - 1 - v, below 2^32): ranked as 32-bit words, ties by list position through the
        // split of the counting loop at the member's own place -- one full-rate compare per step instead of a 64-bit one
        if ((MODE == KS_RANK || MODE == KS_CHASE) && K.kb <= 32) ((uint32_t *)s_key)[jl] = (uint32_t)key[r];
        else s_key[jl] = packed ? ((key[r] << 11) | (uint64_t)jl) : key[r];
    }
    if (t == 0) {
        // does a group start exactly at the first element after the tile?
        const int64_t jx = base + GS_TILE;
        s_head[GS_WORDS] = (jx >= m || U[jx] == G[jx]) ? 1ull : 0ull;
    }
    __syncthreads();
    if (wave_id() == 0) {
        const int l = lane_id();
        const uint64_t hw = l <= GS_WORDS ? s_head[l] : 0ull;
        int first = hw ? l * 64 + __builtin_ctzll(hw) : 0x7fffffff;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int bb = __shfl_down(first, o, WAVE);
            if (l + o < WAVE) first = min(first, bb);
        }
        if (l <= GS_WORDS) s_nextH[l] = first == 0x7fffffff ? -1 : first;
        if (l == 0) s_nextH[GS_WORDS + 1] = -1;
    }
    __syncthreads();
    stamp(2);      // keys + group-start bitmap in LDS, next-start table (3 barriers)
    int dest[GS_ITEMS];
    bool big[GS_ITEMS];
    // KS_CHASE ("chasing", K.iters > 1): a group whose members tie on rank[v + h] is refined further inside this launch by
    // rank[v + 2h], rank[v + 3h], ... -- members that tie on the ranks at v + h .. v + (j-1)h share j * h symbols, so the
    // look-up at v + j * h is as valid as the first one.  After the first step the members sit in LDS in sorted order with
    // the extent (first place, size) of their subgroup; every further step touches only members whose subgroup still has
    // more than one member: they fetch the next rank, count inside their subgroup (not the whole group) and move to their
    // new place.  A group that is still tied after K.iters steps shares (K.iters + 1) * h symbols: when no group of the
    // round went through the global sort, the host multiplies h by K.iters + 1 instead of 2 (long repeats: 15 rounds -> 6).
    constexpr bool CHASE = MODE == KS_CHASE;
    __shared__ uint32_t s_val2[CHASE ? GS_TILE : 1];              // second value buffer + subgroup extents (first << 16 | size)
    __shared__ uint32_t s_rng[CHASE ? 2 : 1][CHASE ? GS_TILE : 1];
    int start_[GS_ITEMS];
    uint32_t owned_mask = 0;
    // end of the group of tile position jl = next group start after jl (GS_TILE: the tile ends with the group; -1: it goes on)
    auto group_end = [&](int jl) -> int {
        const int wi = jl >> 6;
        const uint64_t wbits = (jl & 63) == 63 ? 0ull : (s_head[wi] & (~0ull << ((jl & 63) + 1)));
        return wbits ? wi * 64 + __builtin_ctzll(wbits) : s_nextH[wi + 1];
    };
    // A tile with a LARGE owned group: counting costs a wave its largest group per member -- 8 members x 1 000 steps for a group of
    // a thousand, 100 us on the one compute unit that has the tile, which is what a launch over a few hundred tiles (a text of a
    // few MiB) then takes.  Such a tile sorts (first place of the member's group or, not owned, its own place | key | place)
    // composites through a bitonic network instead: 66 steps of 1 024 exchanges, 5 us whatever the groups -- every owned group
    // comes out ordered in its own places, everything else stays where it is.  (Keys of up to 42 bits: 11 + 42 + 11.)
    bool by_network = false;
    uint16_t *s_place = (uint16_t *)s_val;                       // (s_val is written only behind the barrier that follows the ranks)
    if (!CHASE && K.kb <= 42 && K.net_min > 0) {
        bool large = false;
#pragma unroll
        for (int r = 0; r < GS_ITEMS; ++r) {
            const int jl = r * GS_THREADS + t;
            const int start = jl - (int)(u[r] - g[r]), end = group_end(jl);
            large |= base + jl < m && start >= 0 && end >= 0 && end - start <= cap && end - start > K.net_min;
        }
        by_network = __syncthreads_or(large) != 0;
        if (by_network) {
#pragma unroll
            for (int r = 0; r < GS_ITEMS; ++r) {
                const int jl = r * GS_THREADS + t;
                const int start = jl - (int)(u[r] - g[r]), end = group_end(jl);
                const bool owned = base + jl < m && start >= 0 && end >= 0 && end - start <= cap;
                s_key[jl] = ((uint64_t)(owned ? start : jl) << 53) | (owned ? key[r] << 11 : 0ull) | (uint64_t)jl;
            }
            __syncthreads();
            // (exchange x of a step pairs elements inside the 128-element block x / 64 as long as the distance is at most 64, and a
            // wave's 64 exchanges are one such block: 56 of the 66 steps need no workgroup barrier, only the wave's own LDS traffic
            // to have landed)
            for (int k = 2; k <= GS_TILE; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
                    for (int x = t; x < GS_TILE / 2; x += GS_THREADS) {
                        const int lo = ((x & ~(j - 1)) << 1) | (x & (j - 1)), hi = lo | j;
                        const uint64_t a = s_key[lo], c = s_key[hi];
                        if ((a > c) == ((lo & k) == 0)) { s_key[lo] = c; s_key[hi] = a; }
                    }
                    const int next_j = j > 1 ? (j >> 1) : k;                 // (the next step's distance; behind the last step: a barrier)
                    if (j > 64 || next_j > 64 || (j == 1 && k == GS_TILE)) __syncthreads();
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
#pragma unroll
            for (int r = 0; r < GS_ITEMS; ++r) {
                const int p = r * GS_THREADS + t;
                s_place[(int)(s_key[p] & 2047u)] = (uint16_t)p;
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < GS_ITEMS; ++r) {
        const int jl = r * GS_THREADS + t;
        const bool valid = base + jl < m;
        const int start = jl - (int)(u[r] - g[r]);               // negative: the group starts before the tile
        const int end = group_end(jl);
        const bool owned = valid && start >= 0 && end >= 0 && end - start <= cap;
        int rank = 0;
        start_[r] = start;
        uint32_t rng = ((uint32_t)jl << 16) | 1u;                // not owned: stays where it is, never tied
        if (by_network) {
            rank = owned ? (int)s_place[jl] - start : 0;
        } else if (owned && CHASE && K.kb <= 32) {
            const uint32_t *k32 = (const uint32_t *)s_key;
            const uint32_t mine = (uint32_t)key[r];
            int lt_a = 0, le_a = 0, lt_b = 0, le_b = 0;               // members before / behind me with a smaller, smaller-or-equal key
            for (int i = start; i < jl; ++i) { const uint32_t k = k32[i]; lt_a += k < mine ? 1 : 0; le_a += k <= mine ? 1 : 0; }
            for (int i = jl + 1; i < end; ++i) { const uint32_t k = k32[i]; lt_b += k < mine ? 1 : 0; le_b += k <= mine ? 1 : 0; }
            rank = le_a + lt_b;                                        // an equal key before me orders before me
            const int lt = lt_a + lt_b, le = le_a + le_b + 1;          // (+1: myself)
            rng = ((uint32_t)(start + lt) << 16) | (uint32_t)(le - lt);
        } else if (owned && CHASE) {
            const uint64_t mine = (key[r] << 11) | (uint64_t)jl;
            int lt = 0, le = 0;
            for (int i = start; i < end; ++i) {
                const uint64_t k = s_key[i];
                rank += k < mine ? 1 : 0;
                lt += (k >> 11) < key[r] ? 1 : 0;
                le += (k >> 11) <= key[r] ? 1 : 0;
            }
            rng = ((uint32_t)(start + lt) << 16) | (uint32_t)(le - lt);
        } else if (owned && MODE == KS_RANK && K.kb <= 32) {
            const uint32_t *k32 = (const uint32_t *)s_key;
            const uint32_t mine = (uint32_t)key[r];
            for (int i = start; i < jl; ++i) rank += k32[i] <= mine ? 1 : 0;        // an equal key before me orders before me
            for (int i = jl + 1; i < end; ++i) rank += k32[i] < mine ? 1 : 0;
        } else if (owned && packed) {
            const uint64_t mine = (key[r] << 11) | (uint64_t)jl;
            for (int i = start; i < end; ++i) rank += s_key[i] < mine ? 1 : 0;
        } else if (owned) {
            const uint64_t mine = key[r];
            for (int i = start; i < end; ++i) {
                const uint64_t k = s_key[i];
                rank += (k < mine || (k == mine && i < jl)) ? 1 : 0;
            }
        }
        if (owned) owned_mask |= 1u << r;
        dest[r] = owned ? start + rank : jl;
        big[r] = valid && !owned;
        // first member of a group of more than GS_CAP members, or of one that runs on beyond the tile (its size is not known here:
        // k_group_sort_big looks, and leaves the short ones to k_group_sort_straddle)
        // (of the latter only those with at least GB_LIST_MIN members inside this tile: nearly every tile ends inside some small
        // group, and a hundred thousand list entries that k_group_sort_big only looks at cost more than the few large groups
        // that start in the last slots of a tile and so stay with the global sort)
        if (big_heads && big[r] && u[r] == g[r] && (end < 0 ? GS_TILE - start >= GB_LIST_MIN : end - start > GS_CAP))
            big_heads[atomicAdd(n_big_heads, 1u)] = (uint32_t)(base + jl);
        if (CHASE) { s_val[dest[r]] = v[r]; s_rng[0][dest[r]] = rng; }      // (s_val / s_rng are not read by the counts above)
    }
    stamp(3);      // group extents + rank loops
    if (CHASE) {
        // from here on a thread works on PLACES jl = r * GS_THREADS + t of the sorted tile, not on the members it loaded
        static_assert(!CHASE || GS_TILE <= 2048, "subgroup extents are packed as 16 + 16 bits, places as 11");
        uint32_t *val_cur = s_val, *val_nxt = s_val2;
        int cur = 0;
        __syncthreads();
        for (int it = 2; it <= K.iters; ++it) {
            uint32_t rg[GS_ITEMS], vq[GS_ITEMS];
            uint64_t mine[GS_ITEMS];
            bool any = false;
#pragma unroll
            for (int r = 0; r < GS_ITEMS; ++r) {
                const int q = r * GS_THREADS + t;
                rg[r] = s_rng[cur][q];
                vq[r] = val_cur[q];
                any |= (rg[r] & 0xffffu) > 1u;
            }
            if (!__syncthreads_or(any)) break;
#pragma unroll
            for (int r = 0; r < GS_ITEMS; ++r) {
                const int q = r * GS_THREADS + t;
                mine[r] = 0;
                if ((rg[r] & 0xffffu) > 1u) {
                    const int64_t p = (int64_t)vq[r] + (int64_t)it * K.h;
                    const uint64_t comp = p < n ? (uint64_t)n + (uint64_t)K.isa[p] : (uint64_t)(n - 1 - (int64_t)vq[r]);
                    mine[r] = (comp << 11) | (uint64_t)q;
                    s_key[q] = mine[r];
                }
            }
            __syncthreads();
            const int nxt = cur ^ 1;
#pragma unroll
            for (int r = 0; r < GS_ITEMS; ++r) {
                const int q = r * GS_THREADS + t;
                const int e = (int)(rg[r] & 0xffffu), a = (int)(rg[r] >> 16);
                if (e > 1) {
                    int rank = 0, lt = 0, le = 0;
                    const uint64_t comp = mine[r] >> 11;
                    for (int i = a; i < a + e; ++i) {
                        const uint64_t k = s_key[i];
                        rank += k < mine[r] ? 1 : 0;
                        lt += (k >> 11) < comp ? 1 : 0;
                        le += (k >> 11) <= comp ? 1 : 0;
                    }
                    val_nxt[a + rank] = vq[r];
                    s_rng[nxt][a + rank] = ((uint32_t)(a + lt) << 16) | (uint32_t)(le - lt);
                } else {
                    val_nxt[q] = vq[r];
                    s_rng[nxt][q] = rg[r];
                }
            }
            __syncthreads();
            { uint32_t *tmp = val_cur; val_cur = val_nxt; val_nxt = tmp; }
            cur = nxt;
        }
        stamp(4);
#pragma unroll
        for (int r = 0; r < GS_ITEMS; ++r) {
            const int jl = r * GS_THREADS + t;
            const int64_t j = base + jl;
            if (j < m) {
                // place jl belongs to the group the member loaded from jl belongs to (a group keeps its places); members of
                // one final subgroup share its first place, every other member of the group has another one
                const uint64_t low = ((owned_mask >> r) & 1u) ? (uint64_t)((int)(s_rng[cur][jl] >> 16) - start_[r]) : key[r];
                keys[j] = ((uint64_t)g[r] << K.kb) | low;
                Vout[j] = val_cur[jl];
                bigflag[j] = big[r] ? 1 : 0;
            }
        }
        stamp(5);
        return;
    }
    __syncthreads();                                             // every rank is known: the key slots can be reused
#pragma unroll
    for (int r = 0; r < GS_ITEMS; ++r) {
        s_key[dest[r]] = ((uint64_t)g[r] << K.kb) | key[r];
        s_val[dest[r]] = v[r];
    }
    __syncthreads();
    stamp(4);      // permuted through LDS (2 barriers)
#pragma unroll
    for (int r = 0; r < GS_ITEMS; ++r) {
        const int jl = r * GS_THREADS + t;
        const int64_t j = base + jl;
        if (j < m) { keys[j] = s_key[jl]; Vout[j] = s_val[jl]; bigflag[j] = big[r] ? 1 : 0; }
    }
    stamp(5);      // stores issued
}

// The groups k_group_sort could not own only because they straddle a tile boundary: one workgroup per
// boundary sorts the (at most one) group of up to GS_CAP members that contains it, the same way, on
// the keys the first kernel stored, and clears its flags.  What stays flagged are groups > GS_CAP.
constexpr int GX_THREADS = 256;
constexpr int GX_ITEMS = GS_CAP / GX_THREADS;
constexpr int GX_BITONIC_MIN = 96;          // groups of more members than this are ordered by a bitonic network, not by counting
__global__ __launch_bounds__(GX_THREADS) void k_group_sort_straddle(uint64_t *__restrict__ keys, uint32_t *__restrict__ V,
                                                                     const uint32_t *__restrict__ G, const uint32_t *__restrict__ U,
                                                                     int64_t m, uint8_t *__restrict__ bigflag, int cap, KeySrc K, int64_t n)
{
    __shared__ uint64_t s_key[GS_CAP];
    __shared__ int s_end;
    const int64_t b = ((int64_t)blockIdx.x + 1) * GS_TILE;          // first element of the next tile
    if (b >= m) return;
    const uint32_t ub = U[b], gb = G[b];
    if (ub == gb) return;                                            // a group starts here: nothing straddles
    const int64_t span = (int64_t)(ub - gb);                         // members before the boundary
    if (span >= cap) return;
    const int64_t start = b - span;
    const int t = threadIdx.x;
    if (t == 0) s_end = cap + 1;
    __syncthreads();
    // end of the group: the first group start after the boundary, at most GS_CAP from `start`
    // (searched 256 positions at a time: most groups end within the first few)
    for (int i0 = (int)span + 1; i0 <= cap; i0 += GX_THREADS) {
        const int i = i0 + t;
        const int64_t j = start + i;
        if (i <= cap && (j >= m || U[j] == G[j])) atomicMin(&s_end, i);
        __syncthreads();
        if (s_end <= cap) break;                                     // uniform: every thread reads the same value
        __syncthreads();
    }
    const int size = s_end;
    if (size > cap) return;
    // K.mode == KS_CHASE: the same chase as in k_group_sort, places relative to the group's first member
    const bool chase = K.mode == KS_CHASE;
    const uint64_t kmask = K.kb >= 64 ? ~0ull : ((1ull << K.kb) - 1ull);
    __shared__ uint32_t s_v[2][GS_CAP], s_rg[2][GS_CAP];
    uint64_t key[GX_ITEMS]; uint32_t v[GX_ITEMS];
    static_assert(GS_CAP <= 1024, "10 bits of member index");
#pragma unroll
    for (int r = 0; r < GX_ITEMS; ++r) {
        const int i = r * GX_THREADS + t;
        key[r] = 0; v[r] = 0;
        if (i < size) {
            key[r] = keys[start + i]; v[r] = V[start + i];
            s_key[i] = chase ? (((key[r] & kmask) << 10) | (uint64_t)i) : key[r];
        }
    }
    __syncthreads();
    if (!chase && size > GX_BITONIC_MIN && K.kb <= 54) {
        // A group of a thousand members ranked by counting is a million comparisons on ONE compute unit (100 us: at 1 MiB of English
        // text this kernel took longer than k_group_sort itself): (key2 << 10 | member) composites -- distinct, so the order is the
        // stable one -- go through a bitonic network in LDS instead, 55 steps of 512 exchanges for 1 024 members.
        int P = 128;
        while (P < size) P <<= 1;
#pragma unroll
        for (int r = 0; r < GX_ITEMS; ++r) {
            const int i = r * GX_THREADS + t;
            if (i < size) { s_key[i] = ((key[r] & kmask) << 10) | (uint64_t)i; s_v[0][i] = v[r]; }
            else if (i < P) s_key[i] = ~0ull;
        }
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int x = t; x < P / 2; x += GX_THREADS) {
                    const int lo = ((x & ~(j - 1)) << 1) | (x & (j - 1)), hi = lo | j;
                    const uint64_t a = s_key[lo], c = s_key[hi];
                    if ((a > c) == ((lo & k) == 0)) { s_key[lo] = c; s_key[hi] = a; }
                }
                // (distances of at most 64 stay inside a wave's own 128-element blocks: see k_group_sort)
                const int next_j = j > 1 ? (j >> 1) : k;
                if (j > 64 || next_j > 64 || (j == 1 && k == P)) __syncthreads();
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        __shared__ uint64_t s_ghead;                                // the bits above the secondary key: the group's head, the same for every member
        if (t == 0) s_ghead = key[0] & ~kmask;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GX_ITEMS; ++r) {
            const int i = r * GX_THREADS + t;
            if (i < size) {
                const uint64_t c = s_key[i];
                keys[start + i] = s_ghead | (c >> 10);
                V[start + i] = s_v[0][(int)(c & 1023u)];
                bigflag[start + i] = 0;
            }
        }
        return;
    }
    if (!chase) {
#pragma unroll
        for (int r = 0; r < GX_ITEMS; ++r) {
            const int i = r * GX_THREADS + t;
            if (i < size) {
                int rank = 0;
                for (int q = 0; q < size; ++q) {
                    const uint64_t k = s_key[q];
                    rank += (k < key[r] || (k == key[r] && q < i)) ? 1 : 0;
                }
                keys[start + rank] = key[r];
                V[start + rank] = v[r];
                bigflag[start + i] = 0;
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < GX_ITEMS; ++r) {
        const int i = r * GX_THREADS + t;
        if (i < size) {
            const uint64_t comp = key[r] & kmask, mine = (comp << 10) | (uint64_t)i;
            int rank = 0, lt = 0, le = 0;
            for (int q = 0; q < size; ++q) {
                const uint64_t k = s_key[q];
                rank += k < mine ? 1 : 0;
                lt += (k >> 10) < comp ? 1 : 0;
                le += (k >> 10) <= comp ? 1 : 0;
            }
            s_v[0][rank] = v[r];
            s_rg[0][rank] = ((uint32_t)lt << 16) | (uint32_t)(le - lt);
        }
    }
    __syncthreads();
    int cur = 0;
    for (int it = 2; it <= K.iters; ++it) {
        uint32_t rg[GX_ITEMS], vq[GX_ITEMS];
        uint64_t mine[GX_ITEMS];
        bool any = false;
#pragma unroll
        for (int r = 0; r < GX_ITEMS; ++r) {
            const int q = r * GX_THREADS + t;
            rg[r] = q < size ? s_rg[cur][q] : 1u;
            vq[r] = q < size ? s_v[cur][q] : 0u;
            any |= (rg[r] & 0xffffu) > 1u;
        }
        if (!__syncthreads_or(any)) break;
#pragma unroll
        for (int r = 0; r < GX_ITEMS; ++r) {
            const int q = r * GX_THREADS + t;
            mine[r] = 0;
            if ((rg[r] & 0xffffu) > 1u) {
                const int64_t p = (int64_t)vq[r] + (int64_t)it * K.h;
                const uint64_t comp = p < n ? (uint64_t)n + (uint64_t)K.isa[p] : (uint64_t)(n - 1 - (int64_t)vq[r]);
                mine[r] = (comp << 10) | (uint64_t)q;
                s_key[q] = mine[r];
            }
        }
        __syncthreads();
        const int nxt = cur ^ 1;
#pragma unroll
        for (int r = 0; r < GX_ITEMS; ++r) {
            const int q = r * GX_THREADS + t;
            if (q >= size) continue;
            const int e = (int)(rg[r] & 0xffffu), a = (int)(rg[r] >> 16);
            if (e > 1) {
                int rank = 0, lt = 0, le = 0;
                const uint64_t comp = mine[r] >> 10;
                for (int i = a; i < a + e; ++i) {
                    const uint64_t k = s_key[i];
                    rank += k < mine[r] ? 1 : 0;
                    lt += (k >> 10) < comp ? 1 : 0;
                    le += (k >> 10) <= comp ? 1 : 0;
                }
                s_v[nxt][a + rank] = vq[r];
                s_rg[nxt][a + rank] = ((uint32_t)(a + lt) << 16) | (uint32_t)(le - lt);
            } else {
                s_v[nxt][q] = vq[r];
                s_rg[nxt][q] = rg[r];
            }
        }
        __syncthreads();
        cur = nxt;
    }
    const uint64_t ghead = ((uint64_t)gb) << K.kb;
#pragma unroll
    for (int r = 0; r < GX_ITEMS; ++r) {
        const int i = r * GX_THREADS + t;
        if (i < size) {
            keys[start + i] = ghead | (uint64_t)(s_rg[cur][i] >> 16);
            V[start + i] = s_v[cur][i];
            bigflag[start + i] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_group_sort_big: groups of GS_CAP + 1 .. GB_CAP (8 192) members, one workgroup per group, ordered in LDS the way k_bucket_sort
// (kernels/bucket_sort.hpp) orders a bucket -- instead of six global radix passes over (index of the group, key2) pairs with a
// compaction in front and a scatter behind.  The members' secondary keys (at most 32 bits) are where k_group_sort left them:
// keys[j] = group head << kb | key2, V[j] = suffix, in list order.  First pass on the low 10 key bits by one LDS counter per
// digit (the order among equal digits is left to the atomics: members that agree in ALL key bits stay tied, and which of them
// takes which slot of the subgroup is nobody's business), then stable 6-bit passes (6 ballots + mbcnt) over the rest.
// heads: the list k_group_sort wrote (first members of groups it could not own); a group of at most `cap` members is
// k_group_sort_straddle's, one of more than GB_CAP stays flagged for the global sort.  sorted_members: members ordered here
// (the host counts them with the members of the global sort: no chasing after a round that had such groups).
// ------------------------------------------------------------------------------------------
constexpr int GB_ITEMS = 16;
constexpr int GB_CAP_SMALL = 256 * GB_ITEMS;       // 4 096 members: 256 threads, 37 KiB of LDS, four workgroups per CU
constexpr int GB_CAP = 512 * GB_ITEMS;             // 8 192 members: 512 threads, 70 KiB of LDS, two workgroups per CU (a 1 024-thread instance
                                                   // for 16 384 took 60 ps per member, more than the global sort's share)
constexpr int GB_ABITS = 10, GB_BBITS = 6;
// size_lo < members <= GB_THREADS * GB_ITEMS: the groups this instance orders (two launches over the same list: 256 threads
// for the groups of up to 4 096 members -- a larger workgroup spends its time in barriers on them --, 512 for the rest)
template <int GB_THREADS>
__global__ __launch_bounds__(GB_THREADS) void k_group_sort_big(uint64_t *__restrict__ keys, uint32_t *__restrict__ V, const uint32_t *__restrict__ G,
                                                               int64_t m, int kb, int size_lo, const uint32_t *__restrict__ heads,
                                                               const uint32_t *__restrict__ n_heads, uint8_t *__restrict__ bigflag,
                                                               uint32_t *__restrict__ sorted_members)
{
    constexpr int GB_CAP = GB_THREADS * GB_ITEMS;  // (shadows the constant of the large instance)
    constexpr int NWAVES = GB_THREADS / WAVE;
    constexpr int NB_A = 1 << GB_ABITS, NB_B = 1 << GB_BBITS;
    static_assert(NB_B == WAVE && NB_A % GB_THREADS == 0, "lane d of wave 0 owns digit d of a stable pass; the counters of the first are scanned by the whole workgroup");
    static_assert(GB_CAP < 65536 && GB_ITEMS % 2 == 0, "16-bit places, two to a register");
    __shared__ uint32_t lds_k[GB_CAP];
    __shared__ uint32_t lds_v[GB_CAP];
    __shared__ uint32_t cnt_a[NB_A];
    __shared__ uint16_t wave_hist[NWAVES][NB_B];
    __shared__ uint32_t digit_base[NB_B];
    __shared__ uint32_t scan_lds[NWAVES + 1];
    __shared__ int s_size;
    const int tid = threadIdx.x, l = lane_id(), w = wave_id();
    uint16_t *my_hist = wave_hist[w];
    const uint32_t count = *n_heads;
    const int lo_size = size_lo;                                   // smaller groups are not this instance's
    for (uint32_t ent = blockIdx.x; ent < count; ent += gridDim.x) {
        __syncthreads();                                           // (the previous group's LDS is free)
        const int64_t j0 = (int64_t)heads[ent];
        const uint32_t g0 = G[j0];
        if (tid == 0) s_size = GB_CAP + 1;
        __syncthreads();
        // Mine?  Two look-ups say: the member at lo_size still belongs to the group (more than lo_size members) and the one at
        // GB_CAP does not (at most GB_CAP).  Then the size: the first list position behind j0 whose group head differs -- all
        // positions looked at in one go (a chunk at a time costs a memory round trip per chunk).
        {
            const bool more = j0 + lo_size < m && G[j0 + lo_size] == g0;
            const bool fits = j0 + GB_CAP >= m || G[j0 + GB_CAP] != g0;
            if (!(more && fits)) continue;                         // (uniform: every thread reads the same two words)
        }
        {
            int first = GB_CAP + 1;
#pragma unroll
            for (int i = GB_ITEMS; i >= 0; --i) {                  // (e = GB_CAP is known to differ: the scan stops there)
                const int e = i * GB_THREADS + tid;
                if (e > lo_size && e <= GB_CAP && (j0 + e >= m || G[j0 + e] != g0)) first = e;
            }
            if (first <= GB_CAP) atomicMin(&s_size, first);
        }
        __syncthreads();
        const int size = s_size;
        if (size <= lo_size || size > GB_CAP) continue;            // (cannot happen; uniform)
        const int J = (size + GB_THREADS - 1) / GB_THREADS;
        const int e0 = w * J * WAVE + l;
        const uint32_t kmask = kb >= 32 ? 0xffffffffu : ((1u << kb) - 1u);
        uint32_t key[GB_ITEMS], val[GB_ITEMS], pp[GB_ITEMS / 2];
#define GB_POS(j) ((pp[(j) >> 1] >> (16 * ((j) & 1))) & 0xffffu)
#pragma unroll
        for (int j = 0; j < GB_ITEMS; ++j) {
            const int e = e0 + j * WAVE;
            uint32_t kx = 0, vx = 0;
            if (j < J && e < size) { kx = (uint32_t)keys[j0 + e] & kmask; vx = V[j0 + e]; }
            key[j] = kx; val[j] = vx;
        }
        const int abits = kb < GB_ABITS ? kb : GB_ABITS;
        {
            // ---- first pass: one counter per digit hands out the places ----
            const uint32_t amask = (1u << abits) - 1u;
            for (int i = tid; i < NB_A; i += GB_THREADS) cnt_a[i] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < GB_ITEMS; ++j) {
                uint32_t r = 0;
                if (j < J && (e0 + j * WAVE) < size) r = atomicAdd(&cnt_a[key[j] & amask], 1u);
                if ((j & 1) == 0) pp[j >> 1] = r; else pp[j >> 1] |= r << 16;
            }
            lds_barrier();
            {
                constexpr int BPT = NB_A / GB_THREADS;
                uint32_t c[BPT], sum = 0;
#pragma unroll
                for (int i = 0; i < BPT; ++i) { c[i] = cnt_a[tid * BPT + i]; sum += c[i]; }
                uint32_t all;
                uint32_t run = block_excl_sum_b<GB_THREADS, true>(sum, scan_lds, &all);
#pragma unroll
                for (int i = 0; i < BPT; ++i) { cnt_a[tid * BPT + i] = run; run += c[i]; }
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < GB_ITEMS; ++j) {
                if (j < J && (e0 + j * WAVE) < size) {
                    const uint32_t ps = GB_POS(j) + cnt_a[key[j] & amask];
                    lds_k[ps] = key[j];
                    lds_v[ps] = val[j];
                }
            }
        }
        // ---- stable passes over the remaining key bits, 6 at a time ----
#pragma unroll
        for (int p = 0; p < (32 - 1 + GB_BBITS - 1) / GB_BBITS; ++p) {
            const int shift = abits + p * GB_BBITS;
            if (shift >= kb) break;                                // (uniform)
            lds_barrier();
#pragma unroll
            for (int j = 0; j < GB_ITEMS; ++j) {
                const int e = e0 + j * WAVE;
                uint32_t kx = 0, vx = 0;
                if (j < J && e < size) { kx = lds_k[e]; vx = lds_v[e]; }
                key[j] = kx; val[j] = vx;
            }
            {
                // a digit that is the same for every member (the top bits of n + rank) makes the pass the identity: skip it
                const uint32_t dref = (lds_k[0] >> shift) & (uint32_t)(NB_B - 1);
                bool same = true;
#pragma unroll
                for (int j = 0; j < GB_ITEMS; ++j)
                    if (j < J && (e0 + j * WAVE) < size) same = same && ((key[j] >> shift) & (uint32_t)(NB_B - 1)) == dref;
                if (__syncthreads_and(same ? 1 : 0)) continue;     // (uniform; the bucket stays in LDS as it is)
            }
            for (int i = tid; i < NWAVES * NB_B / 2; i += GB_THREADS) ((uint32_t *)&wave_hist[0][0])[i] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < GB_ITEMS; ++j) {
                uint32_t r = 0;
                if (j < J) {                                       // (uniform)
                    const bool ok = (e0 + j * WAVE) < size;
                    const uint32_t d = (key[j] >> shift) & (uint32_t)(NB_B - 1);
                    const uint64_t okm = __ballot(ok);
                    uint32_t xlo = ~(uint32_t)okm, xhi = ~(uint32_t)(okm >> 32);
#pragma unroll
                    for (int bb = 0; bb < GB_BBITS; ++bb) {
                        const uint32_t sel = (uint32_t)((int32_t)(d << (31 - bb)) >> 31);
                        const uint64_t bal = __ballot(sel != 0);
                        xlo |= (uint32_t)bal ^ sel;
                        xhi |= (uint32_t)(bal >> 32) ^ sel;
                    }
                    const uint32_t mlo = ~xlo, mhi = ~xhi;
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                    const uint32_t prior = my_hist[d];
                    if (ok && below == 0) my_hist[d] = (uint16_t)(prior + (uint32_t)(__popc(mlo) + __popc(mhi)));
                    r = prior + below;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if ((j & 1) == 0) pp[j >> 1] = r; else pp[j >> 1] |= r << 16;
            }
            lds_barrier();
            if (w == 0) {
                uint32_t tot = 0;
#pragma unroll
                for (int ww = 0; ww < NWAVES; ++ww) {
                    const uint32_t cnt = wave_hist[ww][l];
                    wave_hist[ww][l] = (uint16_t)tot;
                    tot += cnt;
                }
                digit_base[l] = wave_incl_sum(tot) - tot;
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < GB_ITEMS; ++j) {
                if (j < J && (e0 + j * WAVE) < size) {
                    const uint32_t d = (key[j] >> shift) & (uint32_t)(NB_B - 1);
                    const uint32_t ps = GB_POS(j) + digit_base[d] + my_hist[d];
                    lds_k[ps] = key[j];
                    lds_v[ps] = val[j];
                }
            }
        }
#undef GB_POS
        __syncthreads();
        const uint64_t ghead = (uint64_t)g0 << kb;
#pragma unroll
        for (int j = 0; j < GB_ITEMS; ++j) {
            const int idx = tid + j * GB_THREADS;
            if (idx < size) {
                keys[j0 + idx] = ghead | (uint64_t)lds_k[idx];
                V[j0 + idx] = lds_v[idx];
                bigflag[j0 + idx] = 0;
            }
        }
        if (tid == 0) atomicAdd(sorted_members, (uint32_t)size);
    }
}

// ------------------------------------------------------------------------------------------
// k_finish_sorted: the round that finishes a top-32-bit initial sort, in ONE pass over the sorted
// keys and SA -- no tied-suffix lists, no re-rank kernels.  The probe only allows the 32-bit first
// stage when the suffixes tied on the top 32 bits are few and sit in tiny groups, so: a workgroup
// takes 2048 slots (+ 256 of overhang), finds the groups (runs of equal 32-bit keys) from a bitmap of
// run starts in LDS, and for every group that STARTS in its 2048 slots, ends inside the overhang
// and has <= cap members: gathers the members' low key bits from the text (KS_LOWKEY), ranks them in
// LDS (as k_group_sort does) and writes the suffixes back to their SA slots in the new order.
// Members that are still tied afterwards (equal 64-bit keys; rare) are recorded -- bit in surv_bits,
// slot of their subgroup's first member in surv_head, count per re-rank tile -- and k_surv_compact
// turns that into the (slot, group head, suffix) list the later rounds expect.  A group it cannot
// own (too large / longer overhang) only bumps counters[1]; the host then runs the general path
// (k_rr_* + refine_list) over everything: correct on any input, fast on the inputs the probe admits.
// Algorithmic traffic per slot: 4 B key + (tied: 4 B SA read, text gather, 4 B SA write).
// ------------------------------------------------------------------------------------------
constexpr int FT_THREADS = 256;
constexpr int FT_MINW = 5;                                  // waves per SIMD the register allocation leaves room for (96 VGPRs, no spills; 6 spills:
                                                            // 256 MiB random bytes 7.96 -> 7.73 ms with 5, 7.89 with 6; 1 GiB DNA 32.0 -> 33.4 ms with 6)
constexpr int FT_TILE = 2048;
constexpr int FT_XITEMS = FT_TILE / FT_THREADS + 1;         // 8 items + 1 of overhang
constexpr int FT_SPAN = FT_THREADS * FT_XITEMS;             // 2304
constexpr int FT_WORDS = FT_SPAN / 64;                      // 36

// KeyT / MODE: uint32_t keys + KS_LOWKEY (finish of the 32-bit first stage) or uint64_t keys + KS_TEXT (the first
// text-keyed round straight from the fully sorted keys).  TODO = false: a group nobody can own only bumps
// counters[1] (the host then runs the general path over everything).  TODO = true: its members are flagged in
// todo_bits (todo_cnt per 2048-slot tile, ft_head = last run start + 1 of every tile for the group-head carry);
// k_todo_compact turns them into a tied list for the general path, whose survivors join surv_bits.
template <typename KeyT, int MODE, bool TODO>
__global__ __launch_bounds__(FT_THREADS, sizeof(KeyT) == 4 ? FT_MINW : 1) void k_finish_sorted(const KeyT *__restrict__ skeys, uint32_t *SA, const uint8_t *__restrict__ T,
                                                               KeyParams P, int64_t n, KeySrc K, int cap, uint32_t *__restrict__ surv_bits,
                                                               uint32_t *__restrict__ surv_head, uint32_t *__restrict__ tile_cnt,
                                                               uint32_t *__restrict__ counters, uint32_t *__restrict__ todo_bits,
                                                               uint32_t *__restrict__ todo_cnt, uint32_t *__restrict__ ft_head)
{
    typedef typename std::conditional<MODE == KS_LOWKEY, uint32_t, uint64_t>::type Key2T;   // low key bits: key_bits - 32 <= 32
    static_assert(sizeof(KeyT) == 4 || sizeof(Key2T) == 8, "64-bit keys are staged in the 64-bit key buffer");
    constexpr int NW = FT_THREADS / WAVE;
    __shared__ Key2T s_key[FT_SPAN];
    __shared__ uint32_t s_val[FT_SPAN];
    __shared__ uint16_t s_list[FT_SPAN];          // local indices of the slots that are in a group of more than one (work list)
    __shared__ uint64_t s_head[FT_WORDS + 1];
    __shared__ uint8_t lcode[256];
    __shared__ uint32_t s_cnt[2];
    __shared__ uint32_t s_woff[FT_XITEMS * NW + 1];
    __shared__ uint32_t s_surv[FT_SPAN / 32], s_todo[FT_SPAN / 32];   // this tile's survivor / todo bits, merged into the global bitmaps once
    __shared__ KeyT s_last[FT_SPAN / 8];            // last key of every group of 8 slots (the next group's left neighbour)
    __shared__ int s_lastH[FT_WORDS], s_nextH[FT_WORDS + 1];   // last run start in words <= w / first one in words >= w (-1: none)
    lcode[threadIdx.x] = P.code[threadIdx.x];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x < FT_SPAN / 32) { s_surv[threadIdx.x] = 0; s_todo[threadIdx.x] = 0; }
    const bool aligned8 = (((uintptr_t)T) & 7) == 0;
    const int64_t base = (int64_t)blockIdx.x * FT_TILE;
    const int t = threadIdx.x, l = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());       // wave-uniform values stay in scalar registers
    const int valid_cnt = (int)(n - base < FT_SPAN ? n - base : FT_SPAN);
    const KeyT prev = base > 0 ? skeys[base - 1] : (KeyT)0;
    // ---- phase 1, every slot, cheap: run starts -> bitmap; slots in runs longer than one -> work list ----
    // A thread takes 8 CONSECUTIVE slots (16-byte loads, neighbours in registers): group t of the tile's own 2048 slots,
    // and threads 0..31 also group 256 + t of the overhang.  Its 8 run-start bits are one byte of the bitmap.
    constexpr int NG = FT_SPAN / 8;                // 288 groups of 8 slots
    uint8_t *s_hbyte = (uint8_t *)s_head;
    uint32_t tied8[2] = { 0, 0 };
    KeyT kq[2][8];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int g = part ? FT_THREADS + t : t;
        const int s0 = 8 * g;
        if (part && t >= NG - FT_THREADS) {
            // (only 32 groups of overhang)
        } else if (s0 + 8 <= valid_cnt) {
            const uint4 *src = (const uint4 *)(skeys + base + s0);
            if (sizeof(KeyT) == 4) {
                const uint4 a = src[0], c = src[1];
                const uint32_t tmp[8] = { a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w };
#pragma unroll
                for (int i = 0; i < 8; ++i) kq[part][i] = (KeyT)tmp[i];
            } else {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const uint4 a = src[h];
                    kq[part][2 * h] = (KeyT)(((uint64_t)a.y << 32) | a.x);
                    kq[part][2 * h + 1] = (KeyT)(((uint64_t)a.w << 32) | a.z);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) kq[part][i] = s0 + i < valid_cnt ? skeys[base + s0 + i] : (KeyT)0;
        }
        if (!part || t < NG - FT_THREADS) s_last[g] = kq[part][7];
    }
    if (t == 0) s_head[FT_WORDS] = 0;              // beyond the span: unknown, treated as "the run goes on"
    __syncthreads();
    uint32_t head8[2] = { 0, 0 };
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int g = part ? FT_THREADS + t : t;
        const int s0 = 8 * g;
        if (!part || t < NG - FT_THREADS) {
            KeyT left = g ? s_last[g - 1] : prev;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool head = s0 + i >= valid_cnt || kq[part][i] != left || (s0 + i == 0 && base == 0);   // past the end: a run start
                head8[part] |= (head ? 1u : 0u) << i;
                left = kq[part][i];
            }
            s_hbyte[g] = (uint8_t)head8[part];
        }
    }
    __syncthreads();
    uint32_t cnt_main = 0, cnt_over = 0;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int g = part ? FT_THREADS + t : t;
        const int s0 = 8 * g;
        if (!part || t < NG - FT_THREADS) {
            const uint32_t nb = (uint32_t)s_hbyte[g + 1] & 1u;                // (byte NG is the zero word behind the bitmap)
            const uint32_t nxt = (head8[part] >> 1) | (nb << 7);
            const int rem = valid_cnt - s0;
            const uint32_t valid8 = rem >= 8 ? 0xffu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
            tied8[part] = valid8 & ~(head8[part] & nxt);
            if (part) cnt_over = (uint32_t)__popc(tied8[part]); else cnt_main = (uint32_t)__popc(tied8[part]);
        }
    }
    uint32_t total_main;
    uint32_t off_main = block_excl_sum<FT_THREADS>(cnt_main, s_woff, &total_main);       // (two barriers inside)
    if (w == 0) {
        // the 32 overhang groups follow the 256 main ones in the list
        const uint32_t inc = wave_incl_sum(l < NG - FT_THREADS ? cnt_over : 0u);
        if (l < NG - FT_THREADS) {
            uint32_t o = total_main + inc - cnt_over;
            for (uint32_t m8 = tied8[1]; m8; m8 &= m8 - 1) s_list[o++] = (uint16_t)(8 * (FT_THREADS + t) + __builtin_ctz(m8));
        }
        if (l == WAVE - 1) s_woff[FT_XITEMS * NW] = total_main + inc;
    } else if (w == 1) {
        // per word: the last run start at or before its end, the first one at or after its beginning (so that the
        // extent of any group is two look-ups, however long the group)
        const uint64_t hw = l < FT_WORDS ? s_head[l] : 0ull;
        int last = hw ? l * 64 + 63 - __builtin_clzll(hw) : -1;
        int first = hw ? l * 64 + __builtin_ctzll(hw) : 0x7fffffff;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int a = __shfl_up(last, o, WAVE), bb = __shfl_down(first, o, WAVE);
            if (l >= o) last = max(last, a);
            if (l + o < WAVE) first = min(first, bb);
        }
        if (l < FT_WORDS) { s_lastH[l] = last; s_nextH[l] = first == 0x7fffffff ? -1 : first; }
        if (l == 0) s_nextH[FT_WORDS] = -1;
    }
    for (uint32_t m8 = tied8[0]; m8; m8 &= m8 - 1) s_list[off_main++] = (uint16_t)(8 * t + __builtin_ctz(m8));
    if (TODO && t == 0) {
        // last real run start of this tile's own 2048 slots (+1; 0: none): the group-head carry of k_todo_compact
        uint32_t lh = 0;
        const int lim = valid_cnt < FT_TILE ? valid_cnt : FT_TILE;
        for (int wi = (lim - 1) >> 6; wi >= 0 && lim > 0; --wi) {
            const int rem = lim - wi * 64;
            const uint64_t wb = s_head[wi] & (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull));
            if (wb) { lh = (uint32_t)(base + wi * 64 + 63 - __builtin_clzll(wb)) + 1u; break; }
        }
        ft_head[blockIdx.x] = lh;
    }
    __syncthreads();
    const uint32_t q_total = s_woff[FT_XITEMS * NW];
    // ---- phase 2, work-list entries only (uniform early exit: with 12-22 % of the slots tied, two or three per thread) ----
    int jl_[FT_XITEMS], start[FT_XITEMS], end[FT_XITEMS];
    uint32_t mine_mask = 0, n_unowned = 0, n_tied = 0;
#pragma unroll
    for (int i = 0; i < FT_XITEMS; ++i) {
        jl_[i] = 0; start[i] = -1; end[i] = -1;
        const uint32_t q = (uint32_t)(i * FT_THREADS + t);
        if ((uint32_t)(i * FT_THREADS) < q_total && q < q_total) {             // (first test: uniform, skips the unused rounds)
        const int jl = s_list[q];
        jl_[i] = jl;
        {
            const int wi = jl >> 6;
            const uint64_t wb = s_head[wi] & ((jl & 63) == 63 ? ~0ull : ((2ull << (jl & 63)) - 1ull));
            start[i] = wb ? wi * 64 + 63 - __builtin_clzll(wb) : (wi ? s_lastH[wi - 1] : -1);    // -1: the run started before this tile
            const uint64_t wa = (jl & 63) == 63 ? 0ull : (s_head[wi] & (~0ull << ((jl & 63) + 1)));
            end[i] = wa ? wi * 64 + __builtin_ctzll(wa) : s_nextH[wi + 1];                        // -1: it goes on beyond the span
        }
        const bool in_main = start[i] >= 0 && start[i] < FT_TILE;              // the group is this workgroup's to handle
        const bool mine = in_main && end[i] > 0 && end[i] - start[i] <= cap;
        if (mine) mine_mask |= 1u << i;
        if (jl < FT_TILE && (in_main || start[i] < 0)) ++n_tied;               // (statistics; counted by the slot's own workgroup)
        if (in_main && !mine && jl < FT_TILE) ++n_unowned;                     // whoever holds a group's first slot owns or reports it
        if (TODO) {
            // every member of a group nobody owns must be flagged by SOME workgroup that sees it:
            //  - the group starts in my slots and I cannot own it: I flag all of it that I see (overhang included);
            //  - it started before my slots: its owner sees at most my first 256 slots, so a member beyond them, or one of a
            //    group that goes on beyond them, is nobody's; one of a group that ends inside them is the earlier workgroup's call
            bool todo;
            if (in_main) todo = !mine;
            else if (start[i] >= FT_TILE) todo = false;
            else todo = jl >= FT_SPAN - FT_TILE || end[i] < 0 || end[i] > FT_SPAN - FT_TILE;
            if (todo) atomicOr(&s_todo[jl >> 5], 1u << (jl & 31));
        }
        }
    }
    uint32_t v[FT_XITEMS];
    Key2T key[FT_XITEMS];
#pragma unroll
    for (int i = 0; i < FT_XITEMS; ++i) {
        v[i] = 0; key[i] = 0;
        if ((mine_mask >> i) & 1u) v[i] = SA[base + jl_[i]];
    }
#pragma unroll
    for (int i = 0; i < FT_XITEMS; ++i) {
        if ((mine_mask >> i) & 1u) {
            key[i] = (Key2T)text_key2<MODE>(T, lcode, P, n, K, v[i], aligned8);
            s_key[jl_[i]] = key[i];
        }
    }
    __syncthreads();
    // ---- rank inside the group = place ----
    int dest[FT_XITEMS];
#pragma unroll
    for (int i = 0; i < FT_XITEMS; ++i) {
        dest[i] = 0;
        if ((mine_mask >> i) & 1u) {
            int rank = 0;
            const Key2T me = key[i];
            const int jl = jl_[i];
            for (int p = start[i]; p < end[i]; ++p) {
                const Key2T k = s_key[p];
                rank += (k < me || (k == me && p < jl)) ? 1 : 0;
            }
            dest[i] = start[i] + rank;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FT_XITEMS; ++i) {
        if ((mine_mask >> i) & 1u) { s_key[dest[i]] = key[i]; s_val[dest[i]] = v[i]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FT_XITEMS; ++i) {
        if (!((mine_mask >> i) & 1u)) continue;
        const int jl = jl_[i];                                                 // now: the POSITION this thread finishes
        const int64_t j = base + jl;
        SA[j] = s_val[jl];
        const Key2T kk = s_key[jl];
        const bool tl = jl > start[i] && s_key[jl - 1] == kk, tr = jl + 1 < end[i] && s_key[jl + 1] == kk;
        if (tl || tr) {                                                        // still tied after this round's key
            int p = jl;
            while (p > start[i] && s_key[p - 1] == kk) --p;
            surv_head[j] = (uint32_t)(base + p);
            atomicOr(&s_surv[jl >> 5], 1u << (jl & 31));
        }
    }
    if (n_unowned) atomicAdd(&s_cnt[0], n_unowned);
    if (n_tied) atomicAdd(&s_cnt[1], n_tied);
    __syncthreads();
    if (t == 0) {
        if (s_cnt[0]) atomicAdd(&counters[1], s_cnt[0]);
        // (statistics, bumped by nearly every workgroup: spread over the words behind the 64 control words -- one word took
        // 8 ns per workgroup, 1.05 of the kernel's 1.7 ms at 256 MiB; the host adds them to counters[2])
        if (s_cnt[1]) atomicAdd(&counters[64 + (blockIdx.x % RR_CHG_COUNTERS) * 32], s_cnt[1]);
    }
    if (t < FT_SPAN / 32) {
        // one atomic per non-empty 32-slot word (base is a multiple of 2048, so words are aligned in the global bitmaps)
        const int64_t j0 = base + 32 * t;
        const uint32_t sw = s_surv[t];
        if (sw) { atomicOr(&surv_bits[j0 >> 5], sw); atomicAdd(&tile_cnt[j0 / RR_TILE], (uint32_t)__popc(sw)); }
        if (TODO) {
            const uint32_t tw = s_todo[t];
            if (tw) {
                const uint32_t fresh = tw & ~atomicOr(&todo_bits[j0 >> 5], tw);      // (a neighbour may have flagged some already)
                if (fresh) atomicAdd(&todo_cnt[j0 / FT_TILE], (uint32_t)__popc(fresh));
            }
        }
    }
}

// Members of the groups k_finish_sorted<.., TODO = true> left to the general path -> (slot, group head, suffix) list in
// slot order.  One workgroup per 2048-slot tile (those without flagged members leave at once); todo_cnt / ft_head
// hold the exclusive sums / running maxima of k_rr_scan.  The group head of a member is the last run start at or
// before it: found in the tile's own run-start bitmap (recomputed from the keys), else carried in.
template <typename KeyT>
__global__ __launch_bounds__(FT_THREADS) void k_todo_compact(const KeyT *__restrict__ skeys, const uint32_t *__restrict__ SA, int64_t n,
                                                              const uint32_t *__restrict__ todo_bits, const uint32_t *__restrict__ todo_cnt,
                                                              const uint32_t *__restrict__ ft_head, const uint32_t *__restrict__ todo_total,
                                                              uint32_t *__restrict__ Uo, uint32_t *__restrict__ Go, uint32_t *__restrict__ Vo)
{
    constexpr int ITEMS = FT_TILE / FT_THREADS;
    constexpr int BW = FT_TILE / 32;                                          // 64 bitmap words of 32 slots
    __shared__ KeyT s_nb[FT_TILE];
    __shared__ uint64_t s_head[FT_TILE / 64];
    __shared__ uint32_t s_bits[BW], s_off[BW];
    const uint32_t here = todo_cnt[blockIdx.x];
    const uint32_t next = (blockIdx.x + 1 < gridDim.x) ? todo_cnt[blockIdx.x + 1] : *todo_total;
    if (next == here) return;
    const uint32_t carry = ft_head[blockIdx.x];                              // (last run start before this tile) + 1
    const int64_t base = (int64_t)blockIdx.x * FT_TILE;
    const int t = threadIdx.x, l = lane_id();
    const int valid_cnt = (int)(n - base < FT_TILE ? n - base : FT_TILE);
    const KeyT prev = base > 0 ? skeys[base - 1] : (KeyT)0;
    KeyT k[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int jl = r * FT_THREADS + t;
        k[r] = jl < valid_cnt ? skeys[base + jl] : (KeyT)0;
        s_nb[jl] = k[r];
    }
    if (t < BW) {
        const int64_t wi = base / 32 + t;
        const uint32_t b = wi < (n + 31) / 32 ? todo_bits[wi] : 0u;
        s_bits[t] = b;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int jl = r * FT_THREADS + t;
        const bool head = jl < valid_cnt && (k[r] != (jl ? s_nb[jl - 1] : prev) || (jl == 0 && base == 0));
        const uint64_t hb = __ballot(head);
        if (l == 0) s_head[jl >> 6] = hb;
    }
    if (t < WAVE) {                                                           // wave 0: exclusive offsets of the 64 words
        const uint32_t c = (uint32_t)__popc(s_bits[t]);
        const uint32_t inc = wave_incl_sum(c);
        s_off[t] = inc - c;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int jl = r * FT_THREADS + t;
        const uint32_t bits = s_bits[jl >> 5];
        if (!((bits >> (jl & 31)) & 1u)) continue;
        int wi = jl >> 6;
        uint64_t wb = s_head[wi] & ((jl & 63) == 63 ? ~0ull : ((2ull << (jl & 63)) - 1ull));
        while (!wb && wi > 0) wb = s_head[--wi];
        const uint32_t g = wb ? (uint32_t)(base + wi * 64 + 63 - __builtin_clzll(wb)) : carry - 1u;
        const uint32_t pos = here + s_off[jl >> 5] + (uint32_t)__popc(bits & ((1u << (jl & 31)) - 1u));
        const uint32_t slot = (uint32_t)(base + jl);
        Uo[pos] = slot; Go[pos] = g; Vo[pos] = SA[slot];
    }
}

// survivors of k_finish_sorted -> (slot, group head, suffix) lists in slot order; tile_cnt holds the exclusive
// offsets of the re-rank tiles (k_rr_scan), one workgroup per tile, one bitmap word per thread
__global__ __launch_bounds__(256) void k_surv_compact(const uint32_t *__restrict__ surv_bits, const uint32_t *__restrict__ surv_head,
                                                       const uint32_t *__restrict__ SA, int64_t n, const uint32_t *__restrict__ tile_cnt,
                                                       const uint32_t *__restrict__ tile_total, uint32_t *__restrict__ Uo,
                                                       uint32_t *__restrict__ Go, uint32_t *__restrict__ Vo)
{
    static_assert(RR_TILE == 256 * 32, "one bitmap word per thread");
    __shared__ uint32_t lds[256 / WAVE + 1];
    const uint32_t here = tile_cnt[blockIdx.x];
    const uint32_t next = (blockIdx.x + 1 < gridDim.x) ? tile_cnt[blockIdx.x + 1] : *tile_total;
    if (next == here) return;
    const int64_t widx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nwords = (n + 31) / 32;
    uint32_t bits = widx < nwords ? surv_bits[widx] : 0u;
    uint32_t tot;
    uint32_t off = here + block_excl_sum<256>((uint32_t)__popc(bits), lds, &tot);
    while (bits) {
        const int b = __builtin_ctz(bits);
        bits &= bits - 1;
        const uint32_t slot = (uint32_t)(widx * 32 + b);
        Uo[off] = slot; Go[off] = surv_head[slot]; Vo[off] = SA[slot];
        ++off;
    }
}

// ---- ordered compaction of the flagged elements (those the local pass could not own) ----
// Every flagged group is flagged as a whole, head included.  Besides the flagged members per tile, the flagged group HEADS
// are counted: the global sort then keys the members by (index of their group among the flagged groups, secondary key)
// instead of (28-bit slot of the group head, secondary key) -- about 47 instead of 58 key bits on C3: two radix passes less.
__global__ __launch_bounds__(RR_THREADS) void k_flag_count(const uint8_t *__restrict__ flag, const uint32_t *__restrict__ U,
                                                            const uint32_t *__restrict__ G, int64_t m, uint32_t *__restrict__ tile_cnt,
                                                            uint32_t *__restrict__ tile_heads)
{
    constexpr int NW = RR_THREADS / WAVE;
    __shared__ uint32_t wc[NW], wh[NW];
    const int l = lane_id(), w = wave_id();
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)w * RR_WAVE_ELEMS;     // wave-striped: 64 consecutive elements per load
    uint32_t c = 0, h = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        const bool f = i < m && (!flag || flag[i] != 0);       // flag == nullptr: every member (the whole list goes through the global sort)
        const bool hd = f && U[i] == G[i];
        c += (uint32_t)__popcll(__ballot(f));
        h += (uint32_t)__popcll(__ballot(hd));
    }
    if (l == 0) { wc[w] = c; wh[w] = h; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tc = 0, th = 0;
        for (int i = 0; i < NW; ++i) { tc += wc[i]; th += wh[i]; }
        tile_cnt[blockIdx.x] = tc; tile_heads[blockIdx.x] = th;
    }
}

// bk = (group index among the flagged groups << kb) | secondary key
__global__ __launch_bounds__(RR_THREADS) void k_flag_gather(const uint8_t *__restrict__ flag, const uint64_t *__restrict__ keys,
                                                             const uint32_t *__restrict__ V, const uint32_t *__restrict__ U,
                                                             const uint32_t *__restrict__ G, int64_t m, const uint32_t *__restrict__ tile_cnt,
                                                             const uint32_t *__restrict__ tile_heads, int kb,
                                                             uint64_t *__restrict__ bk, uint32_t *__restrict__ bv, uint32_t *__restrict__ bidx)
{
    constexpr int NW = RR_THREADS / WAVE;
    __shared__ uint32_t wc[NW], wh[NW];
    const int l = lane_id(), w = wave_id();
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)w * RR_WAVE_ELEMS;
    uint64_t fm[RR_ITEMS], hm[RR_ITEMS];
    uint32_t c = 0, h = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        const bool f = i < m && flag[i] != 0;
        const bool hd = f && U[i] == G[i];
        fm[r] = __ballot(f); hm[r] = __ballot(hd);
        c += (uint32_t)__popcll(fm[r]); h += (uint32_t)__popcll(hm[r]);
    }
    if (l == 0) { wc[w] = c; wh[w] = h; }
    __syncthreads();
    uint32_t off = tile_cnt[blockIdx.x], heads = tile_heads[blockIdx.x];
    for (int ww = 0; ww < w; ++ww) { off += wc[ww]; heads += wh[ww]; }
    const uint64_t kmask = kb >= 64 ? ~0ull : ((1ull << kb) - 1ull);
    const uint64_t lt_mask = (1ull << l) - 1ull, le_mask = (l == 63) ? ~0ull : ((2ull << l) - 1ull);
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        if ((fm[r] >> l) & 1ull) {
            const uint32_t o = off + (uint32_t)__popcll(fm[r] & lt_mask);
            const uint32_t gi = heads + (uint32_t)__popcll(hm[r] & le_mask) - 1u;       // flagged heads at or before i, minus one
            bk[o] = ((uint64_t)gi << kb) | (keys[i] & kmask);
            bv[o] = V[i]; bidx[o] = (uint32_t)i;
        }
        off += (uint32_t)__popcll(fm[r]);
        heads += (uint32_t)__popcll(hm[r]);
    }
}

// The whole list through the global sort (no tile could own most of it: runs, periodic texts): the keys are re-written in place
// as (index of the group in the list << kb) | secondary key -- tile_heads holds the exclusive head counts of k_flag_count /
// k_rr_scan with flag == nullptr.  The group-head slots are not put back after the sort (the re-rank kernels only compare
// neighbouring keys).  A text of 1000 repeated blocks has 1000 groups: 10 + 30 key bits, five radix passes instead of eight.
__global__ __launch_bounds__(RR_THREADS) void k_rekey_dense(uint64_t *__restrict__ keys, const uint32_t *__restrict__ U,
                                                             const uint32_t *__restrict__ G, int64_t m,
                                                             const uint32_t *__restrict__ tile_heads, int kb)
{
    constexpr int NW = RR_THREADS / WAVE;
    __shared__ uint32_t wh[NW];
    const int l = lane_id(), w = wave_id();
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)w * RR_WAVE_ELEMS;
    uint64_t hm[RR_ITEMS];
    uint32_t h = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        hm[r] = __ballot(i < m && U[i] == G[i]);
        h += (uint32_t)__popcll(hm[r]);
    }
    if (l == 0) wh[w] = h;
    __syncthreads();
    uint32_t heads = tile_heads[blockIdx.x];
    for (int ww = 0; ww < w; ++ww) heads += wh[ww];
    const uint64_t kmask = kb >= 64 ? ~0ull : ((1ull << kb) - 1ull);
    const uint64_t le_mask = (l == 63) ? ~0ull : ((2ull << l) - 1ull);
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        if (i < m) {
            const uint32_t gi = heads + (uint32_t)__popcll(hm[r] & le_mask) - 1u;
            keys[i] = ((uint64_t)gi << kb) | (keys[i] & kmask);
        }
        heads += (uint32_t)__popcll(hm[r]);
    }
}

// Gather of the secondary keys for a list that goes through the global sort as a whole, tile-structured (eight look-ups per
// thread in flight) and keyed directly: tile_heads == nullptr: (group-head slot << kb) | key2, otherwise (index of the group
// in the list << kb) | key2 as k_rekey_dense would make it.  Nothing restores the head slots after the sort: the re-rank
// kernels only compare neighbouring keys.
template <int MODE>
__global__ __launch_bounds__(RR_THREADS) void k_gather_keyed(const uint32_t *__restrict__ V, const uint32_t *__restrict__ U,
                                                              const uint32_t *__restrict__ G, const uint8_t *__restrict__ T, KeyParams P,
                                                              int64_t m, int64_t n, KeySrc K, const uint32_t *__restrict__ tile_heads,
                                                              uint64_t *__restrict__ keys, uint32_t *__restrict__ starts, uint32_t groups)
{
    constexpr int NW = RR_THREADS / WAVE;
    __shared__ uint32_t wh[NW];
    __shared__ uint8_t lcode[256];
    if (threadIdx.x < 256) lcode[threadIdx.x] = P.code[threadIdx.x];
    const bool aligned8 = (((uintptr_t)T) & 7) == 0;
    const int l = lane_id(), w = wave_id();
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)w * RR_WAVE_ELEMS;
    uint32_t v[RR_ITEMS], g[RR_ITEMS];
    uint64_t hm[RR_ITEMS];
    uint32_t h = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        v[r] = i < m ? V[i] : 0u;
        g[r] = i < m ? G[i] : 0u;
        hm[r] = 0;
        if (tile_heads) {                                       // (uniform)
            hm[r] = __ballot(i < m && U[i] == g[r]);
            h += (uint32_t)__popcll(hm[r]);
        }
    }
    if (l == 0) wh[w] = h;
    __syncthreads();
    uint64_t k2[RR_ITEMS];
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        k2[r] = i < m ? text_key2<MODE>(T, lcode, P, n, K, v[r], aligned8) : 0ull;
    }
    uint32_t heads = tile_heads ? tile_heads[blockIdx.x] : 0u;
    if (tile_heads) for (int ww = 0; ww < w; ++ww) heads += wh[ww];
    const uint64_t le_mask = (l == 63) ? ~0ull : ((2ull << l) - 1ull);
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        const uint32_t hi = tile_heads ? heads + (uint32_t)__popcll(hm[r] & le_mask) - 1u : g[r];
        if (i < m) keys[i] = ((uint64_t)hi << K.kb) | k2[r];
        if (starts && ((hm[r] >> l) & 1ull)) starts[hi] = (uint32_t)i;          // (what k_group_starts would write: the three-way split's table)
        heads += (uint32_t)__popcll(hm[r]);
    }
    if (starts && blockIdx.x == 0 && threadIdx.x == 0) starts[groups] = (uint32_t)m;
}

// ------------------------------------------------------------------------------------------
// Three-way split of giant groups around their majority key.  A run, a periodic text or a long repeat keeps hundreds of
// millions of suffixes in a handful of groups round after round, and in every round all but a few members of a group carry
// the SAME secondary key (the rank of the one group their look-ups land in): a radix sort of the whole list moves 268 M pairs
// five times to pull out the few thousand that differ.  Instead: pivot(g) = key of the group's middle member; the members
// whose key differs (the minority) are extracted and sorted on their own; the others keep their relative order and only
// shift by the number of minority members that sort below the pivot:
//      [ minority < pivot, sorted | members with the pivot key, in list order | minority > pivot, sorted ]
// The keys carry the dense group index above bit kb (k_gather_keyed / k_rekey_dense).  Streaming passes over the list:
// count, extract, place -- against one histogram and one tile scatter per radix digit.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RR_THREADS) void k_group_starts(const uint32_t *__restrict__ U, const uint32_t *__restrict__ G, int64_t m,
                                                              const uint32_t *__restrict__ tile_heads, uint32_t groups,
                                                              uint32_t *__restrict__ starts)
{
    constexpr int NW = RR_THREADS / WAVE;
    __shared__ uint32_t wh[NW];
    const int l = lane_id(), w = wave_id();
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)w * RR_WAVE_ELEMS;
    uint64_t hm[RR_ITEMS];
    uint32_t h = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        hm[r] = __ballot(i < m && U[i] == G[i]);
        h += (uint32_t)__popcll(hm[r]);
    }
    if (l == 0) wh[w] = h;
    __syncthreads();
    uint32_t heads = tile_heads[blockIdx.x];
    for (int ww = 0; ww < w; ++ww) heads += wh[ww];
    const uint64_t lt_mask = (1ull << l) - 1ull;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        if ((hm[r] >> l) & 1ull) starts[heads + (uint32_t)__popcll(hm[r] & lt_mask)] = (uint32_t)(wbase + 64 * r + l);
        heads += (uint32_t)__popcll(hm[r]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) starts[groups] = (uint32_t)m;
}

__global__ __launch_bounds__(256) void k_split_pivots(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ starts, uint32_t groups,
                                                       int kb, uint64_t *__restrict__ pivot)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint64_t kmask = kb >= 64 ? ~0ull : ((1ull << kb) - 1ull);
    if (g < groups) pivot[g] = keys[((uint64_t)starts[g] + starts[g + 1]) >> 1] & kmask;
}

// minority masks of one wave's 512 elements (wave-striped like the re-rank kernels)
__device__ __forceinline__ void split_wave_masks(const uint64_t *__restrict__ keys, int64_t m, int64_t wbase, int kb,
                                                 const uint64_t *__restrict__ pivot, uint64_t *k, uint64_t *mn)
{
    const int l = lane_id();
    const uint64_t kmask = kb >= 64 ? ~0ull : ((1ull << kb) - 1ull);
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        k[r] = i < m ? keys[i] : 0ull;
        mn[r] = __ballot(i < m && (k[r] & kmask) != pivot[k[r] >> kb]);
    }
}

__global__ __launch_bounds__(RR_THREADS) void k_split_count(const uint64_t *__restrict__ keys, int64_t m, int kb,
                                                             const uint64_t *__restrict__ pivot, uint32_t *__restrict__ tile_cnt)
{
    constexpr int NW = RR_THREADS / WAVE;
    __shared__ uint32_t wc[NW];
    const int w = wave_id();
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)w * RR_WAVE_ELEMS;
    uint64_t k[RR_ITEMS], mn[RR_ITEMS];
    split_wave_masks(keys, m, wbase, kb, pivot, k, mn);
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) c += (uint32_t)__popcll(mn[r]);
    if (lane_id() == 0) wc[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int i = 0; i < NW; ++i) t += wc[i];
        tile_cnt[blockIdx.x] = t;
    }
}

// PLACE = false: the minority members -> (mk, mv) in list order; mps[g] = minority members in front of group g's first member
// (mps[groups] = their total).  PLACE = true: the members with the pivot key -> their final places in (outk, outv).
template <bool PLACE>
__global__ __launch_bounds__(RR_THREADS) void k_split_pass(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ V, int64_t m, int kb,
                                                            const uint64_t *__restrict__ pivot, const uint32_t *__restrict__ tile_cnt,
                                                            const uint32_t *__restrict__ starts, uint32_t groups, uint32_t *__restrict__ mps,
                                                            const uint32_t *__restrict__ total, uint64_t *__restrict__ mk, uint32_t *__restrict__ mv,
                                                            const uint32_t *__restrict__ L, uint64_t *__restrict__ outk, uint32_t *__restrict__ outv)
{
    constexpr int NW = RR_THREADS / WAVE;
    __shared__ uint32_t wc[NW];
    const int l = lane_id(), w = wave_id();
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)w * RR_WAVE_ELEMS;
    uint64_t k[RR_ITEMS], mn[RR_ITEMS];
    split_wave_masks(keys, m, wbase, kb, pivot, k, mn);
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) c += (uint32_t)__popcll(mn[r]);
    if (l == 0) wc[w] = c;
    __syncthreads();
    uint32_t off = tile_cnt[blockIdx.x];                     // minority members in front of this wave's first element
    for (int ww = 0; ww < w; ++ww) off += wc[ww];
    const uint64_t lt_mask = (1ull << l) - 1ull;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        if (i < m) {
            const uint32_t before = off + (uint32_t)__popcll(mn[r] & lt_mask);
            const uint32_t g = (uint32_t)(k[r] >> kb);
            const bool minor = (mn[r] >> l) & 1ull;
            if (!PLACE) {
                if (minor) { mk[before] = k[r]; mv[before] = V[i]; }
                if ((uint32_t)i == starts[g]) mps[g] = before;
            } else if (!minor) {
                // members with the pivot key in front of me in my group = (i - start) - (minority members between); L[g] go below
                const uint32_t pos = (uint32_t)i + L[g] - (before - mps[g]);
                outk[pos] = k[r]; outv[pos] = V[i];
            }
        }
        off += (uint32_t)__popcll(mn[r]);
    }
    if (!PLACE && blockIdx.x == 0 && threadIdx.x == 0) mps[groups] = *total;
}

// L[g] = minority members of group g below its pivot: a binary search in the group's range of the SORTED minority keys
__global__ __launch_bounds__(256) void k_split_less(const uint64_t *__restrict__ mk, const uint32_t *__restrict__ mps,
                                                     const uint64_t *__restrict__ pivot, uint32_t groups, int kb, uint32_t *__restrict__ L)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= groups) return;
    const uint64_t want = ((uint64_t)g << kb) | pivot[g];
    uint32_t lo = mps[g], hi = mps[g + 1];
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (mk[mid] < want) lo = mid + 1; else hi = mid; }
    L[g] = lo - mps[g];
}

__global__ __launch_bounds__(256) void k_split_place_minor(const uint64_t *__restrict__ mk, const uint32_t *__restrict__ mv, int64_t count, int kb,
                                                            const uint32_t *__restrict__ starts, const uint32_t *__restrict__ mps,
                                                            const uint32_t *__restrict__ L, uint64_t *__restrict__ outk, uint32_t *__restrict__ outv)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= count) return;
    const uint64_t key = mk[q];
    const uint32_t g = (uint32_t)(key >> kb);
    const uint32_t t = (uint32_t)q - mps[g], cnt = mps[g + 1] - mps[g];
    const uint32_t pos = t < L[g] ? starts[g] + t : starts[g + 1] - (cnt - t);
    outk[pos] = key; outv[pos] = mv[q];
}

// sorted flagged elements back to their list positions (sorted by group first, and bidx is increasing,
// so the o-th sorted element belongs at the o-th flagged position); the key gets its group-head slot back
// (a group keeps its list positions, so position j still belongs to the group of G[j])
__global__ __launch_bounds__(256) void k_scatter_back(const uint64_t *__restrict__ bk, const uint32_t *__restrict__ bv,
                                                       const uint32_t *__restrict__ bidx, const uint32_t *__restrict__ G, int kb,
                                                       int64_t count, uint64_t *__restrict__ keys, uint32_t *__restrict__ V)
{
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t kmask = kb >= 64 ? ~0ull : ((1ull << kb) - 1ull);
    if (o < count) { const uint32_t j = bidx[o]; keys[j] = ((uint64_t)G[j] << kb) | (bk[o] & kmask); V[j] = bv[o]; }
}

}  // namespace sa
