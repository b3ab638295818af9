// kernels/bucket_sort.hpp -- the low 16 bits of the 32-bit first stage, ordered bucket by bucket in LDS.
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
// Replaces two of the four global tile-scatter passes of the 32-bit initial sort, i.e. part of the arithmetic behind
// `cdivsufsort::sort_in_place` (reference src/saca.rs:14).
//
// A stable LSD sort may stop early: after the passes over the key bits 16..31 the pairs are grouped by their top 16 bits
// (a BUCKET, contiguous; the order of its members among themselves is UNSPECIFIED -- the text-keyed first pass hands a lane
// consecutive positions and is not stable), and what the two passes over the bits 0..15 would have done --
// 2 x 16 bytes per pair through a 256-way scatter -- is a sort INSIDE every bucket.  With n = 2^28 suffixes of random
// bytes a bucket holds 4096 +- 64 pairs: it fits in LDS, so one workgroup per bucket reads its pairs once (coalesced),
// orders them by two counting passes in LDS and writes them back in order (coalesced, sequential): 16 bytes per pair at
// streaming speed instead of 32 at scatter speed.  The first pass (the low 10 of the 16 bits) takes its places from one LDS
// counter per digit (one atomic per pair; the order among equal digits is whatever the atomics give, which is enough: the
// second pass is stable and pairs equal in ALL low bits are equal keys); the second (the top 6 bits) is the stable ranking of
// the tile scatter (6 ballots + mbcnt per key, per-wave digit counts in LDS).  The kernel is instruction-bound, not
// memory-bound -- with two 8-bit ballot passes it took 1.40 ms for 2^28 pairs.
//   k_bucket_starts  start[b] = first pair whose key >> lbits is >= b, by binary search in the grouped keys
//                    (65 537 searches; the first levels are shared and cache-resident), and the largest bucket
//   k_bucket_sort    one workgroup per bucket; THREADS x ITEMS = the largest bucket the shape can hold; the host reads
//                    the largest bucket back and picks the shape -- a text whose buckets are too large for every shape
//                    (a skewed alphabet that the entropy probe nevertheless sent to the 32-bit stage) takes the four
//                    global passes as before
// Only the low 16 key bits are staged (the top 16 are the bucket's number): 6 bytes of LDS per pair.
// FINISH = true: the round that k_finish_sorted (kernels/refine.hpp) runs over the sorted keys happens here, while the bucket
// is still in LDS -- suffixes tied on all 32 bits (a run of equal staged keys; never across buckets) fetch their low key bits
// from the text, are ranked inside their group and change places; the ones that are still tied are recorded for the later
// rounds exactly as k_finish_sorted records them (bitmap, slot of the subgroup's first member, count per re-rank tile).  The
// low keys travel through the values' LDS words, so the stage stays at 6 bytes per pair.
// Algorithmic traffic: 8 B read + 8 B written per pair (+ the text look-ups of the tied suffixes).
#pragma once
#include "common.hpp"
#include "radix_sort.hpp"
#include "rerank.hpp"
#include "refine.hpp"

namespace sa {

constexpr int BK_MAX_LBITS = 16;             // low key bits ordered inside a bucket, at most: a first pass on up to 10 of them, ...
constexpr int BK_BBITS = 6;                  // ... and a stable one on the top 6 (NB_B = 64 = one wave: lane d owns digit d)
constexpr int BK_STARTS_THREADS = 256;

// start[0 .. nb]: first index whose key >> lbits is >= b (start[nb] = n: nb is past the last bucket in use)
__global__ __launch_bounds__(BK_STARTS_THREADS) void k_bucket_starts(const uint32_t *__restrict__ keys, int64_t n, int lbits, uint32_t nb,
                                                                     uint32_t *__restrict__ start)
{
    const uint32_t b = blockIdx.x * BK_STARTS_THREADS + threadIdx.x;
    if (b > nb) return;
    int64_t lo = 0, hi = n;                    // first index in [0, n] with (keys[i] >> lbits) >= b
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((keys[mid] >> lbits) >= b) hi = mid; else lo = mid + 1;
    }
    start[b] = (uint32_t)lo;
}

// words[0] = the largest bucket (zeroed by the host)
__global__ __launch_bounds__(BK_STARTS_THREADS) void k_bucket_max(const uint32_t *__restrict__ start, uint32_t nb, uint32_t *__restrict__ words)
{
    __shared__ uint32_t lds[BK_STARTS_THREADS / WAVE];
    const uint32_t b = blockIdx.x * BK_STARTS_THREADS + threadIdx.x;
    const uint32_t sz = b < nb ? start[b + 1] - start[b] : 0u;
    uint32_t tot;
    (void)block_incl_max<BK_STARTS_THREADS>(sz, lds, &tot);
    if (threadIdx.x == 0 && tot) atomicMax(&words[0], tot);
}

// Before any sorting: how large would the largest bucket be?  The entropy probe's samples (top 32 key bits in the high word,
// kernels/keys.hpp k_sample_keys) are counted per bucket; the host scales the largest count by n / samples.  An estimate -- a
// bucket of 4096 suffixes holds 16 of 2^20 samples -- that only has to tell a skewed alphabet (one bucket with a tenth of
// the text) from a flat one, so that the former does not pay for two passes that the exact check afterwards throws away.
__global__ __launch_bounds__(BK_STARTS_THREADS) void k_sample_bucket_hist(const uint64_t *__restrict__ samples, int64_t count, int top_bits,
                                                                          uint32_t *__restrict__ hist)
{
    const int64_t i = (int64_t)blockIdx.x * BK_STARTS_THREADS + threadIdx.x;
    if (i < count) atomicAdd(&hist[samples[i] >> (64 - top_bits)], 1u);
}

__global__ __launch_bounds__(BK_STARTS_THREADS) void k_u32_max(const uint32_t *__restrict__ a, uint32_t count, uint32_t *__restrict__ out)
{
    __shared__ uint32_t lds[BK_STARTS_THREADS / WAVE];
    const uint32_t i = blockIdx.x * BK_STARTS_THREADS + threadIdx.x;
    uint32_t tot;
    (void)block_incl_max<BK_STARTS_THREADS>(i < count ? a[i] : 0u, lds, &tot);
    if (threadIdx.x == 0 && tot) atomicMax(out, tot);
}

// One workgroup per bucket.  Element e of the bucket is held by wave e / (64 J), item (e / 64) % J, lane e % 64 with
// J = ceil(size / THREADS) items in use (wave-striped, so a stable rank inside the wave is a prefix count over lanes and items).
// what the fused finish needs (FINISH = true): the text and its key geometry come as separate kernel arguments
struct BucketFinish {
    const uint8_t *T;
    int64_t n;
    int cap;                        // largest group ordered here (a larger one only bumps counters[1]: the host runs the general path)
    uint32_t *surv_bits, *surv_head, *tile_cnt, *counters;    // as k_finish_sorted's
};

template <int THREADS, int ITEMS, int MINW = 1, bool FINISH = false>
__global__ __launch_bounds__(THREADS, MINW) void k_bucket_sort(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                         const uint32_t *__restrict__ start, int lbits,
                                                         uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                         uint32_t *__restrict__ err, BucketFinish F, KeyParams P, KeySrc K, int xbits)
{
    constexpr int CAP = THREADS * ITEMS;
    constexpr int NWAVES = THREADS / WAVE;
    constexpr int NB_A = 1 << (BK_MAX_LBITS - BK_BBITS);        // bins of the first pass at most
    constexpr int NB_B = 1 << BK_BBITS;
    static_assert(CAP < 65536, "16-bit bucket-local positions");
    static_assert(ITEMS % 2 == 0, "positions are packed two to a register");
    static_assert(NB_A % THREADS == 0 || THREADS % NB_A == 0, "the bins of the first pass are scanned by the whole workgroup");
    static_assert(NB_B == WAVE, "lane d of wave 0 owns digit d of the second pass");
    // work list of the tied slots (FINISH): in the first pass's counters (free once the bucket is ordered) when a thread stages at
    // most 10 pairs -- room for 40 % of the bucket's slots --, else (20 pairs per thread) in an array of its own for half of them
    constexpr int R_MAX = ITEMS <= 10 ? 4 : 10;
    constexpr int LCAP = ITEMS <= 10 ? 4 * THREADS : CAP / 2;
    static_assert(LCAP <= R_MAX * THREADS, "every list entry has a thread and a round");
    constexpr int CNT_WORDS = (FINISH && ITEMS <= 10 && LCAP / 2 > NB_A) ? LCAP / 2 : NB_A;
    __shared__ uint16_t lds_k[CAP];
    __shared__ uint32_t lds_v[CAP];
    __shared__ uint32_t cnt_a[CNT_WORDS];          // first pass: elements per digit, then where the digit's run starts
    __shared__ uint16_t wave_hist[NWAVES][NB_B];   // second pass: per-wave digit counts, then offsets
    __shared__ uint32_t digit_base[NB_B];
    __shared__ uint32_t scan_lds[NWAVES + 1];
    constexpr int SURV_WORDS = FINISH ? CAP / 32 + 2 : 1;
    __shared__ uint8_t lcode[FINISH ? 256 : 1];
    __shared__ uint32_t s_surv[SURV_WORDS];        // survivor bits of this bucket's slots, word-aligned to the global bitmap
    __shared__ uint32_t s_cnt[2];
    __shared__ uint16_t s_list_own[FINISH && ITEMS > 10 ? LCAP : 1];
    uint16_t *s_list = ITEMS <= 10 ? (uint16_t *)cnt_a : s_list_own;

    const int tid = threadIdx.x, l = lane_id(), w = wave_id();
    if (FINISH) {
        if (tid < 256) lcode[tid] = P.code[tid];
        for (int i = tid; i < SURV_WORDS; i += THREADS) s_surv[i] = 0;
        if (tid < 2) s_cnt[tid] = 0;
    }
    const uint32_t b = blockIdx.x;
    const uint32_t lo = start[b];
    const int size = (int)(start[b + 1] - lo);
    if (size <= 0) return;
    if (size > CAP) { if (tid == 0) atomicAdd(err, 1u); return; }      // (the host picked the shape from the largest bucket: cannot happen)
    const int J = (size + THREADS - 1) / THREADS;
    const int e0 = w * J * WAVE + l;
    uint16_t *my_hist = wave_hist[w];

    uint32_t key[ITEMS], val[ITEMS], pp[ITEMS / 2];
#define BK_POS(j) ((pp[(j) >> 1] >> (16 * ((j) & 1))) & 0xffffu)
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        // (register arrays are only ever assigned unconditionally, from scalars: conditional element stores make the compiler keep
        // the arrays as 32-wide vectors that it copies and spills around every branch)
        const int e = e0 + j * WAVE;
        uint32_t kx = 0, vx = 0;
        if (j < J && e < size) { kx = keys_in[lo + e]; vx = vals_in[lo + e]; }
        // the key from here on: the low key bits, and below them the xbits further key bits that came in the top of the value
        key[j] = ((kx & ((1u << lbits) - 1u)) << xbits) | (uint32_t)((uint64_t)vx >> (32 - xbits));
        val[j] = (uint32_t)(((uint64_t)vx << (32 + xbits)) >> (32 + xbits));
    }
    const uint32_t hi_bits = (keys_in[lo] >> lbits) << lbits;      // the bucket's number, as key bits (1 <= lbits, lbits + xbits <= 16, host-checked)
    const int sbits = lbits + xbits;                               // key bits ordered here
    const uint32_t lo_mask = (1u << sbits) - 1u;
    const int bbits = sbits < BK_BBITS ? sbits : BK_BBITS;        // second pass: the top bits of the staged key, stable
    const int abits = sbits - bbits;                               // first pass: the bits below them

    if (abits > 0) {
        // ---- first pass: a counter per digit hands out the places.  Which of two pairs with the same digit comes first is left
        // to the order the atomics arrive in: the second pass keeps whatever order this one leaves (it is stable), and pairs
        // that agree in all low bits have the same 32-bit key -- their order is the later rounds' business, not the sort's ----
        const uint32_t amask = (1u << abits) - 1u;
        for (int i = tid; i < NB_A; i += THREADS) cnt_a[i] = 0;
        __syncthreads();                           // (also: the keys have arrived)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            uint32_t r = 0;
            if (j < J && (e0 + j * WAVE) < size) r = atomicAdd(&cnt_a[key[j] & amask], 1u);
            if ((j & 1) == 0) pp[j >> 1] = r; else pp[j >> 1] |= r << 16;
        }
        lds_barrier();
        {
            // exclusive sums over the NB_A counters, in place
            constexpr int BPT = NB_A >= THREADS ? NB_A / THREADS : 1;
            const bool scans = tid * BPT < NB_A;
            uint32_t c[BPT], sum = 0;
#pragma unroll
            for (int i = 0; i < BPT; ++i) { c[i] = scans ? cnt_a[tid * BPT + i] : 0u; sum += c[i]; }
            uint32_t all;
            uint32_t run = block_excl_sum_b<THREADS, true>(sum, scan_lds, &all);
#pragma unroll
            for (int i = 0; i < BPT; ++i) { if (scans) cnt_a[tid * BPT + i] = run; run += c[i]; }
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (j < J && (e0 + j * WAVE) < size) {
                const uint32_t ps = BK_POS(j) + cnt_a[key[j] & amask];
                lds_k[ps] = (uint16_t)(key[j] & lo_mask);
                lds_v[ps] = val[j];
            }
        }
        lds_barrier();
        // the order of the first pass: back into registers, wave-striped
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int e = e0 + j * WAVE;
            uint32_t kx = 0, vx = 0;
            if (j < J && e < size) { kx = lds_k[e]; vx = lds_v[e]; }
            key[j] = kx; val[j] = vx;
        }
    }
    {
        // ---- second pass, stable: rank inside the wave (lanes with my digit below me + earlier items: ballots + mbcnt), per-wave
        // digit counts in LDS ----
        const uint32_t dmask = (1u << bbits) - 1u;
        for (int i = tid; i < NWAVES * NB_B / 2; i += THREADS) ((uint32_t *)&wave_hist[0][0])[i] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            uint32_t r = 0;
            if (j < J) {                           // (uniform)
                const bool ok = (e0 + j * WAVE) < size;
                const uint32_t d = (key[j] >> abits) & dmask;
                const uint64_t okm = __ballot(ok);
                uint32_t xlo = ~(uint32_t)okm, xhi = ~(uint32_t)(okm >> 32);
#pragma unroll
                for (int bb = 0; bb < BK_BBITS; ++bb) {
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
        // ---- wave 0, lane d: per-wave offsets of digit d, its start in the bucket ----
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
        // ---- into LDS in the order of this digit (stable) ----
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (j < J && (e0 + j * WAVE) < size) {
                const uint32_t d = (key[j] >> abits) & dmask;
                const uint32_t ps = BK_POS(j) + digit_base[d] + my_hist[d];
                lds_k[ps] = (uint16_t)(key[j] & lo_mask);
                lds_v[ps] = val[j];
            }
        }
    }
#undef BK_POS
    __syncthreads();
    if (FINISH) {
        // ---- suffixes tied on all 32 key bits: ordered by their low key bits, inside the staged bucket ----
        // Work list of the tied slots (any order), then R rounds of THREADS entries; the thread that takes an entry keeps it through
        // all phases.  Texts the probe sends here have few ties (random bytes at n = 2^28: 6 % of the slots; DNA at 2^30: 22 %), so a
        // thread holds one to five entries and that many text look-ups are exposed, not one per slot it staged.
        const bool aligned8 = (((uintptr_t)F.T) & 7) == 0;
        const int cap = F.cap;
        uint32_t n_unowned = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int idx = tid + j * THREADS;
            bool tied = false;
            if (j < J && idx < size) {
                const uint16_t k = lds_k[idx];
                tied = (idx > 0 && lds_k[idx - 1] == k) || (idx + 1 < size && lds_k[idx + 1] == k);
            }
            if (j < J) {                           // (uniform)
                const uint64_t m = __ballot(tied);
                if (m) {
                    uint32_t base = 0;
                    if (l == 0) base = atomicAdd(&s_cnt[1], (uint32_t)__popcll(m));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (tied && slot < (uint32_t)LCAP) s_list[slot] = (uint16_t)idx;
                }
            }
        }
        lds_barrier();
        // No room in the list (far more ties than this route is meant for): nothing in this bucket moves -- a group with
        // members in the list and members outside it must not be ordered by the ones inside -- and the whole bucket is reported.
        const int n_tied = (int)s_cnt[1];
        const int n_list = n_tied <= LCAP ? n_tied : 0;
        if (n_tied > LCAP && tid == 0) n_unowned += (uint32_t)n_tied;
        uint32_t fi[R_MAX], fv[R_MAX], fk[R_MAX], fse[R_MAX], fd[R_MAX];
        uint32_t own_mask = 0;
#pragma unroll
        for (int r = 0; r < R_MAX; ++r) {
            const int q = tid + r * THREADS;
            uint32_t ii = 0, se = 0, vv = 0, own = 0;
            if (q < n_list) {
                const int idx = (int)s_list[q];
                const uint16_t k = lds_k[idx];
                int sb = idx, eb = idx + 1;
                while (sb > 0 && lds_k[sb - 1] == k && idx - sb <= cap) --sb;
                while (eb < size && lds_k[eb] == k && eb - sb <= cap) ++eb;
                if (eb - sb <= cap && !(sb > 0 && lds_k[sb - 1] == k)) {
                    own = 1u;
                    ii = (uint32_t)idx;
                    se = (uint32_t)sb | ((uint32_t)eb << 16);
                    vv = lds_v[idx];
                } else ++n_unowned;                // too large for this round: reported, left as it is (order among equal keys: unspecified)
            }
            fi[r] = ii; fse[r] = se; fv[r] = vv;
            own_mask |= own << r;
        }
#pragma unroll
        for (int r = 0; r < R_MAX; ++r) {
            uint32_t k2 = 0;
            if (r * THREADS < n_list) k2 = (uint32_t)text_key2<KS_LOWKEY>(F.T, lcode, P, F.n, K, fv[r], aligned8);      // (uniform branch)
            fk[r] = k2;
        }
        lds_barrier();                             // every value of a listed slot is in a register
#pragma unroll
        for (int r = 0; r < R_MAX; ++r)
            if ((own_mask >> r) & 1u) lds_v[fi[r]] = fk[r];
        lds_barrier();
#pragma unroll
        for (int r = 0; r < R_MAX; ++r) {
            uint32_t d = 0;
            if ((own_mask >> r) & 1u) {
                const int idx = (int)fi[r], sb = (int)(fse[r] & 0xffffu), eb = (int)(fse[r] >> 16);
                const uint32_t me = fk[r];
                int rank = 0;
                for (int q = sb; q < eb; ++q) {
                    const uint32_t kq = lds_v[q];
                    rank += (kq < me || (kq == me && q < idx)) ? 1 : 0;
                }
                d = (uint32_t)(sb + rank);
            }
            fd[r] = d;
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < R_MAX; ++r)
            if ((own_mask >> r) & 1u) lds_v[fd[r]] = fk[r];
        lds_barrier();
        const uint32_t lsh = lo & 31u;             // bit 0 of s_surv[0] is global slot lo - lsh
#pragma unroll
        for (int r = 0; r < R_MAX; ++r) {
            if ((own_mask >> r) & 1u) {
                const int sb = (int)(fse[r] & 0xffffu), eb = (int)(fse[r] >> 16), d = (int)fd[r];
                const uint32_t kk = fk[r];
                const bool tl = d > sb && lds_v[d - 1] == kk, tr = d + 1 < eb && lds_v[d + 1] == kk;
                if (tl || tr) {                    // still tied on the whole 64-bit key
                    int q = d;
                    while (q > sb && lds_v[q - 1] == kk) --q;
                    F.surv_head[(int64_t)lo + d] = lo + (uint32_t)q;
                    atomicOr(&s_surv[((uint32_t)d + lsh) >> 5], 1u << (((uint32_t)d + lsh) & 31u));
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < R_MAX; ++r)
            if ((own_mask >> r) & 1u) lds_v[fd[r]] = fv[r];
        if (n_unowned) atomicAdd(&s_cnt[0], n_unowned);
        __syncthreads();
        if (tid == 0) {
            if (s_cnt[0]) atomicAdd(&F.counters[1], s_cnt[0]);
            if (n_tied) atomicAdd(&F.counters[64 + (blockIdx.x % RR_CHG_COUNTERS) * 32], (uint32_t)n_tied);
        }
        for (int i = tid; i < SURV_WORDS; i += THREADS) {
            const uint32_t sw = s_surv[i];
            if (sw) {
                const int64_t j0 = (int64_t)(lo - lsh) + 32 * (int64_t)i;
                atomicOr(&F.surv_bits[j0 >> 5], sw);
                atomicAdd(&F.tile_cnt[j0 / RR_TILE], (uint32_t)__popc(sw));
                // (how many there are in all, next to the tied-slot counts: a text without any -- random bytes -- skips the scan of tile_cnt)
                atomicAdd(&F.counters[64 + (blockIdx.x % RR_CHG_COUNTERS) * 32 + 1], (uint32_t)__popc(sw));
            }
        }
    }
    // ---- LDS -> global, in order ----
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = tid + j * THREADS;
        if (idx < size) {
            keys_out[lo + idx] = hi_bits | ((uint32_t)lds_k[idx] >> xbits);
            vals_out[lo + idx] = lds_v[idx];
        }
    }
}

}  // namespace sa
