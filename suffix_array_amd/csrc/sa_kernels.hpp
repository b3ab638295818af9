// sa_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for suffix-array
// construction.  Pure integer / indexing work: every kernel here is HBM- or latency-bound,
// none is GEMM-shaped, so there is no MFMA anywhere.
//
// Pipeline, replacing the arithmetic behind reference src/saca.rs:14:
//   k_byte_hist        which byte values occur -> symbol codes, key geometry
//   k_sample_keys / k_count_sample_dups
//                      entropy probe: may the initial sort run on the top 32 key bits only?
//   k_build_keys       packed-symbol key of every suffix (64-bit, or its top 32 bits), text tile staged in LDS
//   radix sort         stable LSD, 8-bit digits: k_radix_upsweep(32) / k_spine_rows / k_radix_downsweep_wcl<u64 | u32>
//   k_finish_sorted / k_surv_compact / k_todo_compact
//                      first refinement round straight from the sorted keys: small groups ordered in place, one pass
//   k_rr_count / k_rr_scan / k_rr_apply
//                      group heads -> ranks, SA write-back, compaction of the suffixes still tied with a neighbour
//   k_group_sort (+ _straddle), k_flag_count / k_flag_gather / k_scatter_back
//                      one refinement round: gather of the secondary key (text symbols, low key bits or ranks) fused
//                      with an in-LDS sort of the small groups; large groups through the global radix sort
//   sparse_key2, k_isa_from_sa / k_isa_tied / k_scatter_pairs
//                      prefix doubling: rank look-up without an ISA (few ties) or ISA maintenance (many)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace sa {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// ------------------------------------------------------------------------------------------
// wave / block scans (wave64 shuffles, one LDS word per wave for the cross-wave step)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t t = __shfl_up(v, o, WAVE);
        if (l >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t t = __shfl_up(v, o, WAVE);
        if (l >= o) v = v > t ? v : t;
    }
    return v;
}

// Exclusive block sum over THREADS threads; *total receives the block sum. lds: THREADS/64 + 1 words.
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t *lds, uint32_t *total)
{
    constexpr int NW = THREADS / WAVE;
    const int l = lane_id(), w = wave_id();
    uint32_t inc = wave_incl_sum(v);
    if (l == WAVE - 1) lds[w] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t s = lds[i];
        if (i < w) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

// LDS-only workgroup barrier: orders LDS traffic (lgkmcnt) but, unlike __syncthreads(), does not drain the
// vector-memory counter -- a prefetch (global -> LDS DMA) or stores in flight stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// global -> LDS DMA, BYTES per lane (4 or 16); the LDS destination is wave-uniform base + lane * BYTES
template <int BYTES>
__device__ __forceinline__ void glds(const void *src, void *lds_dst)
{
#if defined(__HIP_DEVICE_COMPILE__)      // the builtin exists in the device pass only
    if (BYTES == 16) __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
    else __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)lds_dst, 4, 0, 0);
#endif
}

// block_excl_sum with LDS-only barriers (RAW = true) or plain ones
template <int THREADS, bool RAW>
__device__ __forceinline__ uint32_t block_excl_sum_b(uint32_t v, uint32_t *lds, uint32_t *total)
{
    if (!RAW) return block_excl_sum<THREADS>(v, lds, total);
    constexpr int NW = THREADS / WAVE;
    const int l = lane_id(), w = wave_id();
    uint32_t inc = wave_incl_sum(v);
    if (l == WAVE - 1) lds[w] = inc;
    lds_barrier();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t s = lds[i];
        if (i < w) woff += s;
        tot += s;
    }
    lds_barrier();
    *total = tot;
    return woff + inc - v;
}

// Inclusive block max-scan over THREADS threads; *total receives the block max. lds: THREADS/64 words.
template <int THREADS>
__device__ __forceinline__ uint32_t block_incl_max(uint32_t v, uint32_t *lds, uint32_t *total)
{
    constexpr int NW = THREADS / WAVE;
    const int l = lane_id(), w = wave_id();
    uint32_t inc = wave_incl_max(v);
    if (l == WAVE - 1) lds[w] = inc;
    __syncthreads();
    uint32_t wmax = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t s = lds[i];
        if (i < w) wmax = wmax > s ? wmax : s;
        tot = tot > s ? tot : s;
    }
    __syncthreads();
    *total = tot;
    return inc > wmax ? inc : wmax;
}

// ------------------------------------------------------------------------------------------
// k_byte_hist: WHICH of the 256 byte values occur (hist[c] != 0 <=> c occurs).  The symbol codes
// and the key geometry only need the set of used values, not their counts, so instead of LDS
// atomics (which serialise on a 4-symbol text: 0.9 TB/s on DNA) every byte is a plain LDS store
// of 1 to its flag word -- same-address stores of a wave need no ordering.
// Algorithmic traffic: 1 B read per input byte.
// ------------------------------------------------------------------------------------------
constexpr int BH_THREADS = 256;

__global__ __launch_bounds__(BH_THREADS) void k_byte_hist(const uint8_t *__restrict__ T, int64_t n,
                                                           uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    // 16-byte aligned body, scalar head and tail
    const uintptr_t addr = (uintptr_t)T;
    int64_t head = (int64_t)((16 - (addr & 15)) & 15);
    if (head > n) head = n;
    const int64_t nvec = (n - head) / 16;
    const uint4 *V = (const uint4 *)(T + head);
    const int64_t gtid = (int64_t)blockIdx.x * BH_THREADS + threadIdx.x;
    const int64_t gstride = (int64_t)gridDim.x * BH_THREADS;
    for (int64_t i = gtid; i < nvec; i += gstride) {
        uint4 q = V[i];
        uint32_t w4[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b) h[(w4[a] >> (8 * b)) & 255u] = 1u;
        }
    }
    if (blockIdx.x == 0) {
        for (int64_t i = threadIdx.x; i < head; i += BH_THREADS) h[T[i]] = 1u;
        for (int64_t i = head + nvec * 16 + threadIdx.x; i < n; i += BH_THREADS) h[T[i]] = 1u;
    }
    __syncthreads();
    if (h[threadIdx.x]) hist[threadIdx.x] = 1u;
}

// ------------------------------------------------------------------------------------------
// k_build_keys: key[i] = the first k symbol codes of suffix i, `bits` bits each, most
// significant symbol first, zero codes past the end of the text; val[i] = i.
// The text tile (+ k bytes of halo) is read once, mapped to codes and staged in LDS; each
// thread builds one key from k LDS bytes and rolls it forward for its next 7 positions.
// Algorithmic traffic: 1 B read + 12 B written per suffix.
// ------------------------------------------------------------------------------------------
constexpr int KB_THREADS = 256;
constexpr int KB_ITEMS = 8;
constexpr int KB_TILE = KB_THREADS * KB_ITEMS;   // 2048 suffixes per workgroup
constexpr int KB_HALO = 64 + 8;                  // k <= 64 symbols, +8 so the halo loads stay 8-byte wide

struct KeyParams {
    uint8_t code[256];
    int32_t bits;      // > 0: sigma is a power of two, symbols are `bits`-wide fields (shift/or path)
    int32_t k;         // symbols per key
    uint64_t mask;     // shift path: the k * bits low bits
    uint64_t sigma;    // multiply path (bits == 0): key = sum code_i * sigma^(k-1-i), an order-preserving
    uint64_t top;      //   base-sigma number; top = sigma^(k-1).  sigma = 56 packs 11 symbols, not 10.
    const uint8_t *packed;   // bits in {1, 2, 4}: the text as bit-packed codes, first symbol in the top bits of byte 0
                             // (written by k_build_keys; 4x smaller than DNA bytes, so the rounds' random reads stay in
                             // the 256 MB Infinity Cache), zero-padded by >= 24 bytes; nullptr: read the text itself
};

// TOP32: only the top 32 bits of every key are stored (keys32), for the two-stage initial sort
template <bool TOP32>
__global__ __launch_bounds__(KB_THREADS) void k_build_keys(const uint8_t *__restrict__ T, int64_t n,
                                                            KeyParams P, uint64_t *__restrict__ keys,
                                                            uint32_t *__restrict__ vals, uint32_t *__restrict__ keys32,
                                                            int top_shift, uint8_t *__restrict__ packed_out)
{
    __shared__ uint8_t lcode[256];
    __shared__ __attribute__((aligned(16))) uint8_t c[KB_TILE + KB_HALO];
    const int tid = threadIdx.x;
    lcode[tid] = P.code[tid];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * KB_TILE;
    // stage KB_TILE + KB_HALO codes, 8 bytes per thread per step
    for (int o = tid * 8; o < KB_TILE + KB_HALO; o += KB_THREADS * 8) {
        const int64_t p = base + o;
        uint8_t b[8];
        if (p + 8 <= n && (((uintptr_t)(T + p)) & 7) == 0) {
            uint2 q = *(const uint2 *)(T + p);
            uint32_t w2[2] = { q.x, q.y };
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = lcode[(w2[j >> 2] >> (8 * (j & 3))) & 255u];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = (p + j < n) ? lcode[T[p + j]] : (uint8_t)0;
        }
        uint2 o2;
        o2.x = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        o2.y = (uint32_t)b[4] | ((uint32_t)b[5] << 8) | ((uint32_t)b[6] << 16) | ((uint32_t)b[7] << 24);
        *(uint2 *)(c + o) = o2;
    }
    __syncthreads();
    const int p0 = tid * KB_ITEMS;
    const int k = P.k, bits = P.bits;
    uint64_t kk[KB_ITEMS];
    if (bits > 0) {
        const uint64_t mask = P.mask;
        uint64_t key = 0;
        for (int j = 0; j < k; ++j) key = (key << bits) | (uint64_t)c[p0 + j];
        kk[0] = key;
#pragma unroll
        for (int r = 1; r < KB_ITEMS; ++r) {
            key = ((key << bits) | (uint64_t)c[p0 + k - 1 + r]) & mask;
            kk[r] = key;
        }
    } else {
        const uint64_t sigma = P.sigma, top = P.top;
        uint64_t key = 0;
        for (int j = 0; j < k; ++j) key = key * sigma + (uint64_t)c[p0 + j];
        kk[0] = key;
#pragma unroll
        for (int r = 1; r < KB_ITEMS; ++r) {
            key = (key - (uint64_t)c[p0 + r - 1] * top) * sigma + (uint64_t)c[p0 + k - 1 + r];
            kk[r] = key;
        }
    }
    const int64_t g0 = base + p0;
    if (packed_out && g0 < n) {
        // this thread's 8 codes -> `bits` bytes of the packed text (codes past the end are 0)
        uint32_t val = 0;
#pragma unroll
        for (int r = 0; r < KB_ITEMS; ++r) val = (val << bits) | (uint32_t)c[p0 + r];
        uint8_t *po = packed_out + (g0 >> 3) * bits;
        for (int b = 0; b < bits; ++b) po[b] = (uint8_t)(val >> (8 * (bits - 1 - b)));
    }
    if (g0 + KB_ITEMS <= n) {
        if (TOP32) {
            uint4 *ko = (uint4 *)(keys32 + g0);
            ko[0] = make_uint4((uint32_t)(kk[0] >> top_shift), (uint32_t)(kk[1] >> top_shift), (uint32_t)(kk[2] >> top_shift),
                               (uint32_t)(kk[3] >> top_shift));
            ko[1] = make_uint4((uint32_t)(kk[4] >> top_shift), (uint32_t)(kk[5] >> top_shift), (uint32_t)(kk[6] >> top_shift),
                               (uint32_t)(kk[7] >> top_shift));
        } else {
            ulonglong2 *ko = (ulonglong2 *)(keys + g0);
#pragma unroll
            for (int r = 0; r < KB_ITEMS / 2; ++r) ko[r] = make_ulonglong2(kk[2 * r], kk[2 * r + 1]);
        }
        if (vals) {                                           // nullptr: the first sort pass takes the index as the value
            uint4 *vo = (uint4 *)(vals + g0);
            const uint32_t v0 = (uint32_t)g0;
            vo[0] = make_uint4(v0, v0 + 1, v0 + 2, v0 + 3);
            vo[1] = make_uint4(v0 + 4, v0 + 5, v0 + 6, v0 + 7);
        }
    } else {
#pragma unroll
        for (int r = 0; r < KB_ITEMS; ++r)
            if (g0 + r < n) {
                if (TOP32) keys32[g0 + r] = (uint32_t)(kk[r] >> top_shift); else keys[g0 + r] = kk[r];
                if (vals) vals[g0 + r] = (uint32_t)(g0 + r);
            }
    }
}

// ------------------------------------------------------------------------------------------
// Stable LSD radix sort of (u64 key, u32 value) pairs, 8-bit digits.
// Each workgroup owns a contiguous chunk of `tiles_per_wg` tiles of SORT_TILE elements.
//   upsweep   : per-workgroup digit histogram of its chunk            (8 B read / element)
//   spine     : exclusive scan of counts[digit][workgroup]            (negligible)
//   downsweep : rank inside the tile with wave-wide digit matching (ballots), stage the tile
//               in sorted order in LDS, write digit runs out coalesced (12 B read + 12 B written)
// Algorithmic traffic per pass: 24 B / element (what a single-pass onesweep would move);
// this three-kernel form moves 32 B / element.
// ------------------------------------------------------------------------------------------
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / WAVE;

__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift, uint32_t dmask)
{
    return (uint32_t)(key >> shift) & dmask;
}
__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift, uint32_t dmask)
{
    return (key >> shift) & dmask;
}

__global__ __launch_bounds__(SORT_THREADS) void k_radix_upsweep(const uint64_t *__restrict__ keys,
                                                                 uint32_t *__restrict__ counts, int64_t n,
                                                                 int shift, uint32_t dmask,
                                                                 int64_t chunk_elems, int G, int split,
                                                                 int64_t sub_elems)
{
    __shared__ uint32_t h[SORT_WAVES][RADIX];
    for (int i = threadIdx.x; i < SORT_WAVES * RADIX; i += SORT_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[wave_id()];
    // `split` workgroups share one downsweep chunk (more waves in flight for the streaming read);
    // sub_elems is even, so every part starts 16-byte aligned
    const int g = (int)(blockIdx.x / split), part = (int)(blockIdx.x % split);
    const int64_t cbegin = (int64_t)g * chunk_elems;
    int64_t cend = cbegin + chunk_elems;
    if (cend > n) cend = n;
    int64_t begin = cbegin + (int64_t)part * sub_elems;
    int64_t end = begin + sub_elems;
    if (begin > cend) begin = cend;
    if (end > cend || part == split - 1) end = cend;
    // two keys (16 B) per lane per load, four loads in flight per lane; chunk_elems is a multiple of
    // the tile size so the pairs are 16-byte aligned
    const int64_t npair = (end - begin) / 2;
    const ulonglong2 *K2 = (const ulonglong2 *)(keys + begin);
    auto count2 = [&](const ulonglong2 &q) {
        const uint32_t d0 = digit_of((uint64_t)q.x, shift, dmask), d1 = digit_of((uint64_t)q.y, shift, dmask);
        // constant digits (all-equal high bits) would serialise the LDS atomic 64 ways
        const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)d0);
        const uint64_t act = __ballot(1);                          // evaluated by every active lane
        if (__all(d0 == f && d1 == f)) {
            // the lowest ACTIVE lane adds for the wave (lane 0 may have left the loop already)
            if (lane_id() == __ffsll((unsigned long long)act) - 1) atomicAdd(&mine[f], 2u * (uint32_t)__popcll(act));
        } else {
            atomicAdd(&mine[d0], 1u);
            atomicAdd(&mine[d1], 1u);
        }
    };
    int64_t i = threadIdx.x;
    for (; i + 3 * SORT_THREADS < npair; i += 4 * SORT_THREADS) {
        const ulonglong2 q0 = K2[i], q1 = K2[i + SORT_THREADS], q2 = K2[i + 2 * SORT_THREADS], q3 = K2[i + 3 * SORT_THREADS];
        count2(q0); count2(q1); count2(q2); count2(q3);
    }
    for (; i < npair; i += SORT_THREADS) count2(K2[i]);
    if (((end - begin) & 1) && threadIdx.x == 0) atomicAdd(&mine[digit_of(keys[end - 1], shift, dmask)], 1u);
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) s += h[w][threadIdx.x];
    if (split == 1) counts[(int64_t)threadIdx.x * G + g] = s;
    else if (s) atomicAdd(&counts[(int64_t)threadIdx.x * G + g], s);     // counts zeroed by the host
}

constexpr int SPINE_THREADS = 1024;

// the same for 32-bit keys (two-stage initial sort: only the top 32 key bits are sorted), four keys per 16-byte load
__global__ __launch_bounds__(SORT_THREADS) void k_radix_upsweep32(const uint32_t *__restrict__ keys,
                                                                   uint32_t *__restrict__ counts, int64_t n,
                                                                   int shift, uint32_t dmask,
                                                                   int64_t chunk_elems, int G, int split,
                                                                   int64_t sub_elems)
{
    __shared__ uint32_t h[SORT_WAVES][RADIX];
    for (int i = threadIdx.x; i < SORT_WAVES * RADIX; i += SORT_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[wave_id()];
    const int g = (int)(blockIdx.x / split), part = (int)(blockIdx.x % split);
    const int64_t cbegin = (int64_t)g * chunk_elems;
    int64_t cend = cbegin + chunk_elems;
    if (cend > n) cend = n;
    int64_t begin = cbegin + (int64_t)part * sub_elems;      // sub_elems is a multiple of 4: 16-byte aligned parts
    int64_t end = begin + sub_elems;
    if (begin > cend) begin = cend;
    if (end > cend || part == split - 1) end = cend;
    const int64_t nquad = (end - begin) / 4;
    const uint4 *K4 = (const uint4 *)(keys + begin);
    auto count4 = [&](const uint4 &q) {
        const uint32_t d0 = digit_of((uint32_t)q.x, shift, dmask), d1 = digit_of((uint32_t)q.y, shift, dmask);
        const uint32_t d2 = digit_of((uint32_t)q.z, shift, dmask), d3 = digit_of((uint32_t)q.w, shift, dmask);
        const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)d0);
        const uint64_t act = __ballot(1);
        if (__all(d0 == f && d1 == f && d2 == f && d3 == f)) {
            if (lane_id() == __ffsll((unsigned long long)act) - 1) atomicAdd(&mine[f], 4u * (uint32_t)__popcll(act));
        } else {
            atomicAdd(&mine[d0], 1u); atomicAdd(&mine[d1], 1u); atomicAdd(&mine[d2], 1u); atomicAdd(&mine[d3], 1u);
        }
    };
    int64_t i = threadIdx.x;
    for (; i + 3 * SORT_THREADS < nquad; i += 4 * SORT_THREADS) {
        const uint4 q0 = K4[i], q1 = K4[i + SORT_THREADS], q2 = K4[i + 2 * SORT_THREADS], q3 = K4[i + 3 * SORT_THREADS];
        count4(q0); count4(q1); count4(q2); count4(q3);
    }
    for (; i < nquad; i += SORT_THREADS) count4(K4[i]);
    if (threadIdx.x == 0)
        for (int64_t t = begin + nquad * 4; t < end; ++t) atomicAdd(&mine[digit_of(keys[t], shift, dmask)], 1u);
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) s += h[w][threadIdx.x];
    if (split == 1) counts[(int64_t)threadIdx.x * G + g] = s;
    else if (s) atomicAdd(&counts[(int64_t)threadIdx.x * G + g], s);
}

// Spine of one radix pass: block d turns counts[d][0..G) into exclusive prefixes (in place) and
// writes the digit total; the downsweep prologue scans the 256 totals itself.  G <= 1024.
__global__ __launch_bounds__(SPINE_THREADS) void k_spine_rows(uint32_t *__restrict__ counts,
                                                               uint32_t *__restrict__ digit_tot, int G)
{
    __shared__ uint32_t lds[SPINE_THREADS / WAVE + 1];
    const int d = blockIdx.x, g = threadIdx.x;
    const uint32_t c = g < G ? counts[(int64_t)d * G + g] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_sum<SPINE_THREADS>(c, lds, &tot);
    if (g < G) counts[(int64_t)d * G + g] = ex;
    if (g == 0) digit_tot[d] = tot;
}

// One tile of THREADS * ITEMS pairs.  Element order inside the tile is wave-striped
// (e = wave * 64 * ITEMS + item * 64 + lane), so every global load is a contiguous 512-B (keys) or
// 256-B (values) burst per wave and "item-major, lane-minor" is the tile order that stability
// is defined on.
//
// Ranking: for every item the lanes of a wave that hold the same digit are found with 8 ballots.
// x accumulates, per lane, the lanes that differ from it in some digit bit (ballot XOR the lane's
// own bit, sign-extended), so ~x is the match mask; v_mbcnt gives the number of matching lanes
// below, v_bcnt the group size.  The lowest matching lane reads-then-bumps the wave's LDS counter
// of that digit (no atomics: one wave executes its LDS operations in order).
template <int THREADS, int ITEMS, bool FULL, int ABLATE>
__device__ __forceinline__ void sort_tile(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                          uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                          int64_t base, int valid, int shift, uint32_t dmask,
                                          uint64_t *lds_kv, uint32_t (*wave_hist)[RADIX], uint32_t *digit_base,
                                          uint32_t *run_off, uint32_t *scan_lds)
{
    constexpr int NWAVES = THREADS / WAVE;
    constexpr int WAVE_ELEMS = WAVE * ITEMS;
    static_assert(THREADS >= RADIX, "one thread per digit is assumed");
    static_assert(ITEMS % 4 == 0, "digits are packed four to a register");
    const int tid = threadIdx.x, l = lane_id(), w = wave_id();
    const int e0 = w * WAVE_ELEMS + l;
    uint64_t key[ITEMS];
    uint32_t pos[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = e0 + j * WAVE;
        key[j] = (FULL || e < valid) ? ((ABLATE & 4) ? __builtin_nontemporal_load(keys_in + base + e) : keys_in[base + e]) : ~0ull;
    }
    for (int i = tid; i < NWAVES * RADIX; i += THREADS) (&wave_hist[0][0])[i] = 0;
    __syncthreads();
    uint32_t *my_hist = wave_hist[w];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const bool ok = FULL || (e0 + j * WAVE) < valid;
        const uint32_t d = digit_of(key[j], shift, dmask);
        if (ABLATE & 32) { pos[j] = (uint32_t)(e0 + j * WAVE); continue; }   // timing-only: no ranking at all
        uint32_t xlo = 0, xhi = 0;
        if (!FULL) { const uint64_t okm = __ballot(ok); xlo = ~(uint32_t)okm; xhi = ~(uint32_t)(okm >> 32); }
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const uint32_t sel = (uint32_t)((int32_t)(d << (31 - b)) >> 31);      // 0 or ~0: my bit b
            const uint64_t bal = __ballot(sel != 0);
            xlo |= (uint32_t)bal ^ sel;
            xhi |= (uint32_t)(bal >> 32) ^ sel;
        }
        const uint32_t mlo = ~xlo, mhi = ~xhi;
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        const uint32_t prior = my_hist[d];
        if (ok && below == 0) my_hist[d] = prior + (uint32_t)(__popc(mlo) + __popc(mhi));
        pos[j] = prior + below;
        __builtin_amdgcn_sched_barrier(0);   // keep the items apart: interleaving them only adds SGPR pressure
    }
    // the values are only needed after the keys have left; issue their loads now so that the
    // latency hides behind the prefix step and the key scatter
    uint32_t val[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = e0 + j * WAVE;
        val[j] = (FULL || e < valid) ? (!vals_in ? (uint32_t)(base + e) : (ABLATE & 4) ? __builtin_nontemporal_load(vals_in + base + e) : vals_in[base + e]) : 0u;
    }
    __syncthreads();
    // thread d: per-wave counts of digit d -> per-wave offsets; exclusive scan of the digit totals
    uint32_t tot = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int ww = 0; ww < NWAVES; ++ww) {
            const uint32_t cnt = wave_hist[ww][tid];
            wave_hist[ww][tid] = tot;
            tot += cnt;
        }
    }
    uint32_t tile_total;
    const uint32_t dbase = block_excl_sum<THREADS>(tot, scan_lds, &tile_total);
    if (tid < RADIX) digit_base[tid] = dbase;
    __syncthreads();
    // keys -> LDS in sorted order, then out: a digit's run leaves as one contiguous burst
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t d = digit_of(key[j], shift, dmask);
        if (!(ABLATE & 32)) pos[j] += digit_base[d] + my_hist[d];
        if (FULL || (e0 + j * WAVE) < valid) lds_kv[pos[j]] = key[j];
    }
    __syncthreads();
    uint32_t dpack[ITEMS / 4];      // digits of the elements this thread writes out, 4 per register
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = tid + j * THREADS;
        if ((j & 3) == 0) dpack[j >> 2] = 0;
        if (FULL || idx < valid) {
            const uint64_t kx = lds_kv[idx];
            const uint32_t d = digit_of(kx, shift, dmask);
            dpack[j >> 2] |= d << (8 * (j & 3));
            if ((ABLATE & 16) && kx != 0x0123456789abcdefULL) continue;   // timing-only build: no stores
            if (ABLATE & 1) keys_out[base + idx] = kx;   // timing-only build: sequential instead of scattered stores
            else if (ABLATE & 8) __builtin_nontemporal_store(kx, keys_out + (run_off[d] + ((uint32_t)idx - digit_base[d])));
            else keys_out[run_off[d] + ((uint32_t)idx - digit_base[d])] = kx;
        }
    }
    __syncthreads();
    uint32_t *lds_v = (uint32_t *)lds_kv;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (FULL || (e0 + j * WAVE) < valid) lds_v[pos[j]] = val[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = tid + j * THREADS;
        if (FULL || idx < valid) {
            const uint32_t d = (dpack[j >> 2] >> (8 * (j & 3))) & 255u;
            if ((ABLATE & 16) && lds_v[idx] != 0x01234567u) continue;
            if (ABLATE & 1) vals_out[base + idx] = lds_v[idx];
            else if (ABLATE & 8) __builtin_nontemporal_store(lds_v[idx], vals_out + (run_off[d] + ((uint32_t)idx - digit_base[d])));
            else vals_out[run_off[d] + ((uint32_t)idx - digit_base[d])] = lds_v[idx];
        }
    }
    __syncthreads();
    if (tid < RADIX) run_off[tid] += tot;     // thread d owns run_off[d]; the next tile starts behind a barrier
    __syncthreads();
}

// MINW = minimum waves per SIMD the register allocation has to allow (launch-bounds 2nd argument)
template <int THREADS, int ITEMS, int MINW, int ABLATE = 0>
__global__ __launch_bounds__(THREADS, MINW) void k_radix_downsweep(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint64_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, uint32_t *__restrict__ spine, const uint32_t *__restrict__ digit_tot,
    int64_t n, int shift, uint32_t dmask, int64_t tiles_per_wg, int G)
{
    constexpr int TILE = THREADS * ITEMS;
    constexpr int NWAVES = THREADS / WAVE;
    __shared__ __attribute__((aligned(16))) uint64_t lds_kv[TILE];
    __shared__ uint32_t wave_hist[NWAVES][RADIX];
    __shared__ uint32_t digit_base[RADIX];
    __shared__ uint32_t run_off[RADIX];
    __shared__ uint32_t scan_lds[NWAVES + 1];
    {
        // global start of digit d = totals of the smaller digits + this workgroup's row prefix
        uint32_t all;
        const uint32_t t = threadIdx.x < RADIX ? digit_tot[threadIdx.x] : 0u;
        const uint32_t gbase = block_excl_sum<THREADS>(t, scan_lds, &all);
        if (threadIdx.x < RADIX) {
            run_off[threadIdx.x] = gbase + spine[(int64_t)threadIdx.x * G + blockIdx.x];
            spine[(int64_t)threadIdx.x * G + blockIdx.x] = 0;      // consumed: the next pass's upsweep accumulates from zero again
        }
    }
    __syncthreads();
    const int64_t tile0 = (int64_t)blockIdx.x * tiles_per_wg;
    for (int64_t t = 0; t < tiles_per_wg; ++t) {
        const int64_t base = (tile0 + t) * TILE;
        if (base >= n) break;
        const int64_t rem = n - base;
        if (rem >= TILE)
            sort_tile<THREADS, ITEMS, true, ABLATE>(keys_in, vals_in, keys_out, vals_out, base, TILE, shift, dmask, lds_kv,
                                            wave_hist, digit_base, run_off, scan_lds);
        else
            sort_tile<THREADS, ITEMS, false, ABLATE>(keys_in, vals_in, keys_out, vals_out, base, (int)rem, shift, dmask, lds_kv,
                                             wave_hist, digit_base, run_off, scan_lds);
    }
}

// ------------------------------------------------------------------------------------------
// k_radix_downsweep_wcl: the plain tile scatter of k_radix_downsweep plus a per-digit LDS carry, so
// that every 128-byte line a tile touches is completed by that same tile (within microseconds,
// from one CU) instead of by the workgroup's next tile (tens of microseconds later, after the
// 4 MiB L2 of the XCD has been swept several times).  Per digit d the workgroup keeps
//   c0[d]  next global position,  w0[d]  everything below is stored (16-aligned after the first
//   tile);  carry[d][0 .. c0-w0)  the elements in between (< 16).
// Tile: new elements of digit d would go to [c0, c0+cnt).  Only positions below
// w1 = max(w0, (c0+cnt) & ~15) are stored now -- first the old carry ([w0, c0), loop A), then the
// tile's own elements -- and the rest lands in the carry at index (position - w1).
// ------------------------------------------------------------------------------------------
// diagnostic build only (STAMP): cycles of wave 0 per phase, summed over tiles and workgroups
__device__ unsigned long long g_phase_cycles[16];
__device__ int g_gs_stamp_on;          // diagnostic: k_group_sort adds its per-phase cycles (wave 0) to g_phase_cycles[8..15]

// PF > 0: the first PF of a thread's ITEMS keys of the NEXT tile are prefetched into LDS by global->LDS DMA
// (no registers) right after this tile's keys are in registers, and stay in flight across the LDS-only
// barriers of the ranking / prefix / staging phases; the tile's own stores are issued behind them.
template <int THREADS, int ITEMS, int GR = 16, int MINW = 1, bool STAMP = false, typename KeyT = uint64_t, int PF = 0>
__global__ __launch_bounds__(THREADS, MINW) void k_radix_downsweep_wcl(
    const KeyT *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, KeyT *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, uint32_t *__restrict__ spine, const uint32_t *__restrict__ digit_tot,
    int64_t n, int shift, uint32_t dmask, int64_t tiles_per_wg, int G)
{
    constexpr int TILE = THREADS * ITEMS;
    constexpr int NWAVES = THREADS / WAVE;
    constexpr int WAVE_ELEMS = WAVE * ITEMS;
    constexpr int CSTEPS = (RADIX * GR + THREADS - 1) / THREADS;
    static_assert(THREADS >= RADIX, "thread / digit mapping");
    static_assert(ITEMS % 4 == 0, "digits are packed four to a register");
    static_assert(TILE < 65536, "16-bit tile-local counters");
    __shared__ __attribute__((aligned(16))) KeyT lds_kv[TILE];
    __shared__ __attribute__((aligned(16))) KeyT carry_k[RADIX * GR];
    __shared__ __attribute__((aligned(16))) uint32_t carry_v[RADIX * GR];
    __shared__ uint16_t wave_hist[NWAVES][RADIX];   // a wave holds 64 * ITEMS <= 65535 elements, a tile < 65536
    __shared__ uint32_t digit_base[RADIX];     // first stage slot of digit d
    __shared__ uint32_t goff[RADIX];           // c0 - digit_base: global position = goff + stage slot
    __shared__ uint32_t wlim[RADIX];           // w1: positions below are stored by this tile
    __shared__ uint32_t aold[RADIX];           // w0 | pending << 28 is too narrow -> w0 only; pending in apnd
    __shared__ uint32_t apnd[RADIX];           // old carry elements to flush this tile (0 when nothing is stored)
    __shared__ uint32_t scan_lds[NWAVES + 1];
    constexpr int KPL = 16 / (int)sizeof(KeyT);                   // keys per lane of one 16-byte DMA
    constexpr int PF_WAVE = WAVE * (PF > 0 ? PF : 1);             // prefetched keys per wave
    static_assert(PF == 0 || (PF % KPL == 0 && PF <= ITEMS), "whole 1 KiB DMA instructions");
    __shared__ __attribute__((aligned(16))) KeyT next_k[PF > 0 ? THREADS * PF : KPL];
    bool have_pre = false;                                        // next_k holds this tile's first PF items per thread

    const int tid = threadIdx.x, l = lane_id(), w = wave_id();
    uint32_t c0 = 0, w0 = 0;
    {
        uint32_t all;
        const uint32_t t = tid < RADIX ? digit_tot[tid] : 0u;
        const uint32_t gbase = block_excl_sum<THREADS>(t, scan_lds, &all);
        if (tid < RADIX) {
            c0 = w0 = gbase + spine[(int64_t)tid * G + blockIdx.x];
            spine[(int64_t)tid * G + blockIdx.x] = 0;      // consumed: the next pass's upsweep accumulates from zero again
        }
    }
    uint16_t *my_hist = wave_hist[w];
    uint32_t *lds_v = (uint32_t *)lds_kv;
    const int e0 = w * WAVE_ELEMS + l;
    const int64_t tile0 = (int64_t)blockIdx.x * tiles_per_wg;
    unsigned long long t_prev = 0;
    auto stamp = [&](int phase) {
        if (STAMP && tid == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (phase >= 0) atomicAdd(&g_phase_cycles[phase], now - t_prev);
            t_prev = now;
        }
    };
    // global -> LDS DMA of (the first PF items per thread of) the tile at nb; LDS index = element index (PF == ITEMS)
    // or w * PF_WAVE + 64 j + lane.  A full tile moves as 1 KiB wave-instructions, the last partial one dword by dword.
    auto prefetch = [&](int64_t nb) {
        if (nb >= n) return;
        const int64_t left = n - nb;
        if (left >= TILE) {
#pragma unroll
            for (int i = 0; i < PF / KPL; ++i) {
                const KeyT *src = keys_in + nb + w * WAVE_ELEMS + i * (WAVE * KPL) + l * KPL;
                glds<16>(src, next_k + w * PF_WAVE + i * (WAVE * KPL));
            }
        } else if (PF == ITEMS) {
            constexpr int DW = (int)sizeof(KeyT) / 4;                // dwords per key
            const uint32_t *src32 = (const uint32_t *)(keys_in + nb);
#pragma unroll
            for (int i = 0; i < PF * DW; ++i) {
                const int dw = (w * WAVE_ELEMS) * DW + i * WAVE + l;  // dword index inside the tile
                if (dw < (int)left * DW)
                    glds<4>(src32 + dw, (uint32_t *)next_k + (w * WAVE_ELEMS) * DW + i * WAVE);
            }
        }
    };
    if (PF == ITEMS) {
        prefetch(tile0 * TILE);
        __syncthreads();                                              // vmcnt(0) + barrier: the first tile's keys are in LDS
    }
    for (int64_t t = 0; t < tiles_per_wg; ++t) {
        const int64_t base = (tile0 + t) * TILE;
        if (base >= n) break;
        stamp(-1);
        const int valid = (n - base) >= TILE ? TILE : (int)(n - base);
        const bool full = valid == TILE;
        KeyT key[ITEMS];
        uint32_t pos[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int e = e0 + j * WAVE;
            if (PF == ITEMS) key[j] = (full || e < valid) ? next_k[e] : (KeyT)~(KeyT)0;          // every key comes through LDS
            else if (PF > 0 && j < PF && have_pre) key[j] = next_k[w * PF_WAVE + j * WAVE + l];  // (have_pre: the tile is full)
            else key[j] = (full || e < valid) ? keys_in[base + e] : (KeyT)~(KeyT)0;
        }
        for (int i = tid; i < NWAVES * RADIX / 2; i += THREADS) ((uint32_t *)&wave_hist[0][0])[i] = 0;
        __syncthreads();
        if (PF == ITEMS && t + 1 < tiles_per_wg) prefetch(base + TILE);   // all waves hold their keys: the buffer takes the next tile
        stamp(0);      // key loads issued, counters zeroed
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const bool ok = full || (e0 + j * WAVE) < valid;
            const uint32_t d = digit_of(key[j], shift, dmask);
            const uint64_t okm = __ballot(ok);
            uint32_t xlo = ~(uint32_t)okm, xhi = ~(uint32_t)(okm >> 32);
#pragma unroll
            for (int b = 0; b < RADIX_BITS; ++b) {
                const uint32_t sel = (uint32_t)((int32_t)(d << (31 - b)) >> 31);
                const uint64_t bal = __ballot(sel != 0);
                xlo |= (uint32_t)bal ^ sel;
                xhi |= (uint32_t)(bal >> 32) ^ sel;
            }
            const uint32_t mlo = ~xlo, mhi = ~xhi;
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
            const uint32_t prior = my_hist[d];
            if (ok && below == 0) my_hist[d] = (uint16_t)(prior + (uint32_t)(__popc(mlo) + __popc(mhi)));
            pos[j] = prior + below;
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(1);      // ranking (includes the wait for the keys)
        if (PF > 0 && PF < ITEMS) {
            // partial prefetch: issued only now, after the last use of the keys that came by ordinary loads (hipcc waits
            // vmcnt(0), DMA included, at such a use); it overlaps the prefix, the staging and the carry stores
            const int64_t nb = base + TILE;
            have_pre = (t + 1 < tiles_per_wg) && (nb + TILE <= n);
            if (have_pre) prefetch(nb);
        }
        uint32_t val[ITEMS];
        if (vals_in) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int e = e0 + j * WAVE;
                val[j] = (full || e < valid) ? vals_in[base + e] : 0u;
            }
        } else {                                                  // no values array: the value is the index itself
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) val[j] = (uint32_t)(base + e0 + j * WAVE);
        }
        if (PF > 0) lds_barrier(); else __syncthreads();
        stamp(2);      // value loads issued + barrier
        // ---- thread d: per-wave offsets, digit totals, carry bookkeeping ----
        uint32_t tot = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int ww = 0; ww < NWAVES; ++ww) {
                const uint32_t cnt = wave_hist[ww][tid];
                wave_hist[ww][tid] = (uint16_t)tot;
                tot += cnt;
            }
        }
        uint32_t tile_total;
        const uint32_t dbase = block_excl_sum_b<THREADS, (PF > 0)>(tot, scan_lds, &tile_total);
        if (tid < RADIX) {
            const uint32_t c1 = c0 + tot;
            const uint32_t fl = c1 & ~(uint32_t)(GR - 1);
            const uint32_t w1 = fl > w0 ? fl : w0;
            digit_base[tid] = dbase;
            goff[tid] = c0 - dbase;
            wlim[tid] = w1;
            aold[tid] = w0;
            apnd[tid] = w1 > w0 ? c0 - w0 : 0u;      // flush the old carry only when this tile stores something
            c0 = c1;
            w0 = w1;
        }
        if (PF > 0) lds_barrier(); else __syncthreads();
        stamp(3);      // per-digit prefix + carry bookkeeping
        // ---- keys: stage in sorted order ----
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t d = digit_of(key[j], shift, dmask);
            pos[j] += digit_base[d] + my_hist[d];
            if (full || (e0 + j * WAVE) < valid) lds_kv[pos[j]] = key[j];
        }
        // loop A (keys): the old carry of every digit that stores something goes out first
#pragma unroll
        for (int u = 0; u < CSTEPS; ++u) {
            const uint32_t i = (uint32_t)tid + (uint32_t)u * THREADS;
            const uint32_t d = i / GR, k = i % GR;
            if (i < (uint32_t)(RADIX * GR) && k < apnd[d]) keys_out[aold[d] + k] = carry_k[i];
        }
        if (PF > 0) lds_barrier(); else __syncthreads();     // stores stay in flight across the LDS-only barrier
        stamp(4);      // keys -> LDS, old carry out
        uint32_t dpack[ITEMS / 4];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int idx = tid + j * THREADS;
            if ((j & 3) == 0) dpack[j >> 2] = 0;
            if (full || idx < valid) {
                const KeyT kx = lds_kv[idx];
                const uint32_t d = digit_of(kx, shift, dmask);
                dpack[j >> 2] |= d << (8 * (j & 3));
                const uint32_t gp = goff[d] + (uint32_t)idx, lim = wlim[d];
                if (gp < lim) keys_out[gp] = kx;
                else carry_k[d * GR + (gp - lim)] = kx;
            }
        }
        if (PF > 0) lds_barrier(); else __syncthreads();     // stores stay in flight across the LDS-only barrier
        stamp(5);      // keys LDS -> global
        // ---- values: the same through the same stage ----
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (full || (e0 + j * WAVE) < valid) lds_v[pos[j]] = val[j];
#pragma unroll
        for (int u = 0; u < CSTEPS; ++u) {
            const uint32_t i = (uint32_t)tid + (uint32_t)u * THREADS;
            const uint32_t d = i / GR, k = i % GR;
            if (i < (uint32_t)(RADIX * GR) && k < apnd[d]) vals_out[aold[d] + k] = carry_v[i];
        }
        if (PF > 0) lds_barrier(); else __syncthreads();     // stores stay in flight across the LDS-only barrier
        stamp(6);      // values -> LDS, old carry out
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int idx = tid + j * THREADS;
            if (full || idx < valid) {
                const uint32_t d = (dpack[j >> 2] >> (8 * (j & 3))) & 255u;
                const uint32_t gp = goff[d] + (uint32_t)idx, lim = wlim[d];
                const uint32_t vx = lds_v[idx];
                if (gp < lim) vals_out[gp] = vx;
                else carry_v[d * GR + (gp - lim)] = vx;
            }
        }
        __syncthreads();
        stamp(7);      // values LDS -> global
    }
    // ---- flush what is left in the carries (at most 15 elements per digit) ----
    if (tid < RADIX) { aold[tid] = w0; apnd[tid] = c0 - w0; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CSTEPS; ++u) {
        const uint32_t i = (uint32_t)tid + (uint32_t)u * THREADS;
        const uint32_t d = i / GR, k = i % GR;
        if (i < (uint32_t)(RADIX * GR) && k < apnd[d]) { keys_out[aold[d] + k] = carry_k[i]; vals_out[aold[d] + k] = carry_v[i]; }
    }
}

// ------------------------------------------------------------------------------------------
// Re-rank: m sorted (key, suffix) pairs sitting in slots U[0..m) of SA (FIRST: U[j] = j).
// A group = maximal run of equal keys; its rank is (slot of its first element) + 1.
//   k_rr_count : per tile, how many elements stay tied with a neighbour, and the last group
//                head slot (+1) inside the tile
//   k_rr_scan  : exclusive sum / exclusive max over the tiles (one workgroup)
//   k_rr_apply : SA[U[j]] = V[j]; ISA[V[j]] = rank; compact (slot, group head, suffix) of the
//                elements that are still tied
// Algorithmic traffic per element: 12 B read twice (keys + vals [+ 4 B slot]), 4 B SA write,
// 4 B ISA scatter, 12 B per surviving element.
// ------------------------------------------------------------------------------------------
constexpr int RR_THREADS = 1024;
constexpr int RR_ITEMS = 8;
constexpr int RR_TILE = RR_THREADS * RR_ITEMS;   // 8192

constexpr int RR_WAVE_ELEMS = WAVE * RR_ITEMS;   // 512 consecutive elements per wave, item r of lane l = base + 64 r + l

// Head / tied masks of one wave's 512 elements.  Loads are wave-striped (512 contiguous bytes per
// instruction); the neighbour keys come from shuffles, so the whole classification is 8 ballots
// and scalar bit arithmetic: head[r] bit l = element (r, l) starts a group, tied[r] bit l = it is
// in a group of more than one element, valid[r] = it exists.
struct WaveGroups { uint64_t head[RR_ITEMS], tied[RR_ITEMS], valid[RR_ITEMS]; };

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src, WAVE) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src, WAVE);
}

template <typename KeyT>
__device__ __forceinline__ WaveGroups rr_wave_classify(const KeyT *__restrict__ keys, int64_t m, int64_t wbase, int key_shift)
{
    const int l = lane_id();
    uint64_t k[RR_ITEMS];
    WaveGroups g;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        k[r] = i < m ? ((uint64_t)keys[i] >> key_shift) : 0ull; // groups are runs of equal key >> key_shift
        g.valid[r] = __ballot(i < m);
    }
    const uint64_t before = (wbase > 0 && wbase - 1 < m) ? ((uint64_t)keys[wbase - 1] >> key_shift) : 0ull;
    const uint64_t after = (wbase + RR_WAVE_ELEMS < m) ? ((uint64_t)keys[wbase + RR_WAVE_ELEMS] >> key_shift) : 0ull;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        uint64_t up = shfl64(k[r], l ? l - 1 : 0);
        const uint64_t prev_last = r ? shfl64(k[r ? r - 1 : 0], 63) : before;   // executed by every lane (no shuffle under divergence)
        if (l == 0) up = prev_last;
        const int64_t i = wbase + 64 * r + l;
        g.head[r] = __ballot(i < m && (i == 0 || k[r] != up));
    }
    const uint64_t last_key = shfl64(k[RR_ITEMS - 1], 63);
    const bool boundary_after = (wbase + RR_WAVE_ELEMS >= m) || after != last_key;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        // is the NEXT element a group boundary (a head, or past the end)?
        const uint64_t bnd = g.head[r] | ~g.valid[r];
        const uint64_t bnd_next0 = (r + 1 < RR_ITEMS) ? ((g.head[(r + 1) % RR_ITEMS] | ~g.valid[(r + 1) % RR_ITEMS]) & 1ull)
                                                      : (boundary_after ? 1ull : 0ull);
        const uint64_t next = (bnd >> 1) | (bnd_next0 << 63);
        g.tied[r] = g.valid[r] & ~(g.head[r] & next);
    }
    return g;
}

template <bool FIRST, typename KeyT = uint64_t>
__global__ __launch_bounds__(RR_THREADS) void k_rr_count(const KeyT *__restrict__ keys,
                                                          const uint32_t *__restrict__ U, int64_t m,
                                                          uint32_t *__restrict__ tile_cnt,
                                                          uint32_t *__restrict__ tile_head, int key_shift)
{
    __shared__ uint32_t wcnt[RR_THREADS / WAVE], whead[RR_THREADS / WAVE];
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)wave_id() * RR_WAVE_ELEMS;
    const WaveGroups g = rr_wave_classify(keys, m, wbase, key_shift);
    uint32_t cnt = 0, lasthead = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        cnt += (uint32_t)__popcll(g.tied[r]);
        if (g.head[r]) {
            const int64_t i = wbase + 64 * r + (63 - __builtin_clzll(g.head[r]));
            lasthead = (FIRST ? (uint32_t)i : U[i]) + 1u;       // slots grow with the index: the last head wins
        }
    }
    if (lane_id() == 0) { wcnt[wave_id()] = cnt; whead[wave_id()] = lasthead; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0, mx = 0;
        for (int w = 0; w < RR_THREADS / WAVE; ++w) { tot += wcnt[w]; mx = mx > whead[w] ? mx : whead[w]; }
        tile_cnt[blockIdx.x] = tot;
        tile_head[blockIdx.x] = mx;
    }
}

// one workgroup: tile_cnt -> exclusive sums (+ total), tile_head -> exclusive running max.
// Every thread owns a contiguous run of entries (a multiple of 4, read and written as 16-byte vectors:
// the run is a chain of dependent L2 accesses, so fewer, wider ones).
__global__ __launch_bounds__(SPINE_THREADS) void k_rr_scan(uint32_t *__restrict__ tile_cnt,
                                                            uint32_t *__restrict__ tile_head, int64_t tiles,
                                                            uint32_t *__restrict__ out_total)
{
    __shared__ uint32_t lds[SPINE_THREADS / WAVE + 1];
    const int64_t per = ((tiles + SPINE_THREADS - 1) / SPINE_THREADS + 3) & ~(int64_t)3;
    int64_t b = (int64_t)threadIdx.x * per, e = b + per;
    if (b > tiles) b = tiles;
    if (e > tiles) e = tiles;
    const int64_t ev = b + ((e - b) & ~(int64_t)3);           // end of the whole vectors
    uint32_t s = 0, mx = 0;
    for (int64_t i = b; i < ev; i += 4) {
        const uint4 c = *(const uint4 *)(tile_cnt + i), h = *(const uint4 *)(tile_head + i);
        s += c.x + c.y + c.z + c.w;
        mx = max(max(mx, max(h.x, h.y)), max(h.z, h.w));
    }
    for (int64_t i = ev; i < e; ++i) { s += tile_cnt[i]; mx = max(mx, tile_head[i]); }
    uint32_t tot, mtot;
    uint32_t off = block_excl_sum<SPINE_THREADS>(s, lds, &tot);
    uint32_t incm = block_incl_max<SPINE_THREADS>(mx, lds, &mtot);
    // exclusive max for this thread = inclusive max of the previous thread
    uint32_t prevm = __shfl_up(incm, 1, WAVE);
    __shared__ uint32_t wlast[SPINE_THREADS / WAVE];
    if (lane_id() == WAVE - 1) wlast[wave_id()] = incm;
    __syncthreads();
    if (lane_id() == 0) prevm = wave_id() ? wlast[wave_id() - 1] : 0u;
    uint32_t run = prevm;
    for (int64_t i = b; i < ev; i += 4) {
        const uint4 c = *(const uint4 *)(tile_cnt + i), h = *(const uint4 *)(tile_head + i);
        uint4 oc, oh;
        oc.x = off; off += c.x; oc.y = off; off += c.y; oc.z = off; off += c.z; oc.w = off; off += c.w;
        oh.x = run; run = max(run, h.x); oh.y = run; run = max(run, h.y); oh.z = run; run = max(run, h.z); oh.w = run; run = max(run, h.w);
        *(uint4 *)(tile_cnt + i) = oc;
        *(uint4 *)(tile_head + i) = oh;
    }
    for (int64_t i = ev; i < e; ++i) {
        uint32_t c = tile_cnt[i], h = tile_head[i];
        tile_cnt[i] = off; off += c;
        tile_head[i] = run; run = run > h ? run : h;
    }
    if (threadIdx.x == 0) *out_total = tot;
}

// ISA_MODE: 0 = scatter ISA[suffix] = rank directly, 1 = the same plus the has_isa bitmap (sparse
// refinement), 3 = no rank output at all (text-keyed rounds), 4 = like 3, but the still-tied suffixes are recorded by slot
// (ISA = group heads, has_isa = bitmap, pair_v = counts per tile) instead of being listed, 2 = write (suffix, rank) pairs in slot order; the host bins them by suffix position
// with one radix pass and k_scatter_pairs then writes the ISA window by window (a random 4-byte
// store costs a whole 64-byte memory transaction, a binned one is merged in the caches).
template <bool FIRST, bool WRITE_SA, int ISA_MODE, typename KeyT = uint64_t>
__global__ __launch_bounds__(RR_THREADS) void k_rr_apply(
    const KeyT *__restrict__ keys, const uint32_t *__restrict__ V, const uint32_t *__restrict__ U, int64_t m,
    const uint32_t *__restrict__ tile_cnt, const uint32_t *__restrict__ tile_head, uint32_t *__restrict__ SA,
    uint32_t *__restrict__ ISA, uint32_t *__restrict__ Uo, uint32_t *__restrict__ Go, uint32_t *__restrict__ Vo,
    uint32_t n_text, uint32_t *__restrict__ has_isa, int g_shift, uint64_t *__restrict__ pair_k,
    uint32_t *__restrict__ pair_v, const uint32_t *__restrict__ tile_total, int key_shift)
{
    constexpr bool SPARSE = ISA_MODE == 1;
    constexpr int NW = RR_THREADS / WAVE;
    __shared__ uint32_t wcnt[NW], whead[NW];
    if (FIRST && !WRITE_SA && SPARSE) {
        // compaction-only pass (no SA, no ISA write): a tile without tied suffixes has nothing to do
        const uint32_t here = tile_cnt[blockIdx.x];
        const uint32_t next = (blockIdx.x + 1 < gridDim.x) ? tile_cnt[blockIdx.x + 1] : *tile_total;
        if (next == here) return;
    }
    const int l = lane_id(), w = wave_id();
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)w * RR_WAVE_ELEMS;
    const WaveGroups g = rr_wave_classify(keys, m, wbase, key_shift);
    uint32_t slot[RR_ITEMS], v[RR_ITEMS];
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        const bool in = i < m;
        slot[r] = FIRST ? (uint32_t)i : (in ? U[i] : 0u);
        v[r] = in ? V[i] : 0u;
    }
    // this wave's totals -> offsets of the waves of the tile
    {
        uint32_t cnt = 0, lasthead = 0;
#pragma unroll
        for (int r = 0; r < RR_ITEMS; ++r) {
            cnt += (uint32_t)__popcll(g.tied[r]);
            if (g.head[r]) lasthead = (uint32_t)__shfl((int)slot[r], 63 - __builtin_clzll(g.head[r]), WAVE) + 1u;
        }
        if (l == 0) { wcnt[w] = cnt; whead[w] = lasthead; }
    }
    __syncthreads();
    uint32_t run_cnt = tile_cnt[blockIdx.x], run_head = tile_head[blockIdx.x];     // carried in from the tiles before
    for (int ww = 0; ww < w; ++ww) { run_cnt += wcnt[ww]; run_head = run_head > whead[ww] ? run_head : whead[ww]; }
    const uint64_t le_mask = (l == 63) ? ~0ull : ((2ull << l) - 1ull);              // lanes <= l
    const uint64_t lt_mask = le_mask >> 1;                                          // lanes <  l
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        // rank = (slot of the group's first element) + 1: the last head at or before this lane, else the carry
        const uint64_t hle = g.head[r] & le_mask;
        const int src = hle ? 63 - __builtin_clzll(hle) : 0;
        const uint32_t hslot = (uint32_t)__shfl((int)slot[r], src, WAVE);
        const uint32_t run = hle ? hslot + 1u : run_head;
        const uint32_t off = run_cnt + (uint32_t)__popcll(g.tied[r] & lt_mask);
        if (i < m) {
            if (WRITE_SA && slot[r] < n_text) SA[slot[r]] = v[r];
            if (ISA_MODE == 2) {
                pair_k[i] = (uint64_t)v[r];
                pair_v[i] = run;
            } else if (ISA_MODE != 3 && ISA_MODE != 4 && v[r] < n_text) {
                // refinement rounds: the key's high part is the old group head, i.e. the rank already in ISA;
                // the first subgroup of a split group keeps its rank and is not rewritten
                const uint32_t oldrank = (FIRST || ISA_MODE != 0) ? 0u : (uint32_t)((uint64_t)keys[i] >> g_shift) + 1u;
                if (run != oldrank) {
                    ISA[v[r]] = run;
                    if (SPARSE) atomicOr(&has_isa[v[r] >> 5], 1u << (v[r] & 31u));   // this rank overrides the initial one
                }
            }
            if ((g.tied[r] >> l) & 1ull) {
                if (ISA_MODE == 4) {
                    // survivors join those of k_finish_sorted: bitmap + group head by slot (k_surv_compact lists them in slot order)
                    ISA[slot[r]] = run - 1u;
                    atomicOr(&has_isa[slot[r] >> 5], 1u << (slot[r] & 31u));
                    atomicAdd(&pair_v[slot[r] / RR_TILE], 1u);
                } else {
                    Uo[off] = slot[r]; Go[off] = run - 1u; Vo[off] = v[r];
                }
            }
        }
        run_cnt += (uint32_t)__popcll(g.tied[r]);
        if (g.head[r]) run_head = (uint32_t)__shfl((int)slot[r], 63 - __builtin_clzll(g.head[r]), WAVE) + 1u;
    }
}

// ------------------------------------------------------------------------------------------
// Secondary key of prefix doubling.  For suffix v with offset h:
//   v + h <  n : n + ISA[v + h]     (rank of the suffix h symbols further on; ranks start at 1)
//   v + h >= n : n - 1 - v          (text ended inside the compared prefix: the shorter suffix,
//                                    i.e. the larger v, is smaller; all below every real rank)
// key = (group head << key2_bits) | key2, so one sort by key refines every group at once
// (text_key2<KS_RANK> below; the sparse variant without an ISA: sparse_key2).
// ------------------------------------------------------------------------------------------
constexpr int GK_THREADS = 256;

// symbol code of text position pos (0 past the end: the same padding the initial keys use)
__device__ __forceinline__ uint64_t code_at(const uint8_t *__restrict__ T, const uint8_t *lcode, int64_t n, int64_t pos)
{
    return pos < n ? (uint64_t)lcode[T[pos]] : 0ull;
}

// packed key of the `nsym` symbols T[p .. p + nsym) (zero codes past the end), most significant first.
// Away from the end of an 8-byte aligned text the bytes come from aligned 8-byte loads, not byte loads.
__device__ __forceinline__ uint64_t text_key(const uint8_t *__restrict__ T, const uint8_t *lcode, const KeyParams &P, int64_t n,
                                             int64_t p, int nsym, bool aligned8)
{
    uint64_t tk = 0;
    if (P.packed) {
        // bit-packed codes: the key is a bit field of the packed text (two aligned big-endian 64-bit words)
        const int64_t bo = p * P.bits;
        const int nb = nsym * P.bits;                                    // <= 64
        if (nb == 0 || p >= n) return 0;
        const uint64_t *W = (const uint64_t *)P.packed + (bo >> 6);
        const uint64_t w0 = __builtin_bswap64(W[0]), w1 = __builtin_bswap64(W[1]);
        const int sh = (int)(bo & 63);
        const uint64_t val = sh ? ((w0 << sh) | (w1 >> (64 - sh))) : w0;
        return val >> (64 - nb);
    }
    if (aligned8 && p + nsym + 16 <= n) {
        const uint64_t *W = (const uint64_t *)(T + (p & ~(int64_t)7));
        const int sh = (int)(p & 7) * 8;
        uint64_t w0 = W[0];
        for (int done = 0, wi = 1; done < nsym; done += 8, ++wi) {
            const uint64_t w1 = W[wi];
            const uint64_t bytes = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
            const int cnt = nsym - done < 8 ? nsym - done : 8;
            for (int i = 0; i < cnt; ++i) {
                const uint64_t cs = (uint64_t)lcode[(bytes >> (8 * i)) & 255u];
                tk = P.bits > 0 ? ((tk << P.bits) | cs) : (tk * P.sigma + cs);
            }
            w0 = w1;
        }
    } else {
        for (int i = 0; i < nsym; ++i) {
            const uint64_t cs = code_at(T, lcode, n, p + i);
            tk = P.bits > 0 ? ((tk << P.bits) | cs) : (tk * P.sigma + cs);
        }
    }
    return tk;
}

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
enum { KS_TEXT = 0, KS_LOWKEY = 1, KS_RANK = 2, KS_SPARSE = 3, KS_PRE = 4 };
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
    if (MODE == KS_RANK) {
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
    for (int64_t j = (int64_t)blockIdx.x * GK_THREADS + threadIdx.x; j < m; j += stride)
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

template <int MODE>
__global__ __launch_bounds__(GS_THREADS) void k_group_sort(const uint32_t *Vin, const uint32_t *__restrict__ G,
                                                            const uint32_t *__restrict__ U, const uint8_t *__restrict__ T, KeyParams P,
                                                            int64_t m, int64_t n, KeySrc K, uint64_t *keys,
                                                            uint32_t *Vout, uint8_t *__restrict__ bigflag, int cap)   // Vout may be Vin
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
    const bool stamping = g_gs_stamp_on != 0 && t == 0;
    unsigned long long t_prev = stamping ? __builtin_amdgcn_s_memtime() : 0ull;
    auto stamp = [&](int phase) {
        if (stamping) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            atomicAdd(&g_phase_cycles[8 + phase], now - t_prev);
            t_prev = now;
        }
    };
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
        s_key[jl] = packed ? ((key[r] << 11) | (uint64_t)jl) : key[r];
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
#pragma unroll
    for (int r = 0; r < GS_ITEMS; ++r) {
        const int jl = r * GS_THREADS + t;
        const bool valid = base + jl < m;
        const int start = jl - (int)(u[r] - g[r]);               // negative: the group starts before the tile
        // end of the group = next group start after jl (GS_TILE: the tile ends with the group)
        int end;
        {
            const int wi = jl >> 6;
            const uint64_t wbits = (jl & 63) == 63 ? 0ull : (s_head[wi] & (~0ull << ((jl & 63) + 1)));
            end = wbits ? wi * 64 + __builtin_ctzll(wbits) : s_nextH[wi + 1];
        }
        const bool owned = valid && start >= 0 && end >= 0 && end - start <= cap;
        int rank = 0;
        if (owned && packed) {
            const uint64_t mine = (key[r] << 11) | (uint64_t)jl;
            for (int i = start; i < end; ++i) rank += s_key[i] < mine ? 1 : 0;
        } else if (owned) {
            const uint64_t mine = key[r];
            for (int i = start; i < end; ++i) {
                const uint64_t k = s_key[i];
                rank += (k < mine || (k == mine && i < jl)) ? 1 : 0;
            }
        }
        dest[r] = owned ? start + rank : jl;
        big[r] = valid && !owned;
    }
    stamp(3);      // group extents + rank loops
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
__global__ __launch_bounds__(GX_THREADS) void k_group_sort_straddle(uint64_t *__restrict__ keys, uint32_t *__restrict__ V,
                                                                     const uint32_t *__restrict__ G, const uint32_t *__restrict__ U,
                                                                     int64_t m, uint8_t *__restrict__ bigflag, int cap)
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
    uint64_t key[GX_ITEMS]; uint32_t v[GX_ITEMS];
#pragma unroll
    for (int r = 0; r < GX_ITEMS; ++r) {
        const int i = r * GX_THREADS + t;
        key[r] = 0; v[r] = 0;
        if (i < size) { key[r] = keys[start + i]; v[r] = V[start + i]; s_key[i] = key[r]; }
    }
    __syncthreads();
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
__global__ __launch_bounds__(FT_THREADS) void k_finish_sorted(const KeyT *__restrict__ skeys, uint32_t *SA, const uint8_t *__restrict__ T,
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
        if (s_cnt[1]) atomicAdd(&counters[2], s_cnt[1]);
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
__global__ __launch_bounds__(RR_THREADS) void k_flag_count(const uint8_t *__restrict__ flag, int64_t m, uint32_t *__restrict__ tile_cnt)
{
    __shared__ uint32_t lds[RR_THREADS / WAVE + 1];
    const int64_t idx0 = (int64_t)blockIdx.x * RR_TILE + (int64_t)threadIdx.x * RR_ITEMS;
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) if (idx0 + r < m) c += flag[idx0 + r];
    uint32_t tot;
    block_excl_sum<RR_THREADS>(c, lds, &tot);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}

__global__ __launch_bounds__(RR_THREADS) void k_flag_gather(const uint8_t *__restrict__ flag, const uint64_t *__restrict__ keys,
                                                             const uint32_t *__restrict__ V, int64_t m, const uint32_t *__restrict__ tile_cnt,
                                                             uint64_t *__restrict__ bk, uint32_t *__restrict__ bv, uint32_t *__restrict__ bidx)
{
    __shared__ uint32_t lds[RR_THREADS / WAVE + 1];
    const int64_t idx0 = (int64_t)blockIdx.x * RR_TILE + (int64_t)threadIdx.x * RR_ITEMS;
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) if (idx0 + r < m) c += flag[idx0 + r];
    uint32_t tot;
    uint32_t off = tile_cnt[blockIdx.x] + block_excl_sum<RR_THREADS>(c, lds, &tot);
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t j = idx0 + r;
        if (j < m && flag[j]) { bk[off] = keys[j]; bv[off] = V[j]; bidx[off] = (uint32_t)j; ++off; }
    }
}

// sorted flagged elements back to their list positions (sorted by group first, and bidx is increasing,
// so the o-th sorted element belongs at the o-th flagged position)
__global__ __launch_bounds__(256) void k_scatter_back(const uint64_t *__restrict__ bk, const uint32_t *__restrict__ bv,
                                                       const uint32_t *__restrict__ bidx, int64_t count, uint64_t *__restrict__ keys,
                                                       uint32_t *__restrict__ V)
{
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o < count) { const uint32_t j = bidx[o]; keys[j] = bk[o]; V[j] = bv[o]; }
}

// ---- entropy probe + two-stage initial sort ---------------------------------------------------
// When the top 32 bits of the packed key already separate almost all suffixes (random bytes, DNA),
// the initial sort only needs those 4 digits; the few ties are finished by a refinement round on the
// low key bits (k_group_sort).  Whether that holds is measured, not assumed: the keys of
// SAMPLE pseudo-random suffixes are sorted and their duplicates counted (a word-structured text looks
// harmless under an iid model but is not).
__global__ __launch_bounds__(GK_THREADS) void k_sample_keys(const uint8_t *__restrict__ T, KeyParams P, int64_t n, int64_t samples,
                                                             int top_shift, uint64_t *__restrict__ out)
{
    __shared__ uint8_t lcode[256];
    lcode[threadIdx.x] = P.code[threadIdx.x];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * GK_THREADS + threadIdx.x;
    if (i >= samples) return;
    const uint64_t r = ((uint64_t)i + 1u) * 0x9E3779B97F4A7C15ull;
    const int64_t p = (int64_t)((r >> 11) % (uint64_t)n);
    const uint64_t kq = text_key(T, lcode, P, n, p, P.k, (((uintptr_t)T) & 7) == 0);
    out[i] = ((kq >> top_shift) << 32) | (uint64_t)(uint32_t)p;      // position in the low half: equal positions are not collisions
}

// Duplicates of the top 32 key bits among the samples, without sorting them: every sample ((top bits << 32) | position)
// is inserted into an open-addressing hash table of 64-bit entries (all ones = empty).  Meeting an entry with the
// same top bits and another position counts one duplicate (a value seen c times counts c - 1, as adjacent equal
// neighbours of a sorted sample would); the same position drawn twice is not a collision.
__global__ __launch_bounds__(256) void k_count_sample_dups(const uint64_t *__restrict__ samples, int64_t count, unsigned long long *table,
                                                            uint32_t table_mask, uint32_t *__restrict__ dups)
{
    __shared__ uint32_t wsum[256 / WAVE];
    uint32_t c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const unsigned long long e = samples[i];
        const uint32_t k = (uint32_t)(e >> 32);
        uint32_t h = (k * 2654435761u) >> 7;
        for (;;) {
            h &= table_mask;
            const unsigned long long old = atomicCAS(&table[h], ~0ull, e);
            if (old == ~0ull) break;                                           // inserted
            if ((uint32_t)(old >> 32) == k) { c += old != e ? 1u : 0u; break; }
            ++h;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, WAVE);
    if (lane_id() == 0) wsum[wave_id()] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 256 / WAVE; ++w) t += wsum[w];
        if (t) atomicAdd(dups, t);
    }
}

// Dense fallback after text-keyed rounds that left many suffixes tied (repetitive text): every
// resolved suffix has rank = its slot + 1, the tied ones the slot of their group head + 1.
__global__ __launch_bounds__(256) void k_isa_from_sa(const uint32_t *__restrict__ SA, uint32_t *__restrict__ ISA, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint32_t v = SA[i];
        if ((int64_t)v < n) ISA[v] = (uint32_t)i + 1u;
    }
}

__global__ __launch_bounds__(256) void k_isa_tied(const uint32_t *__restrict__ V, const uint32_t *__restrict__ G,
                                                   uint32_t *__restrict__ ISA, int64_t m, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += stride) {
        const uint32_t v = V[j];
        if ((int64_t)v < n) ISA[v] = G[j] + 1u;
    }
}

// ISA[suffix] = rank for pairs that one radix pass has binned by suffix position: consecutive
// pairs fall into the same few-MiB window of the ISA, so the stores merge in L2 / Infinity Cache.
__global__ __launch_bounds__(256) void k_scatter_pairs(const uint64_t *__restrict__ pk, const uint32_t *__restrict__ pv,
                                                        uint32_t *__restrict__ ISA, int64_t count, uint32_t n_text)
{
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + r;
        if (i < count) {
            const uint32_t v = (uint32_t)pk[i];
            if (v < n_text) ISA[v] = pv[i];
        }
    }
}

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }

__global__ __launch_bounds__(256) void k_copy_u32(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

}  // namespace sa
