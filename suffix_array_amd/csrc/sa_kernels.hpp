// sa_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for suffix-array
// construction.  Pure integer / indexing work: every kernel here is HBM- or latency-bound,
// none is GEMM-shaped, so there is no MFMA anywhere.
//
// Pipeline, replacing the arithmetic behind reference src/saca.rs:14:
//   k_byte_hist      sigma=256 histogram of the text, per-wave counters in LDS
//   k_build_keys     packed-symbol 64-bit key of every suffix, text tile staged in LDS
//   radix sort       stable LSD, 8-bit digits: k_radix_upsweep / k_radix_spine / k_radix_downsweep
//   k_rr_count / k_rr_scan / k_rr_apply
//                    group heads -> ranks (ISA scatter), SA write-back, compaction of the
//                    suffixes still tied with a neighbour
//   k_gather_key2    prefix-doubling secondary key ISA[i+h] with the end-of-text rule
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
// k_byte_hist: sigma = 256 histogram, one private 256-counter table per wave in LDS,
// merged into the global table with one atomic per (block, byte value).
// Algorithmic traffic: 1 B read per input byte.
// ------------------------------------------------------------------------------------------
constexpr int BH_THREADS = 256;

__global__ __launch_bounds__(BH_THREADS) void k_byte_hist(const uint8_t *__restrict__ T, int64_t n,
                                                           uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[BH_THREADS / WAVE][256];
    for (int i = threadIdx.x; i < (BH_THREADS / WAVE) * 256; i += BH_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[wave_id()];
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
            for (int b = 0; b < 4; ++b) atomicAdd(&mine[(w4[a] >> (8 * b)) & 255u], 1u);
        }
    }
    if (blockIdx.x == 0) {
        for (int64_t i = threadIdx.x; i < head; i += BH_THREADS) atomicAdd(&mine[T[i]], 1u);
        for (int64_t i = head + nvec * 16 + threadIdx.x; i < n; i += BH_THREADS) atomicAdd(&mine[T[i]], 1u);
    }
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < BH_THREADS / WAVE; ++w) s += h[w][threadIdx.x];
    if (s) atomicAdd(&hist[threadIdx.x], s);
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
    int32_t bits;
    int32_t k;
    uint64_t mask;
};

__global__ __launch_bounds__(KB_THREADS) void k_build_keys(const uint8_t *__restrict__ T, int64_t n,
                                                            KeyParams P, uint64_t *__restrict__ keys,
                                                            uint32_t *__restrict__ vals)
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
    const uint64_t mask = P.mask;
    uint64_t key = 0;
    for (int j = 0; j < k; ++j) key = (key << bits) | (uint64_t)c[p0 + j];
    uint64_t kk[KB_ITEMS];
    kk[0] = key;
#pragma unroll
    for (int r = 1; r < KB_ITEMS; ++r) {
        key = ((key << bits) | (uint64_t)c[p0 + k - 1 + r]) & mask;
        kk[r] = key;
    }
    const int64_t g0 = base + p0;
    if (g0 + KB_ITEMS <= n) {
        ulonglong2 *ko = (ulonglong2 *)(keys + g0);
#pragma unroll
        for (int r = 0; r < KB_ITEMS / 2; ++r) ko[r] = make_ulonglong2(kk[2 * r], kk[2 * r + 1]);
        uint4 *vo = (uint4 *)(vals + g0);
        const uint32_t v0 = (uint32_t)g0;
        vo[0] = make_uint4(v0, v0 + 1, v0 + 2, v0 + 3);
        vo[1] = make_uint4(v0 + 4, v0 + 5, v0 + 6, v0 + 7);
    } else {
#pragma unroll
        for (int r = 0; r < KB_ITEMS; ++r)
            if (g0 + r < n) { keys[g0 + r] = kk[r]; vals[g0 + r] = (uint32_t)(g0 + r); }
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
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;    // 4096
constexpr int SORT_WAVE_ELEMS = WAVE * SORT_ITEMS;      // 1024

__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift, uint32_t dmask)
{
    return (uint32_t)(key >> shift) & dmask;
}

__global__ __launch_bounds__(SORT_THREADS) void k_radix_upsweep(const uint64_t *__restrict__ keys,
                                                                 uint32_t *__restrict__ counts, int64_t n,
                                                                 int shift, uint32_t dmask,
                                                                 int64_t tiles_per_wg, int G)
{
    __shared__ uint32_t h[SORT_WAVES][RADIX];
    for (int i = threadIdx.x; i < SORT_WAVES * RADIX; i += SORT_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[wave_id()];
    const int64_t begin = (int64_t)blockIdx.x * tiles_per_wg * SORT_TILE;
    int64_t end = begin + tiles_per_wg * SORT_TILE;
    if (end > n) end = n;
    // two keys (16 B) per lane per step; begin is a multiple of SORT_TILE so the pairs are aligned
    const int64_t npair = (end - begin) / 2;
    const ulonglong2 *K2 = (const ulonglong2 *)(keys + begin);
    for (int64_t i = threadIdx.x; i < npair; i += SORT_THREADS) {
        ulonglong2 q = K2[i];
        const uint32_t d0 = digit_of(q.x, shift, dmask), d1 = digit_of(q.y, shift, dmask);
        // constant digits (all-equal high bits) would serialise the LDS atomic 64 ways
        const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)d0);
        const uint32_t active = (uint32_t)__popcll(__ballot(1));   // evaluated by every active lane
        if (__all(d0 == f && d1 == f)) {
            if (lane_id() == 0) atomicAdd(&mine[f], 2u * active);
        } else {
            atomicAdd(&mine[d0], 1u);
            atomicAdd(&mine[d1], 1u);
        }
    }
    if (((end - begin) & 1) && threadIdx.x == 0) atomicAdd(&mine[digit_of(keys[end - 1], shift, dmask)], 1u);
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) s += h[w][threadIdx.x];
    counts[(int64_t)threadIdx.x * G + blockIdx.x] = s;
}

// exclusive scan, in place, of `total` u32 counters by ONE workgroup (digit-major order)
constexpr int SPINE_THREADS = 1024;
__global__ __launch_bounds__(SPINE_THREADS) void k_excl_scan_u32(uint32_t *__restrict__ a, int64_t total,
                                                                  uint32_t *__restrict__ out_total)
{
    __shared__ uint32_t lds[SPINE_THREADS / WAVE + 1];
    const int64_t per = (total + SPINE_THREADS - 1) / SPINE_THREADS;
    int64_t b = (int64_t)threadIdx.x * per, e = b + per;
    if (b > total) b = total;
    if (e > total) e = total;
    uint32_t s = 0;
    for (int64_t i = b; i < e; ++i) s += a[i];
    uint32_t tot;
    uint32_t off = block_excl_sum<SPINE_THREADS>(s, lds, &tot);
    for (int64_t i = b; i < e; ++i) { uint32_t v = a[i]; a[i] = off; off += v; }
    if (out_total && threadIdx.x == 0) *out_total = tot;
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

template <bool FULL>
__device__ __forceinline__ void sort_tile(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                          uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                          int64_t base, int valid, int shift, uint32_t dmask,
                                          uint64_t *lds_kv, uint32_t (*wave_hist)[RADIX], uint32_t *digit_base,
                                          uint32_t *run_off, uint32_t *scan_lds)
{
    const int tid = threadIdx.x, l = lane_id(), w = wave_id();
    const uint64_t lt_mask = (1ull << l) - 1ull;
    uint64_t key[SORT_ITEMS];
    uint32_t val[SORT_ITEMS];
    uint32_t pos[SORT_ITEMS];
    // wave-striped loads: element e = w * 1024 + j * 64 + l
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const int e = w * SORT_WAVE_ELEMS + j * WAVE + l;
        if (FULL || e < valid) { key[j] = keys_in[base + e]; val[j] = vals_in[base + e]; }
        else { key[j] = ~0ull; val[j] = 0; }
    }
    for (int i = tid; i < SORT_WAVES * RADIX; i += SORT_THREADS) (&wave_hist[0][0])[i] = 0;
    __syncthreads();
    // rank inside the wave: lanes with the same digit are found with 8 ballots
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const int e = w * SORT_WAVE_ELEMS + j * WAVE + l;
        const bool ok = FULL || e < valid;
        const uint32_t d = digit_of(key[j], shift, dmask);
        uint64_t m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t prior = wave_hist[w][d];
        const uint32_t below = (uint32_t)__popcll(m & lt_mask);
        if (ok && below == 0) wave_hist[w][d] = prior + (uint32_t)__popcll(m);
        pos[j] = prior + below;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // thread d: turn the per-wave counts of digit d into per-wave offsets, scan the digit totals
    uint32_t tot = 0;
    {
#pragma unroll
        for (int ww = 0; ww < SORT_WAVES; ++ww) {
            const uint32_t cnt = wave_hist[ww][tid];
            wave_hist[ww][tid] = tot;
            tot += cnt;
        }
    }
    uint32_t tile_total;
    const uint32_t dbase = block_excl_sum<SORT_THREADS>(tot, scan_lds, &tile_total);
    digit_base[tid] = dbase;
    __syncthreads();
    // scatter keys into LDS in sorted order
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const int e = w * SORT_WAVE_ELEMS + j * WAVE + l;
        const uint32_t d = digit_of(key[j], shift, dmask);
        pos[j] += digit_base[d] + wave_hist[w][d];
        if (FULL || e < valid) lds_kv[pos[j]] = key[j];
    }
    __syncthreads();
    uint32_t gpos[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const int idx = tid + j * SORT_THREADS;
        if (FULL || idx < valid) {
            const uint64_t kx = lds_kv[idx];
            const uint32_t d = digit_of(kx, shift, dmask);
            gpos[j] = run_off[d] + ((uint32_t)idx - digit_base[d]);
            keys_out[gpos[j]] = kx;
        }
    }
    __syncthreads();
    uint32_t *lds_v = (uint32_t *)lds_kv;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const int e = w * SORT_WAVE_ELEMS + j * WAVE + l;
        if (FULL || e < valid) lds_v[pos[j]] = val[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const int idx = tid + j * SORT_THREADS;
        if (FULL || idx < valid) vals_out[gpos[j]] = lds_v[idx];
    }
    run_off[tid] += tot;     // thread d owns run_off[d]; the next tile starts behind a barrier
    __syncthreads();
}

__global__ __launch_bounds__(SORT_THREADS) void k_radix_downsweep(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint64_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ spine, const uint32_t *__restrict__ digit_tot,
    int64_t n, int shift, uint32_t dmask, int64_t tiles_per_wg, int G)
{
    __shared__ __attribute__((aligned(16))) uint64_t lds_kv[SORT_TILE];
    __shared__ uint32_t wave_hist[SORT_WAVES][RADIX];
    __shared__ uint32_t digit_base[RADIX];
    __shared__ uint32_t run_off[RADIX];
    __shared__ uint32_t scan_lds[SORT_WAVES + 1];
    {
        // global start of digit d = totals of the smaller digits + this workgroup's row prefix
        uint32_t all;
        const uint32_t gbase = block_excl_sum<SORT_THREADS>(digit_tot[threadIdx.x], scan_lds, &all);
        run_off[threadIdx.x] = gbase + spine[(int64_t)threadIdx.x * G + blockIdx.x];
    }
    __syncthreads();
    const int64_t tile0 = (int64_t)blockIdx.x * tiles_per_wg;
    for (int64_t t = 0; t < tiles_per_wg; ++t) {
        const int64_t base = (tile0 + t) * SORT_TILE;
        if (base >= n) break;
        const int64_t rem = n - base;
        if (rem >= SORT_TILE)
            sort_tile<true>(keys_in, vals_in, keys_out, vals_out, base, SORT_TILE, shift, dmask, lds_kv, wave_hist,
                            digit_base, run_off, scan_lds);
        else
            sort_tile<false>(keys_in, vals_in, keys_out, vals_out, base, (int)rem, shift, dmask, lds_kv, wave_hist,
                             digit_base, run_off, scan_lds);
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
constexpr int RR_THREADS = 256;
constexpr int RR_ITEMS = 8;
constexpr int RR_TILE = RR_THREADS * RR_ITEMS;   // 2048

struct RRFlags {
    uint32_t head;   // bit r: element r starts a group
    uint32_t tied;   // bit r: element r is in a group of size > 1
};

// loads the thread's RR_ITEMS keys (blocked) and classifies them; idx0 = first element index
__device__ __forceinline__ RRFlags rr_classify(const uint64_t *__restrict__ keys, int64_t m, int64_t idx0)
{
    uint64_t k[RR_ITEMS + 2];
#pragma unroll
    for (int r = 0; r < RR_ITEMS + 2; ++r) {
        const int64_t i = idx0 - 1 + r;
        k[r] = (i >= 0 && i < m) ? keys[i] : 0;
    }
    RRFlags f; f.head = 0; f.tied = 0;
    uint32_t headx = 0;   // bit r: element idx0 - 1 + r starts a group (r in 1..RR_ITEMS+1)
#pragma unroll
    for (int r = 1; r <= RR_ITEMS + 1; ++r) {
        const int64_t i = idx0 - 1 + r;
        const bool h = (i == 0) || (i >= m) || (k[r] != k[r - 1]);
        headx |= (uint32_t)h << r;
    }
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = idx0 + r;
        if (i < m) {
            const bool h = (headx >> (r + 1)) & 1u, hn = (headx >> (r + 2)) & 1u;
            f.head |= (uint32_t)h << r;
            f.tied |= (uint32_t)(!(h && hn)) << r;
        }
    }
    return f;
}

template <bool FIRST>
__global__ __launch_bounds__(RR_THREADS) void k_rr_count(const uint64_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ U, int64_t m,
                                                          uint32_t *__restrict__ tile_cnt,
                                                          uint32_t *__restrict__ tile_head)
{
    __shared__ uint32_t lds[RR_THREADS / WAVE + 1];
    const int64_t idx0 = (int64_t)blockIdx.x * RR_TILE + (int64_t)threadIdx.x * RR_ITEMS;
    const RRFlags f = rr_classify(keys, m, idx0);
    uint32_t cnt = (uint32_t)__popc(f.tied);
    uint32_t lasthead = 0;
    if (f.head) {
        const int r = 31 - __clz((int)f.head);
        lasthead = (FIRST ? (uint32_t)(idx0 + r) : U[idx0 + r]) + 1u;
    }
    uint32_t tot, mx;
    block_excl_sum<RR_THREADS>(cnt, lds, &tot);
    block_incl_max<RR_THREADS>(lasthead, lds, &mx);
    if (threadIdx.x == 0) { tile_cnt[blockIdx.x] = tot; tile_head[blockIdx.x] = mx; }
}

// one workgroup: tile_cnt -> exclusive sums (+ total), tile_head -> exclusive running max
__global__ __launch_bounds__(SPINE_THREADS) void k_rr_scan(uint32_t *__restrict__ tile_cnt,
                                                            uint32_t *__restrict__ tile_head, int64_t tiles,
                                                            uint32_t *__restrict__ out_total)
{
    __shared__ uint32_t lds[SPINE_THREADS / WAVE + 1];
    const int64_t per = (tiles + SPINE_THREADS - 1) / SPINE_THREADS;
    int64_t b = (int64_t)threadIdx.x * per, e = b + per;
    if (b > tiles) b = tiles;
    if (e > tiles) e = tiles;
    uint32_t s = 0, mx = 0;
    for (int64_t i = b; i < e; ++i) { s += tile_cnt[i]; uint32_t h = tile_head[i]; mx = mx > h ? mx : h; }
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
    for (int64_t i = b; i < e; ++i) {
        uint32_t c = tile_cnt[i], h = tile_head[i];
        tile_cnt[i] = off; off += c;
        tile_head[i] = run; run = run > h ? run : h;
    }
    if (threadIdx.x == 0) *out_total = tot;
}

template <bool FIRST, bool WRITE_SA>
__global__ __launch_bounds__(RR_THREADS) void k_rr_apply(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ V, const uint32_t *__restrict__ U, int64_t m,
    const uint32_t *__restrict__ tile_cnt, const uint32_t *__restrict__ tile_head, uint32_t *__restrict__ SA,
    uint32_t *__restrict__ ISA, uint32_t *__restrict__ Uo, uint32_t *__restrict__ Go, uint32_t *__restrict__ Vo,
    uint32_t n_text)
{
    __shared__ uint32_t lds[RR_THREADS / WAVE + 1];
    const int64_t idx0 = (int64_t)blockIdx.x * RR_TILE + (int64_t)threadIdx.x * RR_ITEMS;
    const RRFlags f = rr_classify(keys, m, idx0);
    uint32_t slot[RR_ITEMS], v[RR_ITEMS];
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = idx0 + r;
        if (i < m) { slot[r] = FIRST ? (uint32_t)i : U[i]; v[r] = V[i]; }
        else { slot[r] = 0; v[r] = 0; }
    }
    // group-head slot (+1): running max over (head ? slot + 1 : 0), seeded by the tiles before
    uint32_t lasthead = 0;
    if (f.head) lasthead = slot[31 - __clz((int)f.head)] + 1u;
    uint32_t dummy;
    uint32_t incm = block_incl_max<RR_THREADS>(lasthead, lds, &dummy);
    uint32_t prevm = __shfl_up(incm, 1, WAVE);
    __shared__ uint32_t wlast[RR_THREADS / WAVE];
    if (lane_id() == WAVE - 1) wlast[wave_id()] = incm;
    __syncthreads();
    if (lane_id() == 0) prevm = wave_id() ? wlast[wave_id() - 1] : 0u;
    const uint32_t carry = tile_head[blockIdx.x];
    uint32_t run = prevm > carry ? prevm : carry;
    uint32_t tot;
    uint32_t off = tile_cnt[blockIdx.x] + block_excl_sum<RR_THREADS>((uint32_t)__popc(f.tied), lds, &tot);
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = idx0 + r;
        if (i < m) {
            if ((f.head >> r) & 1u) run = slot[r] + 1u;
            if (WRITE_SA && slot[r] < n_text) SA[slot[r]] = v[r];
            if (v[r] < n_text) ISA[v[r]] = run;
            if ((f.tied >> r) & 1u) { Uo[off] = slot[r]; Go[off] = run - 1u; Vo[off] = v[r]; ++off; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_gather_key2: secondary key of prefix doubling.  For suffix v with offset h:
//   v + h <  n : n + ISA[v + h]     (rank of the suffix h symbols further on; ranks start at 1)
//   v + h >= n : n - 1 - v          (text ended inside the compared prefix: the shorter suffix,
//                                    i.e. the larger v, is smaller; all below every real rank)
// key = (group head << key2_bits) | key2, so one sort by key refines every group at once.
// Algorithmic traffic: 8 B read + 4 B random ISA read + 8 B written per element.
// ------------------------------------------------------------------------------------------
constexpr int GK_THREADS = 256;
__global__ __launch_bounds__(GK_THREADS) void k_gather_key2(const uint32_t *__restrict__ V,
                                                             const uint32_t *__restrict__ G,
                                                             const uint32_t *__restrict__ ISA, int64_t m, int64_t n,
                                                             int64_t h, int key2_bits, uint64_t *__restrict__ keys)
{
    const int64_t stride = (int64_t)gridDim.x * GK_THREADS;
    for (int64_t j = (int64_t)blockIdx.x * GK_THREADS + threadIdx.x; j < m; j += stride) {
        const uint32_t v = V[j];
        const int64_t p = (int64_t)v + h;
        const uint64_t key2 = (p < n) ? (uint64_t)n + (uint64_t)ISA[p] : (uint64_t)(n - 1 - (int64_t)v);
        keys[j] = ((uint64_t)G[j] << key2_bits) | key2;
    }
}

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }

__global__ __launch_bounds__(256) void k_copy_u32(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

}  // namespace sa
