// kernels/radix_sort.hpp -- digit extraction and the counting kernels of the LSD radix sort (product); the three-kernel pass
// upsweep / spine / carry-completed-line downsweep of rounds 1-2 (diagnostic library only: -DSA_AMD_DIAG).
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
#pragma once
#include "common.hpp"

namespace sa {

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

#ifdef SA_AMD_DIAG
// ---- the three-kernel pass of rounds 1-2 (upsweep / spine / carry-completed-line downsweep): DIAGNOSTIC LIBRARY ONLY since round 4.
// The product sorts with the single-pass tile scatter (kernels/onesweep.hpp) and keeps only the counting kernels above (first pass
// of a sort whose producer did not count).  SA_AMD_NO_ONESWEEP=1 / SA_AMD_SORT_VARIANT select this engine in
// libsuffix_array_amd_diag.so for A/B measurements and for the primitive tests that compare the two engines. ----
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
#ifdef SA_AMD_DIAG
// diagnostic library only (STAMP): cycles of wave 0 per phase, summed over tiles and workgroups
__device__ unsigned long long g_phase_cycles[16];
__device__ int g_gs_stamp_on;          // k_group_sort adds its per-phase cycles (wave 0) to g_phase_cycles[8..15]
#endif

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
    (void)t_prev;
    auto stamp = [&](int phase) {
#ifdef SA_AMD_DIAG
        if (STAMP && tid == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (phase >= 0) atomicAdd(&g_phase_cycles[phase], now - t_prev);
            t_prev = now;
        }
#endif
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

#endif  // SA_AMD_DIAG

}  // namespace sa
