// kernels/radix_sort_diag.hpp -- DIAGNOSTIC ONLY (-DSA_AMD_DIAG): first-generation tile scatter with timing ablations that produce WRONG orders.
// Never compiled into libsuffix_array_amd.so; libsuffix_array_amd_diag.so uses it for profiles/*ablation*.
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
#pragma once
#include "radix_sort.hpp"

namespace sa {

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

}  // namespace sa
