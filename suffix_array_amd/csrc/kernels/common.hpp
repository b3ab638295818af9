// kernels/common.hpp -- wave / block scans, LDS-only barrier, global->LDS DMA helper.
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
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

}  // namespace sa
