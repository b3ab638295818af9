// kernels/isa.hpp -- inverse suffix array maintenance of the dense prefix-doubling rounds, small utility kernels.
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
#pragma once
#include "common.hpp"

namespace sa {

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
template <typename KeyT>
__global__ __launch_bounds__(256) void k_scatter_pairs(const KeyT *__restrict__ pk, const uint32_t *__restrict__ pv,
                                                        uint32_t *__restrict__ ISA, int64_t count, uint32_t n_text)
{
    const int64_t lb = blockIdx.x;       // (an XCD-aware block order -- one window of the ISA per XCD at a time -- measured SLOWER: 4.1 -> 5.6 ms)
    const int64_t i0 = (lb * 256 + threadIdx.x) * 4;
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
