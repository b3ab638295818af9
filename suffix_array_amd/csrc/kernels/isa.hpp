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

// The same after TWO radix passes (pairs sorted by the top 16 bits of the suffix position): a workgroup takes 8192 consecutive
// pairs, which then lie in a few consecutive windows of 2^WLOG ISA entries; window by window it places the ranks in LDS and
// stores the window's touched entries in address order -- whole 64-byte lines when every entry is written (the full build),
// one partial write per line otherwise -- instead of one read-modify-write of a 64-byte HBM burst per pair.
constexpr int SW_THREADS = 1024;
constexpr int SW_ITEMS = 8;
constexpr int SW_CHUNK = SW_THREADS * SW_ITEMS;

template <int WLOG_MAX>
__global__ __launch_bounds__(SW_THREADS) void k_scatter_windows(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pv,
                                                                 uint32_t *__restrict__ ISA, int64_t count, uint32_t n_text, int wlog)
{
    __shared__ uint32_t s_rank[1 << WLOG_MAX];
    __shared__ uint32_t s_bits[(1 << WLOG_MAX) / 32 + 1];
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * SW_CHUNK;
    const int64_t c1 = c0 + SW_CHUNK < count ? c0 + SW_CHUNK : count;
    if (c0 >= count) return;
    uint32_t v[SW_ITEMS], r[SW_ITEMS];
#pragma unroll
    for (int k = 0; k < SW_ITEMS; ++k) {
        const int64_t i = c0 + k * SW_THREADS + t;
        v[k] = i < c1 ? pk[i] : 0xffffffffu;
        r[k] = i < c1 ? pv[i] : 0u;
    }
    const uint32_t W = 1u << wlog, wmask = W - 1u;
    const uint32_t wfirst = pk[c0] >> wlog, wlast = pk[c1 - 1] >> wlog;       // (the pairs are sorted by this value)
    for (uint32_t w = wfirst; w <= wlast; ++w) {
        for (uint32_t i = t; i < (W + 31) / 32; i += SW_THREADS) s_bits[i] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SW_ITEMS; ++k) {
            if (v[k] != 0xffffffffu && (v[k] >> wlog) == w && v[k] < n_text) {
                const uint32_t p = v[k] & wmask;
                s_rank[p] = r[k];
                atomicOr(&s_bits[p >> 5], 1u << (p & 31u));
            }
        }
        __syncthreads();
        const uint64_t wbase = (uint64_t)w << wlog;
        for (uint32_t p = t; p < W; p += SW_THREADS)
            if ((s_bits[p >> 5] >> (p & 31u)) & 1u) ISA[wbase + p] = s_rank[p];
        __syncthreads();
        if (w == 0xffffffffu) break;                                           // (cannot wrap: wlast < 2^32 >> wlog)
    }
}

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }

// ---- early download (host/host_path.hpp): the part of the suffix array that starts its way to the host while the last
// refinement rounds still run.  A slot outside the tied list never changes again, so the array can be copied as it stands;
// the slots that WERE still tied when the copy began (the snapshot list U, marked in a bitmap over the entries of the
// downloaded array: entry = slot + off) are sent again when the build is done -- compacted in entry order (k_early_count /
// k_rr_scan / k_early_gather) -- and patched into the caller's array by the host.
constexpr int EARLY_TILE = 8192;          // entries per counting tile: 256 bitmap words
constexpr int EARLY_THREADS = 256;

// (the list is in slot order: a thread takes eight consecutive members, whose bits fall into one or two bitmap words most of the
// time, and issues one atomic per word instead of one per member -- 37 M members of C3: 1.13 -> ~0.3 ms)
__global__ __launch_bounds__(256) void k_early_mark(const uint32_t *__restrict__ U, int64_t m, uint32_t off, uint32_t *__restrict__ bits)
{
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i0 < m; i0 += stride) {
        uint32_t u[8];
        if (i0 + 8 <= m) {
            const uint4 a = *(const uint4 *)(U + i0), b = *(const uint4 *)(U + i0 + 4);
            u[0] = a.x; u[1] = a.y; u[2] = a.z; u[3] = a.w; u[4] = b.x; u[5] = b.y; u[6] = b.z; u[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] = i0 + k < m ? U[i0 + k] : 0xffffffffu;
        }
        uint32_t word = 0xffffffffu, acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (u[k] == 0xffffffffu) continue;
            const uint32_t j = u[k] + off, w = j >> 5;
            if (w != word) {
                if (acc) atomicOr(&bits[word], acc);
                word = w; acc = 0;
            }
            acc |= 1u << (j & 31u);
        }
        if (acc) atomicOr(&bits[word], acc);
    }
}

// marked entries per tile of EARLY_TILE entries (one workgroup per tile, one bitmap word per thread)
__global__ __launch_bounds__(EARLY_THREADS) void k_early_count(const uint32_t *__restrict__ bits, int64_t words, uint32_t *__restrict__ tile_cnt)
{
    __shared__ uint32_t lds[EARLY_THREADS / WAVE + 1];
    const int64_t w = (int64_t)blockIdx.x * EARLY_THREADS + threadIdx.x;
    const uint32_t c = w < words ? (uint32_t)__popc(bits[w]) : 0u;
    uint32_t total;
    (void)block_excl_sum<EARLY_THREADS>(c, lds, &total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

// holes[tile_off[tile] + rank inside the tile] = src[entry] for every marked entry, in entry order
__global__ __launch_bounds__(EARLY_THREADS) void k_early_gather(const uint32_t *__restrict__ bits, int64_t words, const uint32_t *__restrict__ tile_off,
                                                                 const uint32_t *__restrict__ src, uint32_t *__restrict__ holes)
{
    __shared__ uint32_t lds[EARLY_THREADS / WAVE + 1];
    const int64_t w = (int64_t)blockIdx.x * EARLY_THREADS + threadIdx.x;
    uint32_t b = w < words ? bits[w] : 0u;
    uint32_t total;
    uint32_t k = tile_off[blockIdx.x] + block_excl_sum<EARLY_THREADS>((uint32_t)__popc(b), lds, &total);
    while (b) {
        const int bit = __builtin_ctz(b);
        holes[k++] = src[w * 32 + bit];
        b &= b - 1u;
    }
}

// the suffix array of a text that is ONE byte value repeated (a zero-filled file): the shorter suffix is a proper prefix of the
// longer one, so the order is by length -- SA[i] = n - 1 - i, no sorting at all
__global__ __launch_bounds__(256) void k_fill_descending(uint32_t *__restrict__ SA, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) SA[i] = (uint32_t)(n - 1 - i);
}

__global__ __launch_bounds__(256) void k_copy_u32(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

}  // namespace sa
