// kernels/keys.hpp -- which byte values occur, packed-symbol keys of the suffixes, text-key reads, entropy probe.
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
#pragma once
#include "common.hpp"

namespace sa {

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

// The same read with COUNTS (texts of at most a few MiB, where the launch, not the LDS atomics, is the cost): the host also
// wants to know whether the byte values are evenly used (see DeviceBuild::geometry_and_probes: "flat" texts sort narrower keys).
__global__ __launch_bounds__(BH_THREADS) void k_byte_hist_counts(const uint8_t *__restrict__ T, int64_t n, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[BH_THREADS / WAVE][256];
    for (int i = threadIdx.x; i < (BH_THREADS / WAVE) * 256; i += BH_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[wave_id()];
    const uintptr_t addr = (uintptr_t)T;
    int64_t head = (int64_t)((16 - (addr & 15)) & 15);
    if (head > n) head = n;
    const int64_t nvec = (n - head) / 16;
    const uint4 *V = (const uint4 *)(T + head);
    const int64_t gtid = (int64_t)blockIdx.x * BH_THREADS + threadIdx.x;
    const int64_t gstride = (int64_t)gridDim.x * BH_THREADS;
    for (int64_t i = gtid; i < nvec; i += gstride) {
        const uint4 q = V[i];
        const uint32_t w4[4] = { q.x, q.y, q.z, q.w };
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
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < BH_THREADS / WAVE; ++w) c += h[w][threadIdx.x];
    if (c) atomicAdd(&hist[threadIdx.x], c);
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
    // Gram keys (gram > 0; then bits == 0 and k = gram * gram_m): the key is a base-D number whose digits are the dense
    // ranks of the text's g-grams among the g-grams that OCCUR in the text (gram_table[index of the gram as a base-sigma
    // number]; D = how many occur).  Order-preserving like the base-sigma form, but a word-structured text uses a small
    // part of sigma^g: English-like sigma = 57 has 2.5e5 of 1.06e7 four-grams, so three of them (12 symbols) fit in 54
    // bits -- two symbols more than 57^10 < 2^64 allows, in one radix pass less.
    // gram_tail > 0: behind the gram_m grams the key carries gram_tail further symbols as plain base-sigma digits (the bits a
    // whole further gram does not fit into: C3, 3 four-grams in 54.3 bits + 1 symbol of 5.8 = 13 symbols in 61 bits); then
    // k = gram * gram_m + gram_tail.
    int32_t gram, gram_m, gram_tail;
    uint32_t gram_top;       // sigma^(gram - 1)
    uint64_t gram_D;
    const uint4 *gram_table; // one 16-byte record per 64 gram indices: {bitmap of the grams that occur (2 words), how many occur below, 0}
};

// dense rank of gram `idx` among the grams that occur: a rank directory (2.6 MB for 57^4 indices: cache-resident, where a table
// of one word per index would spread the 2.8e5 grams in use over 36 MB of cache lines)
__device__ __forceinline__ uint32_t gram_rank(const uint4 *__restrict__ table, uint32_t idx)
{
    const uint4 rec = table[idx >> 6];
    const uint64_t bits = ((uint64_t)rec.y << 32) | rec.x;
    return rec.z + (uint32_t)__popcll(bits & ((1ull << (idx & 63u)) - 1ull));
}

// the staged text tile of k_build_keys / k_gram_mark: KB_TILE + KB_HALO symbol codes from text position `base` on, zero past the end
__device__ __forceinline__ void kb_stage_codes(const uint8_t *__restrict__ T, int64_t n, int64_t base, const uint8_t *lcode, uint8_t *c, int tid);

// TOP32: only the top 32 bits of every key are stored (keys32), for the two-stage initial sort
// A workgroup takes KB_TPW consecutive tiles.  counts != nullptr: it also histograms the digit ((key >> dshift) & dmask) of the
// keys it writes -- the digit of the first radix pass -- and adds it to that pass's spine (counts[d * G + chunk], zeroed by
// the host; chunk_elems is a multiple of KB_TILE), so the first k_radix_upsweep[32] launch and its read of every key go away.
constexpr int KB_TPW = 8;
constexpr int GM_LOG = 13;
constexpr int GRAM_CACHE = 1 << GM_LOG;     // direct-mapped per-workgroup cache of gram indices already marked (k_gram_mark)
constexpr int GM_TPW = 64;                  // tiles per workgroup there: the cache is warm after the first few.  Measured at 256 MiB
                                            // (cache entries, tiles): (4096, 8) 809 us, (8192, 16) 761, (8192, 64) 718, (16384, 64) 1056;
                                            // without any flag access the kernel takes 217 us -- the misses' flag reads are the rest
__device__ __forceinline__ uint32_t gram_slot(uint32_t idx) { return (idx * 2654435761u) >> (32 - GM_LOG); }

template <bool TOP32, bool GRAM = false>
__global__ __launch_bounds__(KB_THREADS) void k_build_keys(const uint8_t *__restrict__ T, int64_t n,
                                                            KeyParams P, uint64_t *__restrict__ keys,
                                                            uint32_t *__restrict__ vals, uint32_t *__restrict__ keys32,
                                                            int top_shift, uint8_t *__restrict__ packed_out,
                                                            uint32_t *__restrict__ counts, int64_t chunk_elems, int G, uint32_t dmask, int dshift = 0)
{
    __shared__ uint8_t lcode[256];
    __shared__ __attribute__((aligned(16))) uint8_t c[KB_TILE + KB_HALO];
    __shared__ uint32_t dhist[512];            // (dmask <= 511: eight- or nine-bit first digit)
    __shared__ uint32_t gc[GRAM ? KB_TILE + KB_HALO : 1];
    const int tid = threadIdx.x;
    lcode[tid] = P.code[tid];
    for (int d = tid; d < 512; d += KB_THREADS) dhist[d] = 0;

    __syncthreads();
    const int64_t tiles = (n + KB_TILE - 1) / KB_TILE;
    const int64_t tile0 = (int64_t)blockIdx.x * KB_TPW;
    const int64_t tile1 = tile0 + KB_TPW < tiles ? tile0 + KB_TPW : tiles;
  for (int64_t tile = tile0; tile < tile1; ++tile) {
    const int64_t base = tile * KB_TILE;
    kb_stage_codes(T, n, base, lcode, c, tid);
    __syncthreads();
    const int p0 = tid * KB_ITEMS;
    const int k = P.k, bits = P.bits;
    uint64_t kk[KB_ITEMS];
    if (GRAM) {
        // dense rank of the gram at every staged position (one table look-up each), then key(i) = sum rank(i + g j) D^(m-1-j)
        const int g = P.gram, m = P.gram_m;
        const uint32_t sg = (uint32_t)P.sigma;
        for (int o = tid; o + g <= KB_TILE + KB_HALO; o += KB_THREADS) {
            uint32_t idx = 0;
            for (int t = 0; t < g; ++t) idx = idx * sg + (uint32_t)c[o + t];
            gc[o] = gram_rank(P.gram_table, idx);
        }
        __syncthreads();
        const uint64_t D = P.gram_D;
        const int tail = P.gram_tail;
#pragma unroll
        for (int r = 0; r < KB_ITEMS; ++r) {
            uint64_t key = 0;
            for (int j = 0; j < m; ++j) key = key * D + (uint64_t)gc[p0 + r + g * j];
            for (int t = 0; t < tail; ++t) key = key * (uint64_t)sg + (uint64_t)c[p0 + r + g * m + t];
            kk[r] = key;
        }
    } else if (bits > 0) {
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
    if (counts) {
#pragma unroll
        for (int r = 0; r < KB_ITEMS; ++r) {
            const bool in = g0 + r < n;
            const uint32_t d = (uint32_t)((TOP32 ? (kk[r] >> top_shift) : kk[r]) >> dshift) & dmask;
            const uint64_t act = __ballot(in);
            const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
            if (__all(!in || d == f)) {                       // one digit for the whole wave (runs, tiny alphabets): one add, not 64 on one address
                if (act && lane_id() == __ffsll((unsigned long long)act) - 1) atomicAdd(&dhist[d], (uint32_t)__popcll(act));
            } else if (in) atomicAdd(&dhist[d], 1u);
        }
    }
    __syncthreads();                                          // the staged codes are free; this tile's digits are counted
    if (counts) {
        const int64_t chunk = base / chunk_elems;
        const bool flush = tile + 1 == tile1 || (base + KB_TILE) / chunk_elems != chunk;     // (uniform)
        if (flush) {
            for (int d = tid; d <= (int)dmask; d += KB_THREADS) {
                const uint32_t cnt = dhist[d];
                if (cnt) atomicAdd(&counts[(int64_t)d * G + chunk], cnt);
                dhist[d] = 0;
            }
            __syncthreads();
        }
    }
  }
}

__device__ __forceinline__ void kb_stage_codes(const uint8_t *__restrict__ T, int64_t n, int64_t base, const uint8_t *lcode, uint8_t *c, int tid)
{
    // 8 bytes per thread per step
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
}

// ---- gram keys: which g-grams occur, their dense ranks ----------------------------------------
// k_gram_mark : F[index of the g-gram at p] = 1 for every text position p (zero codes past the end, as the keys pad;
//               the all-zero gram is marked too: key positions behind the text read it).  A workgroup remembers the grams it
//               has handled in a small direct-mapped LDS cache, so most positions of a word-structured text cost no access.
// k_gram_count: flags per 8192-entry tile;  k_rr_scan turns them into exclusive sums and the total D;
// k_gram_table: table[index] = number of marked grams below it = the gram's dense rank.
__global__ __launch_bounds__(KB_THREADS) void k_gram_mark(const uint8_t *__restrict__ T, int64_t n, KeyParams P, int g, uint32_t top,
                                                           uint8_t *__restrict__ F)
{
    __shared__ uint8_t lcode[256];
    __shared__ __attribute__((aligned(16))) uint8_t c[KB_TILE + KB_HALO];
    __shared__ uint32_t s_tag[GRAM_CACHE];     // a word-structured text repeats a few thousand grams over and over: most positions stop here
    const int tid = threadIdx.x;
    lcode[tid] = P.code[tid];
    for (int i = tid; i < GRAM_CACHE; i += KB_THREADS) s_tag[i] = 0xffffffffu;
    __syncthreads();
    const int64_t tiles = (n + KB_TILE - 1) / KB_TILE;
    const int64_t tile0 = (int64_t)blockIdx.x * GM_TPW;
    const int64_t tile1 = tile0 + GM_TPW < tiles ? tile0 + GM_TPW : tiles;
    const uint32_t sg = (uint32_t)P.sigma;
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const int64_t base = tile * KB_TILE;
        kb_stage_codes(T, n, base, lcode, c, tid);
        __syncthreads();
        const int p0 = tid * KB_ITEMS;
        uint32_t idx = 0;
        for (int t = 0; t < g; ++t) idx = idx * sg + (uint32_t)c[p0 + t];
        uint32_t ix[KB_ITEMS];
        bool miss[KB_ITEMS];
#pragma unroll
        for (int r = 0; r < KB_ITEMS; ++r) {
            ix[r] = idx;
            const uint32_t sl = gram_slot(idx);
            miss[r] = base + p0 + r < n && s_tag[sl] != idx;
            if (miss[r]) s_tag[sl] = idx;
            idx = (idx - (uint32_t)c[p0 + r] * top) * sg + (uint32_t)c[p0 + r + g];
        }
        // the flags of the misses are read together (one memory latency per tile, not one per position) and written only where
        // still clear: after the first tiles nearly every gram is marked already
        uint8_t f[KB_ITEMS];
#pragma unroll
        for (int r = 0; r < KB_ITEMS; ++r) f[r] = miss[r] ? F[ix[r]] : (uint8_t)1;
#pragma unroll
        for (int r = 0; r < KB_ITEMS; ++r) if (miss[r] && f[r] == 0) F[ix[r]] = 1;
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid == 0) F[0] = 1;
}

constexpr int GT_THREADS = 1024;
constexpr int GT_TILE = GT_THREADS * 64;         // 65536 flags per workgroup, 64 consecutive ones (one bitmap word) per thread
__device__ __forceinline__ uint64_t gram_word(const uint8_t *__restrict__ F, int64_t S, int64_t i0)
{
    uint64_t w = 0;
    if (i0 + 64 <= S) {                           // (F is 64-byte aligned: the flag array starts on a 256-byte boundary)
        const uint4 *q = (const uint4 *)(F + i0);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const uint4 v = q[a];
            const uint32_t ww[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int b = 0; b < 16; ++b) w |= (uint64_t)((ww[b >> 2] >> (8 * (b & 3))) & 1u) << (16 * a + b);
        }
    } else {
        for (int j = 0; j < 64; ++j) if (i0 + j < S && F[i0 + j]) w |= 1ull << j;
    }
    return w;
}

__global__ __launch_bounds__(GT_THREADS) void k_gram_count(const uint8_t *__restrict__ F, int64_t S, uint32_t *__restrict__ tile_cnt)
{
    __shared__ uint32_t lds[GT_THREADS / WAVE];
    const int64_t i0 = (int64_t)blockIdx.x * GT_TILE + (int64_t)threadIdx.x * 64;
    const uint32_t c = i0 < S ? (uint32_t)__popcll(gram_word(F, S, i0)) : 0u;
    uint32_t tot;
    (void)block_excl_sum<GT_THREADS>(c, lds, &tot);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}

__global__ __launch_bounds__(GT_THREADS) void k_gram_table(const uint8_t *__restrict__ F, int64_t S, const uint32_t *__restrict__ tile_off,
                                                            uint4 *__restrict__ table)
{
    __shared__ uint32_t lds[GT_THREADS / WAVE];
    const int64_t i0 = (int64_t)blockIdx.x * GT_TILE + (int64_t)threadIdx.x * 64;
    const uint64_t w = i0 < S ? gram_word(F, S, i0) : 0ull;
    uint32_t tot;
    const uint32_t off = tile_off[blockIdx.x] + block_excl_sum<GT_THREADS>((uint32_t)__popcll(w), lds, &tot);
    if (i0 < S) table[i0 >> 6] = make_uint4((uint32_t)w, (uint32_t)(w >> 32), off, 0u);
}

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
    if (P.gram > 0 && nsym == P.k) {
        // gram keys: the same base-D number k_build_keys builds (rank look-ups; only the rare re-computations take this path)
        const uint32_t sg = (uint32_t)P.sigma;
        for (int j = 0; j < P.gram_m; ++j) {
            uint32_t idx = 0;
            for (int t = 0; t < P.gram; ++t) idx = idx * sg + (uint32_t)code_at(T, lcode, n, p + (int64_t)P.gram * j + t);
            tk = tk * P.gram_D + (uint64_t)gram_rank(P.gram_table, idx);
        }
        for (int t = 0; t < P.gram_tail; ++t) tk = tk * P.sigma + code_at(T, lcode, n, p + (int64_t)P.gram * P.gram_m + t);
        return tk;
    }
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

// ---- entropy probe + two-stage initial sort ---------------------------------------------------
// When the top 32 bits of the packed key already separate almost all suffixes (random bytes, DNA),
// the initial sort only needs those 32 bits (two global passes + kernels/bucket_sort.hpp, or four global passes); the few ties are finished by a refinement round on the
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
    uint64_t part = kq >> (top_shift < 0 ? 0 : top_shift);
    if (top_shift < 0) {                                              // the WHOLE key, hashed to 32 bits (the host subtracts the chance collisions)
        part ^= part >> 33; part *= 0xff51afd7ed558ccdull; part ^= part >> 33; part *= 0xc4ceb9fe1a85ec53ull; part ^= part >> 33;
    }
    out[i] = (part << 32) | (uint64_t)(uint32_t)p;                    // position in the low half: equal positions are not collisions
}

// Repeat probe: how much of the text lies in repeats longer than a few words?  The first 2k symbols of SAMPLE pseudo-random
// suffixes are hashed to 32 bits; duplicates among the samples (k_count_sample_dups) estimate the fraction of suffixes
// that share 2k symbols with another one.  The host routes texts with many such suffixes (a corpus with copied passages)
// straight to rank doubling, texts without (iid words, random bytes) through the text-keyed rounds.
__global__ __launch_bounds__(GK_THREADS) void k_sample_repeat_keys(const uint8_t *__restrict__ T, KeyParams P, int64_t n, int64_t samples,
                                                                    uint64_t *__restrict__ out)
{
    __shared__ uint8_t lcode[256];
    lcode[threadIdx.x] = P.code[threadIdx.x];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * GK_THREADS + threadIdx.x;
    if (i >= samples) return;
    const uint64_t r = ((uint64_t)i + 1u) * 0xD1B54A32D192ED03ull;
    const int64_t p = (int64_t)((r >> 11) % (uint64_t)n);
    const bool al = (((uintptr_t)T) & 7) == 0;
    uint64_t h = text_key(T, lcode, P, n, p, P.k, al) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h += text_key(T, lcode, P, n, p + P.k, P.k, al) * 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    h *= 0x94D049BB133111EBull;
    out[i] = (h & 0xffffffff00000000ull) | (uint64_t)(uint32_t)p;
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
            // (a look first: a text that is one run or one period hashes every sample to the same few entries, and a million
            // compare-and-swaps on one address take 8 ms; a stale "empty" only means the swap is tried after all)
            unsigned long long old = __atomic_load_n(&table[h], __ATOMIC_RELAXED);
            if (old == ~0ull) old = atomicCAS(&table[h], ~0ull, e);
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

}  // namespace sa
