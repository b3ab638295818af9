// kernels/onesweep.hpp -- single-pass radix tile scatter: every key is read ONCE per pass (no histogram pre-pass).
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
// The stable LSD sort behind the initial sort and the refinement rounds, i.e. part of the arithmetic that replaces
// `cdivsufsort::sort_in_place` (reference src/saca.rs:14).
//
// The three-kernel pass of radix_sort.hpp reads every key twice (histogram, then scatter) because a workgroup must know
// where its chunk's digit runs start before it stores anything.  Here a tile learns that from its predecessors while the
// pass is running (decoupled look-back), and the pass is laid out for the eight L2s of the chip:
//   * The input is cut into at most OS_NSEG SEGMENTS of whole tiles.  How many elements of digit d each segment holds is
//     known BEFORE the pass -- the previous pass counted it while it had the keys in registers (one LDS atomic per key:
//     the destination position tells the segment of the next pass), the producer of the keys counted it for the first pass
//     (k_build_keys) or one k_radix_upsweep launch did -- so every segment knows where its digit runs start.
//   * Inside a segment the tiles are handed out in ticket order.  A workgroup asks for tiles of the segments that belong
//     to ITS XCD first (segment s -> XCD s mod 8, read from HW_REG_XCC_ID), so the tiles in flight next to each other in
//     the input are in flight on the same XCD at the same time: the 128-byte lines at the seams of their digit runs are
//     completed in that XCD's L2 within a microsecond, by two CUs, instead of travelling to memory as two partial
//     lines.  Measured (tools/scatter_probe.hip, 2^28 pairs of 8 + 4 bytes): 3.99 TB/s against 3.22 TB/s with chip-wide
//     tickets and 2.79 TB/s for workgroup-owned chunks without carries.  A workgroup that finds its own segments
//     drained takes tiles of any other segment: placement only changes the speed, never the result.
//   * Look-back: per tile and digit one 8-byte granule {epoch << 2 | flag, value} written by ONE agent-scope store and
//     polled with agent-scope loads (no fences: the data is its own flag, MI355X guide R2).  flag 1 = the tile's own count
//     (published right after the ranking), flag 2 = the inclusive prefix of the segment up to and including the tile.
//     A tile sums its predecessors' counts backwards until it meets an inclusive prefix, three predecessors per round trip;
//     the first look comes after the tile has been staged in LDS, when its neighbours' counts have had time to become visible.
//     A tile only ever waits for tiles with smaller tickets, which are held by running workgroups: no co-residency
//     assumption, no deadlock when another stream shares the device.
// Algorithmic traffic per pass: sizeof(key) + 4 read, the same written, + 16 bytes of granules per 256 / TILE elements.
#pragma once
#include "common.hpp"
#include "radix_sort.hpp"

namespace sa {

constexpr int OS_NSEG = 8;           // segments per pass (one per XCD; fewer when there are fewer tiles)
constexpr int OS_LB = 3;             // predecessors read speculatively per look-back round (measured: 1 slower, 2 and 3 equal, 4 and 8
                                     // slower on 32-bit keys -- rows of granules nobody needs are L2 traffic too; profiles/r03_onesweep_experiments.txt)
// four words from a 4-byte aligned address in one load (global_load_dwordx4 needs dword alignment only; a plain uint4 dereference
// would promise the compiler 16): the text-key readers below round a byte address down to a word and cut their keys out of the
// words with v_alignbyte.  Up to three bytes in front of an unaligned text pointer are read that way (same allocation: a
// caller's dT that is not 4-byte aligned sits inside a block whose start is) -- stated in include/suffix_array_amd.h.
struct __attribute__((packed, aligned(4))) Words4 { uint32_t x, y, z, w; };
constexpr int OS_MIN_TILE = 4096;    // smallest tile of any shape in use (the granule slab is sized by it)

__device__ __forceinline__ unsigned os_xcc_id()
{
    return (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID, bits [3:0]
}

typedef __attribute__((address_space(1))) unsigned long long os_gu64;

// ---- a text of all 256 byte values: the top 32 key bits of suffix i ARE the bytes T[i .. i + 3], big-endian (symbol code =
// byte value, 8 bits per symbol), zero past the end.  The first pass of the 32-bit stage reads them from the text -- a
// quarter of the bytes, and no k_build_keys launch in front (0.45 ms at 256 MiB).  Two aligned words + a byte alignment
// instead of one unaligned load; the last few positions are read bytewise (no access beyond T + n).
__device__ __forceinline__ uint32_t text_key32(const uint8_t *__restrict__ T, int64_t n, int64_t i)
{
    if (i + 8 <= n) {
        const uintptr_t a = (uintptr_t)(T + i);
        const uint32_t *W = (const uint32_t *)(a & ~(uintptr_t)3);
        const uint32_t lo = W[0], hi = W[1];
        const uint32_t le = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(a & 3u));      // bytes T[i .. i + 3], little-endian
        return __builtin_bswap32(le);
    }
    uint32_t k = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) k = (k << 8) | (i + b < n ? (uint32_t)T[i + b] : 0u);
    return k;
}

// counts[d * G + g] += suffixes of chunk g whose key digit (text_key32 >> shift) & dmask is d (the counts of the first pass;
// zeroed by the host); `split` workgroups share a chunk
__global__ __launch_bounds__(SORT_THREADS) void k_text_upsweep32(const uint8_t *__restrict__ T, int64_t n, uint32_t *__restrict__ counts,
                                                                 int shift, uint32_t dmask, int64_t chunk_elems, int G, int split, int64_t sub_elems)
{
    __shared__ uint32_t h[SORT_WAVES][512];
    for (int i = threadIdx.x; i < SORT_WAVES * 512; i += SORT_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[wave_id()];
    const int g = (int)(blockIdx.x / split), part = (int)(blockIdx.x % split);
    const int64_t cbegin = (int64_t)g * chunk_elems;
    int64_t cend = cbegin + chunk_elems;
    if (cend > n) cend = n;
    int64_t begin = cbegin + (int64_t)part * sub_elems;
    int64_t end = begin + sub_elems;
    if (begin > cend) begin = cend;
    if (end > cend || part == split - 1) end = cend;
    // sixteen consecutive positions per thread and step out of six aligned words, while that stays inside the text; the rest one by one
    int64_t i = begin + (int64_t)threadIdx.x * 16;
    for (; i + 16 <= end && i + 32 <= n; i += (int64_t)SORT_THREADS * 16) {
        const uintptr_t a = (uintptr_t)(T + i);
        const uint32_t *W = (const uint32_t *)(a & ~(uintptr_t)3);
        const uint32_t sh = (uint32_t)(a & 3u);
        const Words4 q = *(const Words4 *)W;
        const uint32_t w4 = W[4], w5 = W[5];
        const uint32_t v[6] = { __builtin_amdgcn_alignbyte(q.y, q.x, sh), __builtin_amdgcn_alignbyte(q.z, q.y, sh), __builtin_amdgcn_alignbyte(q.w, q.z, sh),
                                __builtin_amdgcn_alignbyte(w4, q.w, sh), __builtin_amdgcn_alignbyte(w5, w4, sh), 0u };
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t key = __builtin_bswap32(__builtin_amdgcn_alignbyte(v[(j >> 2) + 1], v[j >> 2], (uint32_t)(j & 3)));
            atomicAdd(&mine[(key >> shift) & dmask], 1u);
        }
    }
    // (what the loop above left: a thread's last, partial or too-close-to-the-end blocks of sixteen)
    for (; i < end; i += (int64_t)SORT_THREADS * 16)
        for (int64_t p = i; p < i + 16 && p < end; ++p) atomicAdd(&mine[(text_key32(T, n, p) >> shift) & dmask], 1u);
    __syncthreads();
    for (int d = threadIdx.x; d <= (int)dmask; d += SORT_THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) sum += h[w][d];
        if (sum) atomicAdd(&counts[(int64_t)d * G + g], sum);
    }
}

// ---- the same for an alphabet of four symbols (DNA): the bit-packed text (2 bits per symbol, big-endian bit order, zero
// codes behind the end; k_pack_text2) IS the stream of keys -- the top 32 key bits of suffix i are its bits [2i, 2i + 32).
// k_pack_text2: 16 symbols per thread -> one 32-bit word (stored big-endian), instead of k_build_keys' key array (4 B per
// suffix written and read back: 1.7 ms at 1 GiB)
__global__ __launch_bounds__(256) void k_pack_text2(const uint8_t *__restrict__ T, int64_t n, KeyParams P, uint8_t *__restrict__ packed)
{
    __shared__ uint8_t lcode[256];
    lcode[threadIdx.x] = P.code[threadIdx.x];
    __syncthreads();
    const int64_t words = (n + 15) / 16;
    for (int64_t wi = (int64_t)blockIdx.x * 256 + threadIdx.x; wi < words; wi += (int64_t)gridDim.x * 256) {
        const int64_t p = wi * 16;
        uint32_t v = 0;
        if (p + 16 <= n && (((uintptr_t)(T + p)) & 15) == 0) {
            const uint4 q = *(const uint4 *)(T + p);
            const uint32_t w4[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
            for (int j = 0; j < 16; ++j) v = (v << 2) | (uint32_t)lcode[(w4[j >> 2] >> (8 * (j & 3))) & 255u];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v = (v << 2) | (p + j < n ? (uint32_t)lcode[T[p + j]] : 0u);
        }
        ((uint32_t *)packed)[wi] = __builtin_bswap32(v);         // (the packed text is a byte stream: first symbol in the top bits of byte 0)
    }
}

// 64 stream bits from byte address B on (big-endian), out of three aligned words
__device__ __forceinline__ uint64_t packed_window64(const uint8_t *B)
{
    const uintptr_t a = (uintptr_t)B;
    const uint32_t *W = (const uint32_t *)(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3u);
    const uint32_t w0 = W[0], w1 = W[1], w2 = W[2];
    const uint32_t v0 = __builtin_amdgcn_alignbyte(w1, w0, sh), v1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
    return ((uint64_t)__builtin_bswap32(v0) << 32) | (uint64_t)__builtin_bswap32(v1);
}

// the counts of the first pass's digit, from the packed text (counts[d * G + g], zeroed by the host), sixteen suffixes per thread and step
__global__ __launch_bounds__(SORT_THREADS) void k_packed2_upsweep32(const uint8_t *__restrict__ packed, int64_t n, uint32_t *__restrict__ counts,
                                                                    int shift, uint32_t dmask, int64_t chunk_elems, int G, int split, int64_t sub_elems)
{
    __shared__ uint32_t h[SORT_WAVES][512];
    for (int i = threadIdx.x; i < SORT_WAVES * 512; i += SORT_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[wave_id()];
    const int g = (int)(blockIdx.x / split), part = (int)(blockIdx.x % split);
    const int64_t cbegin = (int64_t)g * chunk_elems;                  // (a multiple of the tile size: of 4)
    int64_t cend = cbegin + chunk_elems;
    if (cend > n) cend = n;
    int64_t begin = cbegin + (int64_t)part * sub_elems;               // (sub_elems is a multiple of 16)
    int64_t end = begin + sub_elems;
    if (begin > cend) begin = cend;
    if (end > cend || part == split - 1) end = cend;
    for (int64_t i = begin + (int64_t)threadIdx.x * 16; i < end; i += (int64_t)SORT_THREADS * 16) {
        const uint64_t win = packed_window64(packed + (i >> 2));      // (the 64 zero bytes behind the packed text cover the last reads)
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (i + j < end) atomicAdd(&mine[((uint32_t)(win >> (32 - 2 * j)) >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d <= (int)dmask; d += SORT_THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) sum += h[w][d];
        if (sum) atomicAdd(&counts[(int64_t)d * G + g], sum);
    }
}

struct OnesweepPass {
    const uint32_t *hist_cur;       // [256][nseg]: elements of digit d in segment s, in this pass's input order
    uint32_t *hist_next;            // [256][nseg]: the same for the NEXT pass's digit (zeroed by the host), or nullptr
    uint32_t *tickets;              // [nseg], zeroed by the host
    unsigned long long *status;     // [tiles][256] granules (zeroed once per sort call; the epoch tells the passes apart)
    uint32_t *err;                  // bumped when a look-back spin gives up (never in a correct run; the host checks)
    int64_t n;
    int shift, shift_next;
    uint32_t dmask, dmask_next;
    int nseg, tiles_per_seg, tiles;
    uint32_t epoch;                 // pass number + 1
    const uint8_t *text;            // TEXT_KEYS instances: the keys are read from here (text_key32), keys_in is not looked at;
    int64_t text_n;                 //   text_bits = 2: `text` is the bit-packed text of a four-symbol alphabet (packed_window64)
    int text_bits;
    int val_extra;                  // TEXT_KEYS: the e key bits BELOW the 32 of the key travel in the top e bits of the value (the index needs only
                                    //   32 - e: host-checked), so that the bucket sort orders 32 + e key bits (kernels/bucket_sort.hpp); e <= 2
    uint32_t flags;                 // bit 0: look at the predecessors' granules before the staging, not after (scheduling A/B, same
                                    // result); bit 7 (diagnostic library only): phase stamps
};

// THREADS x ITEMS elements per tile.  SEQ = false: keys and values are staged in LDS side by side (one workgroup per CU at the
// sizes in use); SEQ = true: the values go through the keys' buffer once the keys are out, which leaves room for TWO
// workgroups per CU -- while one waits (look-back, barriers, the ranking's ALU work) the other one's loads and stores flow.
// RBITS: digit width (8, or 9 for the two global passes in front of the bucket sort of a text of more than 2^29 suffixes)
// TEXT_KEYS: the first pass of the 32-bit stage over a text of all 256 byte values -- key of element i = text_key32(P.text, i)
template <int THREADS, int ITEMS, typename KeyT, bool SEQ, int WG_PER_CU, int RBITS = RADIX_BITS, bool TEXT_KEYS = false>
__global__ __launch_bounds__(THREADS, WG_PER_CU * THREADS / 256) void k_onesweep(
    const KeyT *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, KeyT *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
    OnesweepPass P)
{
    constexpr int RADIX = 1 << RBITS;              // (shadows sa::RADIX inside this kernel)
    constexpr int RADIX_BITS = RBITS;
    static_assert(RBITS == 8 || !SEQ, "the digits of the sequential shapes travel as bytes");
    static_assert(!TEXT_KEYS || sizeof(KeyT) == 4, "text keys are the top 32 key bits");
    constexpr int TILE = THREADS * ITEMS;
    constexpr int NWAVES = THREADS / WAVE;
    constexpr int WAVE_ELEMS = WAVE * ITEMS;
    static_assert(THREADS >= RADIX, "thread d owns digit d");
    static_assert(TILE < 65536, "16-bit tile-local counters");
    static_assert(ITEMS % 4 == 0, "digits are packed four to a register, tile positions two");
    __shared__ __attribute__((aligned(16))) KeyT lds_k[TILE];
    __shared__ __attribute__((aligned(16))) uint32_t lds_v_sep[SEQ ? 1 : TILE];
    __shared__ uint16_t wave_hist[NWAVES][RADIX];
    __shared__ uint32_t digit_base[RADIX];     // first stage slot of digit d
    __shared__ uint32_t goff[RADIX];           // global position = goff + stage slot
    __shared__ uint32_t bnd[RADIX];            // next pass: first global position behind the segment the digit run starts in ...
    __shared__ uint8_t seg0[RADIX];            // ... and that segment
    __shared__ uint32_t hist2[RADIX * OS_NSEG];
    __shared__ uint32_t scan_lds[NWAVES + 1];
    __shared__ int s_tile, s_seg;
    uint32_t *lds_v = SEQ ? (uint32_t *)lds_k : lds_v_sep;

    const int tid = threadIdx.x, l = lane_id(), w = wave_id();
    const unsigned xcc = os_xcc_id();
    const int nseg = P.nseg;
    const bool count_next = P.hist_next != nullptr;
    for (int i = tid; i < RADIX * OS_NSEG; i += THREADS) hist2[i] = 0;
    int cand = 0;                              // index into my list of candidate segments: my XCD's first, then all
    int cur_seg = -1;
    uint32_t my_segbase = 0;                   // thread d: where the current segment's run of digit d starts
    uint16_t *my_hist = wave_hist[w];
    const int e0 = w * WAVE_ELEMS + l;
    const uint32_t seg_elems = (uint32_t)P.tiles_per_seg * (uint32_t)TILE;
    const unsigned long long tagA = ((unsigned long long)(P.epoch * 4u + 1u)) << 32, tagI = ((unsigned long long)(P.epoch * 4u + 2u)) << 32;
    os_gu64 *status = (os_gu64 *)P.status;
#ifdef SA_AMD_DIAG
    // diagnostic library only (P.flags bit 7): cycles of wave 0 per phase, summed over tiles and workgroups (tools/onesweep_stamps.py)
    const bool stamping = (P.flags & 128u) != 0 && tid == 0;
    unsigned long long t_prev = stamping ? __builtin_amdgcn_s_memtime() : 0ull;
    auto stamp = [&](int phase) {
        if (stamping) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            atomicAdd(&g_phase_cycles[phase], now - t_prev);
            t_prev = now;
        }
    };
#else
    auto stamp = [](int) {};
#endif

    // tid 0 asks for tiles: the segments of my XCD first (xcc, xcc + 8, ...), then any segment that has tiles left
    const int own = ((int)xcc < nseg) ? (nseg - 1 - (int)xcc) / 8 + 1 : 0;
    auto cand_seg = [&](int c) { return c < own ? (int)xcc + 8 * c : c - own; };
    auto seg_tiles = [&](int s) { int cnt = P.tiles - s * P.tiles_per_seg; return cnt > P.tiles_per_seg ? P.tiles_per_seg : cnt; };
    auto take_ticket = [&](int *sg) -> int {                      // waits for the atomics it issues
        while (cand < own + nseg) {
            const int s = cand_seg(cand), cnt = seg_tiles(s);
            if (cnt > 0) {
                const int k = (int)atomicAdd(&P.tickets[s], 1u);
                if (k < cnt) { *sg = s; return s * P.tiles_per_seg + k; }
            }
            ++cand;
        }
        *sg = -1;
        return -1;
    };
    // Tickets are taken ONE TILE AHEAD and not waited for: the atomic for the next tile is issued at the top of a tile (its
    // round trip hides behind the tile's key loads) and looked at when the tile is done.  Every workgroup does so at the
    // same point of its loop, so ticket order is still the order in which tiles start.
    int next_t = -1, next_seg = -1, pend_s = -1, pend_k = 0x7fffffff;
    if (tid == 0) next_t = take_ticket(&next_seg);
    for (;;) {
        if (tid == 0) { s_tile = next_t; s_seg = next_seg; }
        __syncthreads();                       // (also: the previous tile's LDS stage is free)
        const int t = s_tile, seg = s_seg;
        if (t < 0) break;
        if (tid == 0) {
            pend_s = cand < own + nseg ? cand_seg(cand) : -1;
            pend_k = (pend_s >= 0 && seg_tiles(pend_s) > 0) ? (int)atomicAdd(&P.tickets[pend_s], 1u) : 0x7fffffff;
        }
        const int seg_first = seg * P.tiles_per_seg;
        // (opaque copies: the per-item offsets derived from them are recomputed per tile instead of being hoisted out of the
        // tile loop into forty registers that then spill)
        int e0v = e0, tidv = tid;
        asm volatile("" : "+v"(e0v), "+v"(tidv));
        // element of the tile that item j of this lane holds: wave-striped (lane l, item j -> 64 j + l inside the wave's block: the
        // rank inside the wave is then a stable one), or -- TEXT_KEYS, the first pass of a sort, where the order among equal
        // digits is free -- ITEMS consecutive positions per lane, whose overlapping 4-byte keys come out of ONE 20-byte read
        int ebv = w * WAVE_ELEMS + l * ITEMS;
        asm volatile("" : "+v"(ebv));
#define OS_ELEM(j) (TEXT_KEYS ? ebv + (j) : e0v + (j) * WAVE)
        const int64_t base = (int64_t)t * TILE;
        const int valid = (P.n - base) >= TILE ? TILE : (int)(P.n - base);
        const bool full = valid == TILE;
        KeyT key[ITEMS];
        uint32_t val[ITEMS], pp[ITEMS / 2];    // pp: tile positions (< 65536), two to a register
        uint32_t xpack = 0;                    // TEXT_KEYS: the two key bits below key[j] in bits [2j, 2j + 2)
#define OS_POS(j) ((pp[(j) >> 1] >> (16 * ((j) & 1))) & 0xffffu)
        if (TEXT_KEYS && P.text_bits == 2) {
            // four symbols: 12 consecutive suffixes = 24 + 30 stream bits from byte (base + first position) / 4 on (the zero bytes behind
            // the packed text and the slab's slack cover the reads of the last tile; positions behind the text are masked below)
            const uint64_t win = packed_window64(P.text + ((base + ebv) >> 2));
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) key[j] = (full || ebv + j < valid) ? (KeyT)(uint32_t)(win >> (32 - 2 * j)) : (KeyT)~(KeyT)0;
            if (P.val_extra) {                 // (uniform) the symbol behind the key: stream bits [2j + 32, 2j + 34)
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) xpack |= (((uint32_t)win >> (30 - 2 * j)) & 3u) << (2 * j);
            }
        } else if (TEXT_KEYS && base + TILE + 24 <= P.text_n) {   // (uniform) every read below stays inside the text
            static_assert(!TEXT_KEYS || (ITEMS <= 13 && ITEMS % 4 == 0), "ITEMS + 3 bytes out of five aligned words (2 ITEMS + 30 bits out of 64 for the packed text); a lane's first position is word-aligned relative to the tile");
            const uintptr_t a = (uintptr_t)(P.text + base + ebv);
            const uint32_t *W = (const uint32_t *)(a & ~(uintptr_t)3);
            const uint32_t sh = (uint32_t)(a & 3u);                // (the same for every lane: tile, wave block and lane stride are multiples of 4)
            const Words4 q = *(const Words4 *)W;                   // (one 16-byte load from a dword-aligned address)
            const uint32_t w4 = W[4];
            const uint32_t v0 = __builtin_amdgcn_alignbyte(q.y, q.x, sh), v1 = __builtin_amdgcn_alignbyte(q.z, q.y, sh),
                           v2 = __builtin_amdgcn_alignbyte(q.w, q.z, sh), v3 = __builtin_amdgcn_alignbyte(w4, q.w, sh);
            const uint32_t v[5] = { v0, v1, v2, v3, 0u };
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                key[j] = (KeyT)__builtin_bswap32(__builtin_amdgcn_alignbyte(v[(j >> 2) + 1], v[j >> 2], (uint32_t)(j & 3)));
            if (P.val_extra) {                 // (uniform) the top two bits of the byte behind the key
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) xpack |= ((v[(j + 4) >> 2] >> (8 * ((j + 4) & 3) + 6)) & 3u) << (2 * j);
            }
        } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int e = OS_ELEM(j);
            if (TEXT_KEYS) {
                key[j] = (full || e < valid) ? (KeyT)text_key32(P.text, P.text_n, base + e) : (KeyT)~(KeyT)0;
                if (P.val_extra && base + e + 4 < P.text_n) xpack |= ((uint32_t)P.text[base + e + 4] >> 6) << (2 * j);
            } else key[j] = (full || e < valid) ? keys_in[base + e] : (KeyT)~(KeyT)0;
        }
        }
        if (seg != cur_seg) {                  // (uniform) where do this segment's digit runs start?
            uint32_t tot = 0, below = 0;
            if (tid < RADIX) {
                for (int s = 0; s < nseg; ++s) {
                    const uint32_t c = P.hist_cur[tid * nseg + s];
                    if (s < seg) below += c;
                    tot += c;
                }
            }
            uint32_t all;
            const uint32_t dstart = block_excl_sum<THREADS>(tot, scan_lds, &all);
            my_segbase = dstart + below;
            cur_seg = seg;
        }
        for (int i = tid; i < NWAVES * RADIX / 2; i += THREADS) ((uint32_t *)&wave_hist[0][0])[i] = 0;
        __syncthreads();
        stamp(0);      // ticket, key loads issued, segment switch, counters zeroed, barrier (waits for the keys)
        // ---- rank inside the wave: lanes with my digit below me (8 ballots + mbcnt), wave totals in LDS ----
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const bool ok = full || OS_ELEM(j) < valid;
            const uint32_t d = digit_of(key[j], P.shift, P.dmask);
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
            if ((j & 1) == 0) pp[j >> 1] = prior + below; else pp[j >> 1] |= (prior + below) << 16;
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(1);      // ranking
        // the values are loaded only now (registers), and stay in flight across the LDS-only barriers below
        // (SEQ: later still, when the keys' registers are free)
        auto load_vals = [&]() {
        if (vals_in) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int e = OS_ELEM(j);
                val[j] = (full || e < valid) ? vals_in[base + e] : 0u;
            }
        } else {                               // no values array: the value is the index itself
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) val[j] = (uint32_t)(base + OS_ELEM(j));
            if (TEXT_KEYS && P.val_extra) {
                const int e = P.val_extra;
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) val[j] |= (((xpack >> (2 * j)) & 3u) >> (2 - e)) << (32 - e);
            }
        }
        };
        if (!SEQ) load_vals();
        lds_barrier();
        stamp(2);      // value loads issued + barrier
        // ---- thread d: per-wave offsets, the tile's count of digit d ----
        uint32_t tot = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int ww = 0; ww < NWAVES; ++ww) {
                const uint32_t cnt = wave_hist[ww][tid];
                wave_hist[ww][tid] = (uint16_t)tot;
                tot += cnt;
            }
        }
        // publish the count, start the look-back: OS_LB predecessors read in one go, consumed after the key staging
        unsigned long long lb[OS_LB];
        const bool walk = tid < RADIX && t > seg_first;
        // The first look at the predecessors comes only when the tile is staged: a granule published a microsecond ago is not
        // visible yet, and 256 threads per workgroup polling for it cost more than the round trip they tried to hide (measured:
        // 256 MiB random text, four 32-bit passes 5.56 -> 5.37 ms; C3-iid, seven 64-bit passes 13.6 -> 12.9 ms).  flags bit 0: look early (A/B)
        const bool late_look = (P.flags & 1u) == 0;
        if (walk) {
            __hip_atomic_store(status + (int64_t)t * RADIX + tid, tagA | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!late_look) {
#pragma unroll
                for (int q = 0; q < OS_LB; ++q) {
                    const int k = t - 1 - q;
                    lb[q] = k >= seg_first ? __hip_atomic_load(status + (int64_t)k * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                }
            }
        }
        uint32_t tile_total;
        const uint32_t dbase = block_excl_sum_b<THREADS, true>(tot, scan_lds, &tile_total);
        if (tid < RADIX) digit_base[tid] = dbase;
        lds_barrier();
        stamp(3);      // digit totals, count published, look-back loads issued, tile prefix (3 barriers)
        // ---- keys (and, side by side, values): stage in sorted order ----
        uint32_t dpack[SEQ ? ITEMS / 4 : 1];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t d = digit_of(key[j], P.shift, P.dmask);
            const uint32_t ps = OS_POS(j) + digit_base[d] + my_hist[d];
            if ((j & 1) == 0) pp[j >> 1] = (pp[j >> 1] & 0xffff0000u) | ps; else pp[j >> 1] = (pp[j >> 1] & 0xffffu) | (ps << 16);
            if (full || OS_ELEM(j) < valid) lds_k[ps] = key[j];
        }
        if (!SEQ) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (full || OS_ELEM(j) < valid) lds_v[OS_POS(j)] = val[j];
        } else load_vals();                    // (the keys' registers are free: the values travel while the keys go out)
        if (walk && late_look) {
#pragma unroll
            for (int q = 0; q < OS_LB; ++q) {
                const int k = t - 1 - q;
                lb[q] = k >= seg_first ? __hip_atomic_load(status + (int64_t)k * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
        }
        stamp(4);      // staged in LDS
        // ---- thread d: finish the look-back ----
        if (tid < RADIX) {
            uint32_t g = my_segbase;                                // (first tile of its segment: nothing in front)
            if (walk) {
                uint32_t acc = 0;
                int k = t - 1;
                bool done = false;
                unsigned spins = 0;
                while (!done) {
#pragma unroll
                    for (int q = 0; q < OS_LB; ++q) {
                        if (done || k - q < seg_first) break;
                        unsigned long long x = lb[q];
                        for (;;) {
                            const unsigned long long tg = x & 0xffffffff00000000ull;
                            if (tg == tagI) { acc += (uint32_t)x; done = true; break; }
                            if (tg == tagA) { acc += (uint32_t)x; break; }
                            if (++spins > (1u << 22)) { atomicAdd(P.err, 1u); done = true; break; }
                            __builtin_amdgcn_s_sleep(2);
                            x = __hip_atomic_load(status + (int64_t)(k - q) * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    k -= OS_LB;
                    if (!done) {
                        if (k < seg_first) { atomicAdd(P.err, 1u); break; }      // (cannot happen: the segment's first tile publishes an inclusive prefix)
#pragma unroll
                        for (int q = 0; q < OS_LB; ++q)
                            lb[q] = k - q >= seg_first ? __hip_atomic_load(status + (int64_t)(k - q) * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    }
                }
                g = acc;                                            // (the inclusive prefix the walk ended on carries the segment's start)
#ifdef SA_AMD_DIAG
                if ((P.flags & 128u) != 0 && tid == 7) {            // look-back statistics of digit 7's thread: walks, tiles walked, polls
                    atomicAdd(&g_phase_cycles[8], 1ull);
                    atomicAdd(&g_phase_cycles[9], (unsigned long long)(t - 1 - k));
                    atomicAdd(&g_phase_cycles[10], (unsigned long long)spins);
                }
#endif
            }
            __hip_atomic_store(status + (int64_t)t * RADIX + tid, tagI | (unsigned long long)(g + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            goff[tid] = g - dbase;
            if (count_next) {
                const uint32_t s0 = g / seg_elems;
                seg0[tid] = (uint8_t)s0;
                bnd[tid] = (s0 + 1u) * seg_elems;
            }
        }
        __syncthreads();
        stamp(5);      // look-back finished, barrier
        // ---- LDS -> global, digit runs coalesced; the next pass's digit is counted on the way ----
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int idx = tidv + j * THREADS;
            const bool ok = full || idx < valid;
            KeyT kx = 0;
            uint32_t d = 0, gp = 0;
            if (SEQ && (j & 3) == 0) dpack[j >> 2] = 0;
            if (ok) {
                kx = lds_k[idx];
                d = digit_of(kx, P.shift, P.dmask);
                gp = goff[d] + (uint32_t)idx;
                keys_out[gp] = kx;
                if (!SEQ) vals_out[gp] = lds_v[idx];
                else dpack[j >> 2] |= d << (8 * (j & 3));
            }
            if (count_next) {
                // (segment-major in LDS: neighbouring lanes carry different digits, i.e. different banks -- digit-major put the whole
                // wave on RADIX * OS_NSEG / 32 = 4 banks)
                const uint32_t slot = ok ? ((uint32_t)seg0[d] + (gp >= bnd[d] ? 1u : 0u)) * RADIX + digit_of(kx, P.shift_next, P.dmask_next) : 0u;
                const uint64_t act = __ballot(ok);
                const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
                // one (digit, segment) for the whole wave (constant high digits, runs): one add, not 64 on one address
                // (lanes past the end of a ragged tile are the wave's last ones: lane 0 is valid whenever any lane is)
                if (__all(!ok || slot == f)) {
                    if (act && l == 0) atomicAdd(&hist2[f], (uint32_t)__popcll(act));
                } else if (ok) atomicAdd(&hist2[slot], 1u);
            }
        }
        stamp(6);      // keys (and values) LDS -> global
        if (SEQ) {
            // ---- the values through the same stage ----
            lds_barrier();                     // every key has been read
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (full || OS_ELEM(j) < valid) lds_v[OS_POS(j)] = val[j];
            lds_barrier();
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int idx = tidv + j * THREADS;
                if (full || idx < valid) {
                    const uint32_t d = (dpack[j >> 2] >> (8 * (j & 3))) & 255u;
                    vals_out[goff[d] + (uint32_t)idx] = lds_v[idx];
                }
            }
        }
        stamp(7);      // values through the stage (SEQ)
#undef OS_POS
#undef OS_ELEM
        if (tid == 0) {                        // the ticket asked for at the top of this tile
            if (pend_s >= 0 && pend_k < seg_tiles(pend_s)) { next_t = pend_s * P.tiles_per_seg + pend_k; next_seg = pend_s; }
            else { if (pend_s >= 0) ++cand; next_t = take_ticket(&next_seg); }
        }
    }
    // ---- the next pass's counts: [digit][segment] ----
    if (count_next) {
        __syncthreads();
        for (int i = tid; i < RADIX * OS_NSEG; i += THREADS) {
            const uint32_t c = hist2[i];
            const int s = i / RADIX, d = i % RADIX;
            if (c && s < nseg) atomicAdd(&P.hist_next[d * nseg + s], c);
        }
    }
}

}  // namespace sa
