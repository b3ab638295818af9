// host/pipeline.hpp -- the device pipeline of one build: workspace layout, radix-sort driver, refinement rounds,
// build_device().  Replaces the arithmetic behind `cdivsufsort::sort_in_place` (reference src/saca.rs:14).
// There is deliberately no CPU fallback: every step launches the HIP kernels of kernels/*.hpp or returns an error code.
#pragma once
#include "support.hpp"
#include "tuning.hpp"
#include "pool.hpp"

#include <functional>

namespace sa {

constexpr int SORT_MAX_WG = 1024;   // spine rows are scanned by one 1024-thread block
// bucket sort of the 32-bit first stage (kernels/bucket_sort.hpp): the top 16 key bits by two global passes of 8 bits (the low 16
// inside the buckets), or the top 18 by two passes of 9 bits (the low 14 inside) for texts whose 16-bit buckets outgrow a workgroup
constexpr int BK_TOP_BITS_MAX = 18;
constexpr uint32_t BK_BUCKETS_MAX = 1u << BK_TOP_BITS_MAX;
static_assert(32 - 16 <= BK_MAX_LBITS, "two 8-bit passes inside a bucket");
constexpr size_t GRAM_MAX_ENTRIES = (size_t)1 << 24;   // gram keys: the rank table has sigma^g <= min(n, 2^24) entries
static_assert(GROUP_CAP_MAX == GS_CAP, "Tuning clamps SA_AMD_GROUP_CAP to the kernel's cap");

#ifdef SA_AMD_DIAG
// (three-kernel pass of rounds 1-2: diagnostic library only, see kernels/radix_sort.hpp)
// Tile-scatter kernel shapes (threads, items per thread, workgroups per CU).  SA_AMD_SORT_VARIANT selects one at run
// time for A/B measurements; every entry of the PRODUCT table sorts correctly.  The diagnostic library appends the
// first-generation scatter, its timing ablations (wrong orders, on purpose) and the phase-stamp build.
typedef void (*DownsweepFn)(const uint64_t *, const uint32_t *, uint64_t *, uint32_t *, uint32_t *,
                            const uint32_t *, int64_t, int, uint32_t, int64_t, int);
struct SortVariant { int threads, items, wg_per_cu; DownsweepFn fn; const char *name; };
static const SortVariant sort_variants[] = {
    { 1024, 8, 2, k_radix_downsweep_wcl<1024, 8, 16, 1, false, uint64_t, 4>, "carry-completed lines 1024x8 + LDS prefetch of half of the next tile's keys (default)" },
    { 1024, 8, 2, k_radix_downsweep_wcl<1024, 8>, "carry-completed lines 1024x8" },
    { 512, 16, 1, k_radix_downsweep_wcl<512, 16>, "carry-completed lines 512x16" },
    { 1024, 8, 2, k_radix_downsweep_wcl<1024, 8, 8>, "carry 1024x8, granule 8" },
    // (two workgroups per CU at 64 VGPRs -- 1024x4 or 512x8 with granule 8 -- measured slower: C3-iid 28.0 -> 29.8 .. 33.9 ms)
#ifdef SA_AMD_DIAG
    { 1024, 8, 2, k_radix_downsweep<1024, 8, 4>, "plain tile scatter 1024x8 (first generation)" },
    { 1024, 8, 2, k_radix_downsweep<1024, 8, 4, 1>, "plain 1024x8 ABLATION sequential stores (wrong results)" },
    { 1024, 8, 2, k_radix_downsweep<1024, 8, 4, 33>, "plain 1024x8 ABLATION no ranking + sequential stores (wrong results)" },
    { 1024, 8, 2, k_radix_downsweep<1024, 8, 4, 16>, "plain 1024x8 ABLATION no stores (wrong results)" },
    { 1024, 8, 2, k_radix_downsweep<1024, 8, 4, 49>, "plain 1024x8 ABLATION no ranking, no stores (wrong results)" },
    { 1024, 8, 2, k_radix_downsweep_wcl<1024, 8, 16, 1, true>, "carry 1024x8 DIAGNOSTIC phase stamps (tools/phase_stamps.py)" },
#endif
};
constexpr int N_SORT_VARIANTS = (int)(sizeof(sort_variants) / sizeof(sort_variants[0]));

struct SortGrid { int G; int64_t tiles_per_wg; int tile; };
static SortGrid sort_grid(int64_t count, const SortVariant &sv)
{
    SortGrid g;
    g.tile = sv.threads * sv.items;
    const int64_t tiles = ceil_div(count, g.tile);
    int max_wg = 256 * sv.wg_per_cu;
    if (max_wg > SORT_MAX_WG) max_wg = SORT_MAX_WG;
    g.tiles_per_wg = ceil_div(tiles, max_wg);
    if (g.tiles_per_wg < 1) g.tiles_per_wg = 1;
    g.G = (int)ceil_div(tiles, g.tiles_per_wg);
    if (g.G < 1) g.G = 1;
    return g;
}

#else
constexpr int N_SORT_VARIANTS = 1;      // (the product has one sort engine: the single-pass tile scatter)
#endif

// what a radix sort needs besides its ping-pong buffers
struct SortScratch {
    uint32_t *spine;                // RADIX * SORT_MAX_WG words: per-chunk counts (three-kernel pass) / two zones of segment counts + tickets (single-pass)
    uint32_t *digit_tot;            // RADIX words
    unsigned long long *status;     // single-pass scatter: 256 granules per tile (nullptr: the three-kernel pass is used)
    uint32_t *err;                  // single-pass scatter: look-back give-ups (must stay 0).  A tile whose look-back gave up (2^22 polls without
                                    // an answer: never seen, the bound turns a hang into an error) has scattered with a partial prefix -- in bounds,
                                    // wrong order --, so EVERY user of a SortScratch reads err[0] before it trusts a result: build_device at its end
                                    // (SA_AMD_EINTERNAL), sa_amd_check_integrity_device behind its sorts, the diagnostic hooks behind theirs
};

// device scratch layout for a text of n bytes
struct Workspace {
    uint64_t *keysA, *keysB, *keysC;
    uint32_t *valsA, *valsB, *isa, *U0, *U1, *G0, *G1;
    uint32_t *spine, *digit_tot, *tcnt, *thead, *tnext, *hist, *total, *chg, *has_isa;
    uint8_t *packed;           // bit-packed text (alphabets of 2, 4 or 16 symbols): n / 2 + 64 bytes
    uint8_t *gram_flags;       // gram keys: which g-grams occur (sigma^g <= min(n, 2^24) flags) and
    uint4 *gram_table;         //   the rank directory over them (16 bytes per 64 indices)
    uint32_t *surv_bits, *surv_cnt, *todo_bits, *ft_cnt, *ft_head;   // first refinement round straight from the sorted keys (k_finish_sorted)
    unsigned long long *os_status;  // look-back granules of the single-pass tile scatter: 2 KiB per 8192-element tile
    uint32_t *os_err;
    uint32_t *bk_start;             // bucket sort of the 32-bit first stage: 2^16 + 1 or 2^18 + 1 bucket starts
    uint32_t *early_bits, *early_cnt;    // early download: bitmap over the n + 1 entries of the downloaded array, marked entries per tile
    SortScratch ss;
    size_t bytes, bytes2;           // in the first (device) block, in the second (reduced-memory route: pinned host) block
};

// Slabs are laid out in the order of the take() calls below.  cap / base2: the REDUCED-MEMORY route of the host-pointer entry
// points (host/host_path.hpp): when the device cannot give the whole workspace, the first `cap` bytes' worth of slabs live in
// the device block `base` and every slab that no longer fits lives in `base2`, a block of pinned host memory that the kernels
// reach over PCIe (slow, correct: the alternative is SA_AMD_ENOMEM).  So the order below is by need: first the small slabs that
// workgroups talk through (look-back granules, tickets, counters -- they must be device memory), then the big ones from the most
// to the least used.  w.bytes = bytes in the first block, w.bytes2 = bytes in the second (0 without a cap).
static Workspace carve(void *base, int64_t n, size_t cap = ~(size_t)0, void *base2 = nullptr)
{
    Workspace w;
    const size_t N = (size_t)(n > 0 ? n : 1);
    size_t off = 0, off2 = 0;
    auto take = [&](size_t b) {
        if (off + b <= cap) { size_t o = off; off = align_up(off + b, 256); return (char *)base + o; }
        size_t o = off2; off2 = align_up(off2 + b, 256); return (char *)base2 + o;
    };
    const size_t rr_tiles = (size_t)ceil_div((int64_t)N, RR_TILE);
    const size_t ft_tiles = (size_t)ceil_div((int64_t)N, FT_TILE) + 1;
    const size_t gram_entries = N < GRAM_MAX_ENTRIES ? N : GRAM_MAX_ENTRIES;
    // ---- small, shared between workgroups ----
    w.spine = (uint32_t *)take((size_t)RADIX * SORT_MAX_WG * 4);
    w.digit_tot = (uint32_t *)take(RADIX * 4);
    w.os_status = (unsigned long long *)take(((size_t)ceil_div((int64_t)N, OS_MIN_TILE) + 1) * RADIX * 8);
    w.os_err = (uint32_t *)take(256);
    w.hist = (uint32_t *)take(256 * 4);
    w.total = (uint32_t *)take(256);
    w.chg = (uint32_t *)take((size_t)RR_CHG_COUNTERS * 32 * 4);      // (directly behind w.total: read back together)
    w.tcnt = (uint32_t *)take(rr_tiles * 4);
    w.thead = (uint32_t *)take(rr_tiles * 4);
    w.tnext = (uint32_t *)take(rr_tiles * 4);
    w.surv_cnt = (uint32_t *)take(rr_tiles * 4);          // (not tcnt: refine_list uses that one for its own compaction)
    w.ft_cnt = (uint32_t *)take(ft_tiles * 4);
    w.ft_head = (uint32_t *)take(ft_tiles * 4);
    w.bk_start = (uint32_t *)take(((size_t)BK_BUCKETS_MAX + 1) * 4);
    w.early_cnt = (uint32_t *)take(((N + 1 + EARLY_TILE - 1) / EARLY_TILE + 8) * 4);
    w.gram_table = (uint4 *)take((gram_entries / 64 + 1) * 16);
    // ---- n / 8 .. n / 2 bytes ----
    w.has_isa = (uint32_t *)take((N + 31) / 32 * 4);
    w.surv_bits = (uint32_t *)take((N + 31) / 32 * 4);
    w.todo_bits = (uint32_t *)take((N + 31) / 32 * 4);
    w.early_bits = (uint32_t *)take(((N + 1 + 31) / 32 + EARLY_THREADS) * 4);           // (whole tiles of 256 words)
    w.gram_flags = (uint8_t *)take(gram_entries + 64);
    w.packed = (uint8_t *)take(N / 2 + 64);
    // ---- the big ones, most used first ----
    w.keysA = (uint64_t *)take((N + 64) * 8);         // (+64: as two halves of n + 1 32-bit entries each, see scatter_binned)
    w.keysB = (uint64_t *)take((N + 64) * 8);
    w.valsA = (uint32_t *)take(N * 4);
    w.valsB = (uint32_t *)take(N * 4);
    w.isa = (uint32_t *)take(N * 4);
    w.U0 = (uint32_t *)take(N * 4);
    w.G0 = (uint32_t *)take(N * 4);
    w.U1 = (uint32_t *)take(N * 4);
    w.G1 = (uint32_t *)take(N * 4);
    w.keysC = (uint64_t *)take((N + 64) * 8);
    w.ss.spine = w.spine; w.ss.digit_tot = w.digit_tot; w.ss.status = w.os_status; w.ss.err = w.os_err;
    w.bytes = off;
    w.bytes2 = off2;
    return w;
}


// ------------------------------------------------------------------------------------------
// Single-pass tile scatter (kernels/onesweep.hpp): host side of one LSD sort.
// Scratch inside the spine slab: ZONES of OS_ZONE words, zone = [OS_TICKETS ticket words | RADIX * OS_NSEG segment counts].
// Pass p reads its digit's counts from zone z and writes the next digit's counts -- and takes its tickets -- in zone z + 1.
// Every pass has a zone of its own and ONE memset in front of the sort zeroes them all (a memset per pass was a 5 us launch
// per pass: eight of them in the initial sort of a 1 MiB text, whose passes take 40 us).
// ------------------------------------------------------------------------------------------
// Tile shapes (threads, keys per thread, values through the keys' LDS buffer, workgroups per CU).  SA_AMD_ONESWEEP64_SHAPE /
// SA_AMD_ONESWEEP32_SHAPE select one for A/B measurements; every shape sorts correctly.
struct OsShape { int threads, items; bool seq; int wg_per_cu; };
static const OsShape os_shapes64[] = { { 1024, 8, false, 1 }, { 512, 16, true, 2 }, { 512, 8, false, 2 } };
static const OsShape os_shapes32[] = { { 1024, 12, false, 1 }, { 512, 16, true, 2 }, { 512, 12, false, 2 }, { 1024, 8, false, 1 } };
// (measured slower at 256 MiB, profiles/r03_onesweep_shapes.txt: 1024 x 16 and larger tiles -- spills at the 128-register limit of a
// 1024-thread workgroup --, 512 x 24 likewise; two workgroups per CU bought nothing at equal tile size)
constexpr int N_OS_SHAPES64 = (int)(sizeof(os_shapes64) / sizeof(os_shapes64[0]));
constexpr int N_OS_SHAPES32 = (int)(sizeof(os_shapes32) / sizeof(os_shapes32[0]));
constexpr int OS_TICKETS = 64;                 // words in front of a zone's counts: one ticket counter per segment
constexpr int OS_MAX_RADIX = 512;              // widest digit of the single-pass scatter (9 bits: the two passes in front of the bucket sort of large texts)
constexpr int OS_ZONE = OS_TICKETS + OS_MAX_RADIX * OS_NSEG;  // words
static_assert(OS_NSEG <= OS_TICKETS, "one ticket word per segment");
constexpr int OS_MAX_ZONES = 2 * 8 + 2;         // eight passes, each possibly behind a skipped one that needed a recount, + the producer's zone
static_assert(OS_MAX_ZONES * OS_ZONE <= RADIX * SORT_MAX_WG, "the zones live in the spine slab");

#ifdef SA_AMD_DIAG
static bool onesweep_on(const SortScratch &ss, const Tuning &tn) { return ss.status != nullptr && !tn.no_onesweep; }
#else
static bool onesweep_on(const SortScratch &ss, const Tuning &) { return ss.status != nullptr; }      // (the product's only engine)
#endif

struct OnesweepGeom { int tiles, nseg, tiles_per_seg; int64_t seg_elems; };
static OnesweepGeom onesweep_geom(int64_t count, int tile)
{
    OnesweepGeom g;
    g.tiles = (int)ceil_div(count, tile);
    if (g.tiles < 1) g.tiles = 1;
    int nseg = g.tiles < OS_NSEG ? g.tiles : OS_NSEG;
    g.tiles_per_seg = (int)ceil_div(g.tiles, nseg);
    g.nseg = (int)ceil_div(g.tiles, g.tiles_per_seg);
    g.seg_elems = (int64_t)g.tiles_per_seg * tile;
    return g;
}

// Where a producer of the keys adds the counts of the first pass's digit (k_build_keys: counts[d * G + chunk]), for the
// sort that will run on `count` pairs with this scratch: pointer, chunk size in elements, chunks.  The producer's stream
// must zero *zero_bytes bytes at *zero_ptr first.
struct FirstCounts { uint32_t *counts; int64_t chunk_elems; int G; void *zero_ptr; size_t zero_bytes; };
static FirstCounts sort_first_counts(const SortScratch &ss, const Tuning &tn, int64_t count, bool keys32);

static int cu_count()
{
    static int cus = 0;
    if (cus > 0) return cus;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    else { (void)hipGetLastError(); cus = 256; }
    return cus;
}

static int read_words(void *dst, const void *dsrc, size_t bytes, hipStream_t st);

__global__ __launch_bounds__(RADIX) void k_os_digit_totals(const uint32_t *__restrict__ hist, int nseg, uint32_t *__restrict__ digit_tot)
{
    uint32_t s = 0;
    for (int q = 0; q < nseg; ++q) s += hist[threadIdx.x * nseg + q];
    digit_tot[threadIdx.x] = s;
}

// Short sorts (fewer than SA_AMD_COUNT_NEXT_MIN_N pairs): ONE segment, and the counts of EVERY pass's digit from one read of the
// keys in front of the first pass -- a digit's totals do not depend on the order the pairs are in, and with one segment the
// totals are all a pass needs.  (Per pass either a counting kernel of its own, 7 us, or the flush of the in-pass count, more:
// 1 MiB of random bytes, five passes: 0.26 -> 0.23 ms.)
constexpr int HA_THREADS = 256;
template <typename KeyT>
__global__ __launch_bounds__(HA_THREADS) void k_radix_hist_all(const KeyT *__restrict__ keys, int64_t count, int begin_bit, int end_bit,
                                                               uint32_t *__restrict__ zone0, int zone_words, int ticket_words)
{
    __shared__ uint32_t h[8][RADIX];
    for (int i = threadIdx.x; i < 8 * RADIX; i += HA_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const int np = (end_bit - begin_bit + RADIX_BITS - 1) / RADIX_BITS;      // (<= 8, host-checked)
    for (int64_t i = (int64_t)blockIdx.x * HA_THREADS + threadIdx.x; i < count; i += (int64_t)gridDim.x * HA_THREADS) {
        const uint64_t k = (uint64_t)keys[i];
        for (int p = 0; p < np; ++p) {
            const int sh = begin_bit + p * RADIX_BITS, nb = end_bit - sh < RADIX_BITS ? end_bit - sh : RADIX_BITS;
            atomicAdd(&h[p][(k >> sh) & ((1u << nb) - 1u)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * RADIX; i += HA_THREADS) {
        const uint32_t c = (&h[0][0])[i];
        if (c) atomicAdd(&zone0[(size_t)(i / RADIX) * zone_words + ticket_words + (i % RADIX)], c);      // (one segment: counts[d * 1 + 0])
    }
}

template <typename KeyT, int THREADS, int ITEMS, bool SEQ, int RBITS = RADIX_BITS>
static int sort_pairs_onesweep(KeyT *keys_in, uint32_t *vals_in, KeyT *keys_alt, uint32_t *vals_alt, int64_t count, int begin_bit, int end_bit,
                               const SortScratch &ss, uint32_t *final_vals, hipStream_t st, KeyT **keys_res, uint32_t **vals_res, int *passes,
                               int *skipped, const Tuning &tn, bool iota, bool may_skip, bool first_counted,
                               const uint8_t *text = nullptr, int64_t text_n = 0,      // != nullptr (32-bit keys only): the FIRST pass reads its keys from the text (k_onesweep<..., TEXT_KEYS>)
                               int text_bits = 8,                                      //   8: the text itself (all 256 byte values), 2: the bit-packed text of a four-symbol alphabet
                               int val_extra = 0)                                      //   the first pass puts that many key bits below the 32 into the top bits of the values (OnesweepPass::val_extra)
{
    constexpr int TILE = THREADS * ITEMS;
    constexpr int R = 1 << RBITS;
    static_assert(R <= OS_MAX_RADIX && (RBITS == RADIX_BITS || TILE >= 2 * OS_MIN_TILE), "zones and granule slab are sized for 8-bit digits of 4 Ki-element tiles");
    static_assert(TILE >= OS_MIN_TILE, "the granule slab is sized for tiles of at least OS_MIN_TILE elements");
    constexpr bool K64 = sizeof(KeyT) == 8;
    OnesweepGeom g = onesweep_geom(count, TILE);
    auto zone = [&](int i) { return ss.spine + (size_t)i * OS_ZONE; };
    int z = 0;                                    // zone that holds (or will hold) the counts of the coming pass's digit
    bool have_counts = first_counted;
    const int npass = (int)ceil_div(end_bit - begin_bit, RBITS);
    // every pass's counts up front (k_radix_hist_all): short sorts of keys that exist as an array, no pass to be skipped
    const bool upfront = RBITS == RADIX_BITS && !tn.no_upfront_counts && count < tn.count_next_min_n && !may_skip && !text && npass <= 8 && npass >= 2;
    if (upfront) { g.nseg = 1; g.tiles_per_seg = g.tiles; g.seg_elems = (int64_t)g.tiles * TILE; }
    HIP_TRY(hipMemsetAsync(ss.status, 0, (size_t)g.tiles * R * 8, st));
    {
        // zone 0 holds the producer's counts (first_counted) and stays; everything behind it starts from zero
        int zones = (may_skip ? 2 * npass : npass) + 1;
        if (zones > OS_MAX_ZONES) return SA_AMD_EINTERNAL;
        const int z0 = (first_counted && !upfront) ? 1 : 0;      // (up front: the producer counted per segment of another geometry -- counted again)
        HIP_TRY(hipMemsetAsync(zone(z0), 0, (size_t)(zones - z0) * OS_ZONE * 4, st));
    }
    if (upfront) {
        int blocks = (int)ceil_div(count, (int64_t)HA_THREADS * 16);
        if (blocks > 512) blocks = 512;
        if (blocks < 1) blocks = 1;
        PROF(K64 ? KC_UPSWEEP : KC_UPSWEEP32, count, st, hipLaunchKernelGGL((k_radix_hist_all<KeyT>), dim3((unsigned)blocks), dim3(HA_THREADS), 0, st, (const KeyT *)keys_in, count,
                                                                        begin_bit, end_bit, zone(0), (int)OS_ZONE, (int)OS_TICKETS));
        have_counts = true;
    }
    KeyT *kin = keys_in, *kout = keys_alt;
    uint32_t *vin = vals_in, *vout = vals_alt;
    constexpr int WG_PER_CU = THREADS <= 512 ? 2 : 1;
    int grid = cu_count() * WG_PER_CU;
    if (grid > g.tiles) grid = g.tiles;
    uint32_t epoch = 0;
    for (int shift = begin_bit; shift < end_bit; shift += RBITS) {
        const int nb = (end_bit - shift) < RBITS ? (end_bit - shift) : RBITS;
        const uint32_t dmask = (1u << nb) - 1u;
        const bool last = shift + RBITS >= end_bit;
        uint32_t *vdst = (last && final_vals) ? final_vals : vout;
        if (!have_counts) {
            if (RBITS != RADIX_BITS) return SA_AMD_EINTERNAL;     // (the counting kernels have 256 bins: a wide-digit sort gets its first counts from the producer of the keys)
            // one read of the keys for the counts of this digit (first pass of a sort whose producer did not count, or the
            // pass after a skipped one); zone z is still zero
            int split = 2048 / g.nseg;
            while (split > 1 && g.seg_elems / split < 8192) split /= 2;
            const int64_t sub = K64 ? ((ceil_div(g.seg_elems, split) + 1) & ~(int64_t)1) : ((ceil_div(g.seg_elems, split) + 3) & ~(int64_t)3);
            if (K64)
                PROF(KC_UPSWEEP, count, st, hipLaunchKernelGGL((k_radix_upsweep), dim3(g.nseg * split), dim3(SORT_THREADS), 0, st, (const uint64_t *)kin,
                                                               zone(z) + OS_TICKETS, count, shift, dmask, g.seg_elems, g.nseg, split, sub));
            else
                PROF(KC_UPSWEEP32, count, st, hipLaunchKernelGGL((k_radix_upsweep32), dim3(g.nseg * split), dim3(SORT_THREADS), 0, st, (const uint32_t *)kin,
                                                                 zone(z) + OS_TICKETS, count, shift, dmask, g.seg_elems, g.nseg, split, sub));
        }
        if (RBITS == RADIX_BITS && may_skip && !tn.no_run_skip && count >= tn.run_skip_min && !(iota && *passes == 0) && !(last && final_vals)) {
            // a digit that is the same for EVERY element makes the pass the identity (the sort is stable): skip it
            hipLaunchKernelGGL(k_os_digit_totals, dim3(1), dim3(RADIX), 0, st, (const uint32_t *)(zone(z) + OS_TICKETS), g.nseg, ss.digit_tot);
            LAUNCH_CHECK(st);
            uint32_t tot[RADIX];
            { const int rcw = read_words(tot, ss.digit_tot, sizeof(tot), st); if (rcw) return rcw; }
            bool constant = false;
            for (int d = 0; d < RADIX; ++d) constant |= (int64_t)tot[d] == count;
            if (constant) { ++*skipped; have_counts = false; ++z; continue; }     // (zone z holds the skipped digit's counts: the recount takes the next, clean one)
        }
        OnesweepPass P;
        P.hist_cur = zone(z) + OS_TICKETS;
        // (short inputs: the flush of the next digit's counts -- 256 x segments atomics per workgroup on the same few lines -- costs
        // more than a counting kernel of its own)
        // (measured, tools/midsize_knobs.py: 1 MiB of random bytes 0.420 -> 0.369 ms, 2 MiB of English 0.902 -> 0.777; the other way
        // round below 400 K pairs -- a launch more per pass -- and from 8 M on)
        const bool count_next = !upfront && !last && (RBITS != RADIX_BITS || count >= tn.count_next_min_n || count < tn.count_next_below_n);
        P.hist_next = count_next ? zone(z + 1) + OS_TICKETS : nullptr;
        P.tickets = zone(z + 1);
        P.status = ss.status;
        P.err = ss.err;
        P.n = count;
        P.shift = shift; P.dmask = dmask;
        P.shift_next = shift + RBITS;
        { const int nbn = (end_bit - P.shift_next) < RBITS ? (end_bit - P.shift_next) : RBITS; P.dmask_next = last ? 0u : (1u << nbn) - 1u; }
        P.nseg = g.nseg; P.tiles_per_seg = g.tiles_per_seg; P.tiles = g.tiles;
        P.epoch = ++epoch;
        P.flags = (uint32_t)tn.onesweep_flags;
        P.text = text; P.text_n = text_n; P.text_bits = text_bits;
        P.val_extra = (text && *passes == 0 && iota) ? val_extra : 0;
        if (!K64 && !SEQ && text && *passes == 0)
            PROF(KC_ONESWEEP32, count, st,
                 hipLaunchKernelGGL((k_onesweep<THREADS, ITEMS, KeyT, SEQ, WG_PER_CU, RBITS, !K64 && !SEQ>), dim3(grid), dim3(THREADS), 0, st, (const KeyT *)kin,
                                    (const uint32_t *)((iota && *passes == 0) ? nullptr : vin), kout, vdst, P));
        else
        PROF(K64 ? KC_ONESWEEP : KC_ONESWEEP32, count, st,
             hipLaunchKernelGGL((k_onesweep<THREADS, ITEMS, KeyT, SEQ, WG_PER_CU, RBITS>), dim3(grid), dim3(THREADS), 0, st, (const KeyT *)kin,
                                (const uint32_t *)((iota && *passes == 0) ? nullptr : vin), kout, vdst, P));
        KeyT *tk = kin; kin = kout; kout = tk;
        uint32_t *free_v = vin;                   // the values just consumed become the next scratch target
        vin = vdst;
        vout = free_v;
        ++*passes;
        ++z;
        have_counts = count_next || upfront;
    }
    *keys_res = kin; *vals_res = vin;
    return SA_AMD_OK;
}

struct SortResult { uint64_t *keys; uint32_t *vals; int passes; int skipped; };

// stable LSD sort of `count` pairs on key bits [begin_bit, end_bit); ping-pongs between in/alt.
// spine: RADIX * SORT_MAX_WG words, digit_tot: RADIX words.  final_vals (optional): the LAST pass
// writes its values there instead of into the ping-pong buffer (the initial sort delivers
// straight into SA this way).
static int sort_pairs(uint64_t *keys_in, uint32_t *vals_in, uint64_t *keys_alt, uint32_t *vals_alt, int64_t count,
                      int begin_bit, int end_bit, const SortScratch &ss, uint32_t *final_vals,
                      hipStream_t st, SortResult *res, const Tuning &tn, bool iota = false,   // iota: value i = index i, vals_in is scratch only
                      bool may_skip = false,                                                   // look for passes that are the identity (costs a read-back per pass)
                      bool first_counted = false)                                              // the producer of keys_in has histogrammed the first digit (sort_first_counts says where and how)
{
    res->keys = keys_in; res->vals = vals_in; res->passes = 0; res->skipped = 0;
    if (count <= 1 || end_bit <= begin_bit) return SA_AMD_OK;
    if (onesweep_on(ss, tn)) {
#define OS_CALL64(T, I, S) sort_pairs_onesweep<uint64_t, T, I, S>(keys_in, vals_in, keys_alt, vals_alt, count, begin_bit, end_bit, ss, final_vals, st, \
                                                                 &res->keys, &res->vals, &res->passes, &res->skipped, tn, iota, may_skip, first_counted)
        switch (tn.onesweep64_shape) {
        case 1: return OS_CALL64(512, 16, true);
        case 2: return OS_CALL64(512, 8, false);
        default: return OS_CALL64(1024, 8, false);
        }
#undef OS_CALL64
    }
#ifdef SA_AMD_DIAG
    uint32_t *spine = ss.spine, *digit_tot = ss.digit_tot;
    const SortVariant &sv = sort_variants[tn.sort_variant];
    const SortGrid g = sort_grid(count, sv);
    uint64_t *kin = keys_in, *kout = keys_alt;
    uint32_t *vin = vals_in, *vout = vals_alt;
    for (int shift = begin_bit; shift < end_bit; shift += RADIX_BITS) {
        const int nb = (end_bit - shift) < RADIX_BITS ? (end_bit - shift) : RADIX_BITS;
        const uint32_t dmask = (1u << nb) - 1u;
        const bool last = shift + RADIX_BITS >= end_bit;
        uint32_t *vdst = (last && final_vals) ? final_vals : vout;
        {
            const int64_t chunk = g.tiles_per_wg * g.tile;
            int split = 2048 / g.G;
            if (split < 1) split = 1;
            while (split > 1 && chunk / split < 4096) split /= 2;
            const int64_t sub = (ceil_div(chunk, split) + 1) & ~(int64_t)1;
            // (atomic accumulation needs a zeroed spine: once here, afterwards every downsweep zeroes what it consumed)
            if (first_counted && shift == begin_bit) {
                // (nothing to do: k_build_keys has added this pass's digit counts to the spine)
            } else {
            if (split > 1 && res->passes == 0) HIP_TRY(hipMemsetAsync(spine, 0, (size_t)RADIX * g.G * 4, st));
            PROF(KC_UPSWEEP, count, st, hipLaunchKernelGGL((k_radix_upsweep), dim3(g.G * split), dim3(SORT_THREADS), 0, st, kin, spine,
                                                           count, shift, dmask, chunk, g.G, split, sub));
            }
        }
        PROF(KC_SPINE, (int64_t)RADIX * g.G, st, hipLaunchKernelGGL((k_spine_rows), dim3(RADIX), dim3(SPINE_THREADS), 0, st,
                                                                    spine, digit_tot, g.G));
        // A digit that is the same for EVERY element makes the pass the identity (the sort is stable): skip the tile scatter.
        // Worth a 1 KiB read-back (a host round trip of ~30 us) only for the large global sorts of the refinement rounds: texts
        // that are one run or one period keep hundreds of millions of suffixes in a few groups round after round, and their
        // (group, rank) keys are constant in most digits.  The ISA passes never look, the initial sort only for a text of one byte value.
        if (may_skip && !tn.no_run_skip && count >= tn.run_skip_min && !(iota && res->passes == 0) && !(last && final_vals)) {
            uint32_t tot[RADIX];
            { const int rcw = read_words(tot, digit_tot, sizeof(tot), st); if (rcw) return rcw; }
            bool constant = false;
            for (int d = 0; d < RADIX; ++d) constant |= (int64_t)tot[d] == count;
            if (constant) {
                HIP_TRY(hipMemsetAsync(spine, 0, (size_t)RADIX * g.G * 4, st));     // (the tile scatter would have zeroed what it consumed)
                res->skipped++;
                continue;
            }
        }
        PROF(KC_DOWNSWEEP, count, st, hipLaunchKernelGGL((sv.fn), dim3(g.G), dim3(sv.threads), 0, st,
                                                         (const uint64_t *)kin, (const uint32_t *)((iota && res->passes == 0) ? nullptr : vin), kout, vdst,
                                                         spine, (const uint32_t *)digit_tot, count, shift,
                                                         dmask, g.tiles_per_wg, g.G));
        uint64_t *tk = kin; kin = kout; kout = tk;
        uint32_t *free_v = vin;     // the values just consumed become the next scratch target
        vin = vdst;
        vout = free_v;
        res->passes++;
    }
    res->keys = kin; res->vals = vin;
    return SA_AMD_OK;
#else
    return SA_AMD_EINTERNAL;        // (no scratch for the single-pass scatter: cannot happen, every caller carves it)
#endif
}

struct SortResult32 { uint32_t *keys; uint32_t *vals; int passes; };
#ifdef SA_AMD_DIAG
// 32-bit keys (two-stage initial sort): same three-kernel pass, 12 Ki-pair tiles by default (the LDS stage holds more 4-byte elements)
constexpr int SORT32_THREADS = 1024;
typedef void (*Downsweep32Fn)(const uint32_t *, const uint32_t *, uint32_t *, uint32_t *, uint32_t *, const uint32_t *, int64_t, int,
                              uint32_t, int64_t, int);
struct Sort32Variant { int items; Downsweep32Fn fn; };
static const Sort32Variant sort32_variants[] = {
    { 12, k_radix_downsweep_wcl<SORT32_THREADS, 12, 16, 1, false, uint32_t, 12> },    // default: next tile's keys prefetched into LDS
    { 12, k_radix_downsweep_wcl<SORT32_THREADS, 12, 16, 1, false, uint32_t> },
    { 8, k_radix_downsweep_wcl<SORT32_THREADS, 8, 16, 1, false, uint32_t> },
    { 16, k_radix_downsweep_wcl<SORT32_THREADS, 16, 16, 1, false, uint32_t> },        // spills
    { 8, k_radix_downsweep_wcl<SORT32_THREADS, 8, 16, 1, false, uint32_t, 8> },
    // (two workgroups per CU: 1024 x 4 or 1024 x 8 with granule 8 and 64 VGPRs measured slower, 8.6 -> 9.0 .. 10.2 ms at 256 MiB)
};
constexpr int N_SORT32_VARIANTS = (int)(sizeof(sort32_variants) / sizeof(sort32_variants[0]));

struct SortGrid32 { int G; int64_t tiles_per_wg, tile; };
static SortGrid32 sort_grid32(int64_t count, const Sort32Variant &sv)
{
    SortGrid32 g;
    g.tile = (int64_t)SORT32_THREADS * sv.items;
    const int64_t tiles = ceil_div(count, g.tile);
    g.tiles_per_wg = ceil_div(tiles, 512);
    if (g.tiles_per_wg < 1) g.tiles_per_wg = 1;
    g.G = (int)ceil_div(tiles, g.tiles_per_wg);
    return g;
}

#else
constexpr int N_SORT32_VARIANTS = 1;
#endif

static int sort_pairs32(uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_alt, uint32_t *vals_alt, int64_t count, int begin_bit,
                        int end_bit, const SortScratch &ss, uint32_t *final_vals, hipStream_t st, SortResult32 *res,
                        const Tuning &tn, bool iota = false, bool first_counted = false,
                        int rbits = RADIX_BITS,       // 9: nine-bit digits (single-pass engine only, first digit counted by the producer)
                        const uint8_t *text = nullptr, int64_t text_n = 0, int text_bits = 8,      // the first pass reads its keys from this text (single-pass engine, default tile, counted)
                        int val_extra = 0)
{
    res->keys = keys_in; res->vals = vals_in; res->passes = 0;
    if (count <= 1 || end_bit <= begin_bit) return SA_AMD_OK;
    if (rbits != RADIX_BITS && (rbits != 9 || !onesweep_on(ss, tn) || !first_counted)) return SA_AMD_EINTERNAL;
    if (text && (!onesweep_on(ss, tn) || !first_counted || tn.onesweep32_shape != 0)) return SA_AMD_EINTERNAL;
    if (onesweep_on(ss, tn)) {
        int skipped = 0;
        if (rbits == 9)
            return sort_pairs_onesweep<uint32_t, 1024, 12, false, 9>(keys_in, vals_in, keys_alt, vals_alt, count, begin_bit, end_bit, ss, final_vals, st,
                                                                     &res->keys, &res->vals, &res->passes, &skipped, tn, iota, false, first_counted, text, text_n, text_bits, val_extra);
        if (text)
            return sort_pairs_onesweep<uint32_t, 1024, 12, false, RADIX_BITS>(keys_in, vals_in, keys_alt, vals_alt, count, begin_bit, end_bit, ss, final_vals, st,
                                                                              &res->keys, &res->vals, &res->passes, &skipped, tn, iota, false, first_counted, text, text_n, text_bits, val_extra);
#define OS_CALL32(T, I, S) sort_pairs_onesweep<uint32_t, T, I, S>(keys_in, vals_in, keys_alt, vals_alt, count, begin_bit, end_bit, ss, final_vals, st, \
                                                                 &res->keys, &res->vals, &res->passes, &skipped, tn, iota, false, first_counted)
        switch (tn.onesweep32_shape) {
        case 1: return OS_CALL32(512, 16, true);
        case 2: return OS_CALL32(512, 12, false);
        case 3: return OS_CALL32(1024, 8, false);
        default: return OS_CALL32(1024, 12, false);
        }
#undef OS_CALL32
    }
#ifdef SA_AMD_DIAG
    uint32_t *spine = ss.spine, *digit_tot = ss.digit_tot;
    const Sort32Variant &sv = sort32_variants[tn.sort32_variant];
    const SortGrid32 g32 = sort_grid32(count, sv);
    const int64_t SORT32_TILE = g32.tile, tiles_per_wg = g32.tiles_per_wg;
    const int G = g32.G;
    uint32_t *kin = keys_in, *kout = keys_alt, *vin = vals_in, *vout = vals_alt;
    for (int shift = begin_bit; shift < end_bit; shift += RADIX_BITS) {
        const int nb = (end_bit - shift) < RADIX_BITS ? (end_bit - shift) : RADIX_BITS;
        const uint32_t dmask = (1u << nb) - 1u;
        const bool last = shift + RADIX_BITS >= end_bit;
        uint32_t *vdst = (last && final_vals) ? final_vals : vout;
        {
            const int64_t chunk = tiles_per_wg * SORT32_TILE;
            int split = 2048 / G;
            if (split < 1) split = 1;
            while (split > 1 && chunk / split < 8192) split /= 2;
            const int64_t sub = (ceil_div(chunk, split) + 3) & ~(int64_t)3;
            if (first_counted && shift == begin_bit) {
                // (k_build_keys has added this pass's digit counts to the spine)
            } else {
            if (split > 1 && res->passes == 0) HIP_TRY(hipMemsetAsync(spine, 0, (size_t)RADIX * G * 4, st));
            PROF(KC_UPSWEEP32, count, st, hipLaunchKernelGGL((k_radix_upsweep32), dim3(G * split), dim3(SORT_THREADS), 0, st,
                                                           (const uint32_t *)kin, spine, count, shift, dmask, chunk, G, split, sub));
            }
        }
        PROF(KC_SPINE, (int64_t)RADIX * G, st, hipLaunchKernelGGL((k_spine_rows), dim3(RADIX), dim3(SPINE_THREADS), 0, st, spine, digit_tot, G));
        PROF(KC_DOWNSWEEP32, count, st, hipLaunchKernelGGL((sv.fn),
                                                         dim3(G), dim3(SORT32_THREADS), 0, st, (const uint32_t *)kin,
                                                         (const uint32_t *)((iota && res->passes == 0) ? nullptr : vin), kout,
                                                         vdst, spine, (const uint32_t *)digit_tot, count, shift, dmask,
                                                         tiles_per_wg, G));
        uint32_t *tk = kin; kin = kout; kout = tk;
        uint32_t *free_v = vin;
        vin = vdst;
        vout = free_v;
        res->passes++;
    }
    res->keys = kin; res->vals = vin;
    return SA_AMD_OK;
#else
    return SA_AMD_EINTERNAL;
#endif
}

static FirstCounts sort_first_counts(const SortScratch &ss, const Tuning &tn, int64_t count, bool keys32)
{
    FirstCounts f;
    if (onesweep_on(ss, tn)) {
        const OsShape &sh = keys32 ? os_shapes32[tn.onesweep32_shape] : os_shapes64[tn.onesweep64_shape];
        const int tile = sh.threads * sh.items;
        const OnesweepGeom g = onesweep_geom(count, tile);
        f.counts = ss.spine + OS_TICKETS; f.chunk_elems = g.seg_elems; f.G = g.nseg;
        f.zero_ptr = ss.spine; f.zero_bytes = (size_t)OS_ZONE * 4;
        return f;
    }
#ifdef SA_AMD_DIAG
    f.counts = ss.spine; f.zero_ptr = ss.spine;
    if (keys32) {
        const SortGrid32 g32 = sort_grid32(count, sort32_variants[tn.sort32_variant]);
        f.chunk_elems = g32.tiles_per_wg * g32.tile; f.G = g32.G;
    } else {
        const SortGrid g64 = sort_grid(count, sort_variants[tn.sort_variant]);
        f.chunk_elems = g64.tiles_per_wg * g64.tile; f.G = g64.G;
    }
    f.zero_bytes = (size_t)RADIX * f.G * 4;
    return f;
#else
    f.counts = ss.spine; f.zero_ptr = ss.spine; f.chunk_elems = count; f.G = 1; f.zero_bytes = (size_t)RADIX * 4;      // (not reached)
    return f;
#endif
}

// ------------------------------------------------------------------------------------------
// Bucket sort of the 32-bit first stage (kernels/bucket_sort.hpp): pairs grouped by their top 16 key bits (two stable
// global passes) -> pairs in the order of the whole 32-bit key, one workgroup per bucket, everything in LDS.
// ------------------------------------------------------------------------------------------
struct BkShape { int threads, items, minw; };
// (threads, pairs per thread, waves per SIMD the registers are held to); the first three are the ones in use -- the smallest that
// holds the largest bucket is taken --, the last is kept for A/B measurements (SA_AMD_BUCKET_SHAPE tries that one first)
static const BkShape bk_shapes[] = { { 256, 10, 6 }, { 512, 10, 8 }, { 1024, 10, 8 }, { 1024, 20, 1 }, { 256, 20, 1 } };
constexpr int N_BK_SHAPES = (int)(sizeof(bk_shapes) / sizeof(bk_shapes[0]));
constexpr int N_BK_DEFAULT = 4;
static int64_t bucket_cap(int shape) { return (int64_t)bk_shapes[shape].threads * bk_shapes[shape].items; }
static int64_t bucket_cap_max() { return bucket_cap(N_BK_DEFAULT - 1); }

// top_bits: 16 or 18 key bits that the global passes have ordered.  *done = false: some bucket is larger than every shape holds
// (nothing was written; the caller sorts the low bits globally).  words: two scratch words (largest bucket, error count);
// start: 2^top_bits + 1 words.  Read-back: the largest bucket.
// fin != nullptr: the tied suffixes are ordered by their low key bits in the same launch (what k_finish_sorted does in a pass of
// its own) when the shape has room to do it well (*fused); the caller has zeroed fin's bitmap and counters.
static int bucket_sort32(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out, int64_t count, int top_bits,
                         uint32_t *start, uint32_t *words, hipStream_t st, const Tuning &tn, bool *done, uint32_t *largest,
                         const BucketFinish *fin = nullptr, const KeyParams *P = nullptr, const KeySrc *K = nullptr, bool *fused = nullptr,
                         int val_extra = 0)      // the top val_extra bits of every value are the key bits below the 32 (ordered with them, stripped on the way out)
{
    *done = false; *largest = 0;
    if (fused) *fused = false;
    if (top_bits < 32 - BK_MAX_LBITS || top_bits > BK_TOP_BITS_MAX) return SA_AMD_EINTERNAL;
    const int lbits = 32 - top_bits;
    if (val_extra < 0 || lbits + val_extra > BK_MAX_LBITS) return SA_AMD_EINTERNAL;
    const uint32_t nb = 1u << top_bits;
    HIP_TRY(hipMemsetAsync(words, 0, 8, st));
    PROF(KC_MISC, nb, st, hipLaunchKernelGGL((k_bucket_starts), dim3((unsigned)ceil_div((int64_t)nb + 1, BK_STARTS_THREADS)), dim3(BK_STARTS_THREADS),
                                              0, st, keys_in, count, lbits, nb, start));
    PROF(KC_MISC, nb, st, hipLaunchKernelGGL((k_bucket_max), dim3((unsigned)ceil_div((int64_t)nb, BK_STARTS_THREADS)), dim3(BK_STARTS_THREADS), 0, st,
                                              (const uint32_t *)start, nb, words));
    uint32_t maxb = 0;
    { const int rcw = read_words(&maxb, words, 4, st); if (rcw) return rcw; }
    *largest = maxb;
    int shape = -1;
    if (tn.bucket_shape >= 0 && tn.bucket_shape < N_BK_SHAPES && bucket_cap(tn.bucket_shape) >= (int64_t)maxb) shape = tn.bucket_shape;
    for (int c = 0; shape < 0 && c < N_BK_DEFAULT; ++c)
        if (bucket_cap(c) >= (int64_t)maxb) shape = c;
    if (shape < 0) return SA_AMD_OK;
    // the 20-pairs-per-thread shapes leave one workgroup per CU (or three waves per SIMD): the tied suffixes' text look-ups have
    // nothing to hide behind there (1 GiB DNA: 7.7 + 7.5 ms as two kernels, 16.9 ms fused) -- k_finish_sorted follows instead
    const bool fuse = fin != nullptr && (bk_shapes[shape].items <= 10 || tn.bucket_finish_always);
    const BucketFinish F0 = BucketFinish();
    const KeyParams P0 = KeyParams();
    const KeySrc K0 = KeySrc();
#define BK_LAUNCH(T, I, W)                                                                                                               \
    do {                                                                                                                                 \
        if (fuse) PROF(KC_BUCKET, count, st, hipLaunchKernelGGL((k_bucket_sort<T, I, W, true>), dim3(nb), dim3(T), 0, st, keys_in, vals_in,         \
                                                                (const uint32_t *)start, lbits, keys_out, vals_out, words + 1, *fin, *P, *K, val_extra)); \
        else PROF(KC_BUCKET, count, st, hipLaunchKernelGGL((k_bucket_sort<T, I, W, false>), dim3(nb), dim3(T), 0, st, keys_in, vals_in,             \
                                                            (const uint32_t *)start, lbits, keys_out, vals_out, words + 1, F0, P0, K0, val_extra)); \
    } while (0)
    switch (shape) {
    case 0: BK_LAUNCH(256, 10, 6); break;
    case 1: BK_LAUNCH(512, 10, 8); break;
    case 2: BK_LAUNCH(1024, 10, 8); break;
    case 3: BK_LAUNCH(1024, 20, 1); break;
    default: BK_LAUNCH(256, 20, 1); break;
    }
#undef BK_LAUNCH
    *done = true;
    if (fused) *fused = fuse;
    return SA_AMD_OK;
}

#ifdef SA_AMD_DIAG
// ------------------------------------------------------------------------------------------
// Sample sort of the 64-bit stage (kernels/sample_sort.hpp): (key, i) pairs of keys_a[0 .. n) -> keys in order in keys_b, the
// suffixes in final_vals.  Scratch: keys_c (the sample and its sort), vals_a / vals_b (the values between the levels), u0 / u1
// (values of the sample's sort), big (n / 8 + 1 MiB bytes at least: the tiles' counts), small (2 MiB: totals, bases, segments,
// tile descriptors come behind), words (two counters + the list of reported buckets).
// *done = false: some bucket that is no equality bucket did not fit a workgroup (keys_a no longer holds the keys): the caller
// builds the keys again and sorts them with the LSD engine.  One read-back (the reported buckets).
// ------------------------------------------------------------------------------------------
static int64_t sample_count(int64_t n, const Tuning &tn)
{
    int lg = tn.sample_log ? tn.sample_log : (n >= ((int64_t)1 << 28) ? 22 : (n >= ((int64_t)1 << 27) ? 21 : 20));
    while (lg > 16 && ((int64_t)1 << lg) * 4 > n) --lg;
    return (int64_t)1 << lg;
}

static int sample_sort64(uint64_t *keys_a, uint64_t *keys_b, uint64_t *keys_c, uint32_t *vals_a, uint32_t *vals_b, uint32_t *u0, uint32_t *u1,
                         uint32_t *big, uint32_t *small, uint32_t *words, uint32_t *final_vals, int64_t n, int key_bits, const SortScratch &ss,
                         hipStream_t st, sa_amd_stats *local, const Tuning &tn, bool *done, bool trace)
{
    *done = false;
    const int64_t S = sample_count(n, tn);
    if (S < 65536 || n < 4 * S || n >= ((int64_t)1 << 32)) return SA_AMD_OK;
    // ---- the sample, sorted (its values are scratch) ----
    uint64_t *samp = keys_c, *samp_alt = keys_c + S;
    PROF(KC_MISC, S, st, hipLaunchKernelGGL((k_ss_sample), dim3((unsigned)ceil_div(S, 256)), dim3(256), 0, st, (const uint64_t *)keys_a, n, S, samp));
    SortResult sr;
    int rc = sort_pairs(samp, u0, samp_alt, u1, S, 0, key_bits, ss, nullptr, st, &sr, tn, true);
    if (rc) return rc;
    local->sort_passes += sr.passes; local->sorted_elements += (int64_t)sr.passes * S;
    const uint64_t *sample = sr.keys;
    // ---- scratch layout ----
    const int64_t tiles1 = ceil_div(n, SS_TILE), max_tiles2 = tiles1 + SS_WAYS;
    const int64_t per = ceil_div(tiles1, SS_CHUNKS);
    const int chunks = (int)ceil_div(tiles1, per);
    uint32_t *counts = big;                                            // level 1: tiles1 x 256, level 2: max_tiles2 x 512
    uint32_t *tot1 = small, *base1 = tot1 + SS_CHUNKS * SS_WAYS, *seg_start = base1 + SS_CHUNKS * SS_WAYS, *seg_first = seg_start + 320;
    uint32_t *tot2 = seg_first + 320, *bstart = tot2 + SS_WAYS * SS_IDS2;
    uint32_t *tile_seg = bstart + SS_BUCKETS + 64, *tile_base = tile_seg + ((max_tiles2 + 63) & ~(int64_t)63);
    // ---- level 1 ----
    PROF(KC_SS_COUNT, n, st, hipLaunchKernelGGL((k_ss_count<1>), dim3((unsigned)tiles1), dim3(SS_THREADS), 0, st, (const uint64_t *)keys_a, n, sample, S,
                                                (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, counts));
    PROF(KC_RR_SCAN, tiles1, st, hipLaunchKernelGGL((k_ss_scan_tiles<SS_WAYS>), dim3((unsigned)chunks), dim3(SS_WAYS), 0, st, counts, tiles1, per,
                                                    (const uint32_t *)nullptr, tot1));
    PROF(KC_RR_SCAN, chunks, st, hipLaunchKernelGGL((k_ss_bases1), dim3(1), dim3(SS_WAYS), 0, st, (const uint32_t *)tot1, chunks, base1, seg_start, seg_first));
    PROF(KC_RR_SCAN, max_tiles2, st, hipLaunchKernelGGL((k_ss_tiles), dim3(SS_WAYS + 1), dim3(256), 0, st, (const uint32_t *)seg_start, (const uint32_t *)seg_first,
                                                        max_tiles2, tile_seg, tile_base));
    PROF(KC_SS_SCATTER, n, st, hipLaunchKernelGGL((k_ss_scatter<1>), dim3((unsigned)tiles1), dim3(SS_THREADS), 0, st, (const uint64_t *)keys_a, (const uint32_t *)nullptr, n,
                                                  sample, S, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                                                  (const uint32_t *)counts, (const uint32_t *)base1, per, keys_b, vals_a));
    // ---- level 2 (the pairs of a segment stay inside it: keys_b -> keys_a) ----
    PROF(KC_SS_COUNT, n, st, hipLaunchKernelGGL((k_ss_count<2>), dim3((unsigned)max_tiles2), dim3(SS_THREADS), 0, st, (const uint64_t *)keys_b, n, sample, S,
                                                (const uint32_t *)tile_seg, (const uint32_t *)tile_base, (const uint32_t *)seg_start, counts));
    PROF(KC_RR_SCAN, max_tiles2, st, hipLaunchKernelGGL((k_ss_scan_tiles<SS_IDS2>), dim3(SS_WAYS), dim3(SS_IDS2), 0, st, counts, max_tiles2, (int64_t)0,
                                                        (const uint32_t *)seg_first, tot2));
    PROF(KC_RR_SCAN, SS_WAYS, st, hipLaunchKernelGGL((k_ss_bases2), dim3(SS_WAYS), dim3(SS_IDS2), 0, st, (const uint32_t *)tot2, (const uint32_t *)seg_start, bstart, (uint32_t)n));
    PROF(KC_SS_SCATTER, n, st, hipLaunchKernelGGL((k_ss_scatter<2>), dim3((unsigned)max_tiles2), dim3(SS_THREADS), 0, st, (const uint64_t *)keys_b, (const uint32_t *)vals_a, n,
                                                  sample, S, (const uint32_t *)tile_seg, (const uint32_t *)tile_base, (const uint32_t *)seg_start,
                                                  (const uint32_t *)counts, (const uint32_t *)bstart, (int64_t)1, keys_a, vals_b));
    // ---- level 3: every bucket in LDS (keys_a -> keys_b, values -> final_vals) ----
    HIP_TRY(hipMemsetAsync(words, 0, 16, st));
    if (!tn.sample_merge)
    PROF(KC_SS_BUCKET, n, st, hipLaunchKernelGGL((k_ss_bucket_sort<SB_SMALL_THREADS, SB_SMALL_ITEMS, 9, 0, false>), dim3((unsigned)SS_BUCKETS), dim3(SB_SMALL_THREADS), 0, st,
                                                 (const uint64_t *)keys_a, (const uint32_t *)vals_b, (const uint32_t *)bstart, keys_b, final_vals, words, words + 4));
    else
    PROF(KC_SS_BUCKET, n, st, hipLaunchKernelGGL((k_ss_bucket_merge<SB_SMALL_THREADS, SB_SMALL_ITEMS>), dim3((unsigned)SS_BUCKETS), dim3(SB_SMALL_THREADS), 0, st,
                                                 (const uint64_t *)keys_a, (const uint32_t *)vals_b, (const uint32_t *)bstart, keys_b, final_vals, words));
    PROF(KC_SS_BUCKET, 0, st, hipLaunchKernelGGL((k_ss_bucket_sort<SB_THREADS, SB_ITEMS, 10, SB_SMALL_CAP, true>), dim3((unsigned)SS_BUCKETS), dim3(SB_THREADS), 0, st,
                                                 (const uint64_t *)keys_a, (const uint32_t *)vals_b, (const uint32_t *)bstart, keys_b, final_vals, words, words + 4));
    uint32_t res[2] = { 0, 0 };
    { const int rcw = read_words(res, words, 8, st); if (rcw) return rcw; }
    local->sort_passes += 3; local->sorted_elements += 3 * n;
    if (trace) fprintf(stderr, "suffix_array_amd: sample sort: %lld samples, largest bucket %u (a workgroup holds %d), %u oversize buckets that are no equality buckets\n",
                       (long long)S, res[1], SB_CAP, res[0]);
    *done = res[0] == 0;
    return SA_AMD_OK;
}

#endif  // SA_AMD_DIAG

// symbol codes and key geometry from the sigma = 256 histogram; returns the number of key bits to sort
static int make_key_params(const uint32_t *hist, KeyParams *P, int *sigma_out, int kb_max = 64)
{
    int sigma = 0;
    for (int c = 0; c < 256; ++c) {
        if (hist[c]) P->code[c] = (uint8_t)sigma++;
        else P->code[c] = 0;
    }
    *sigma_out = sigma;
    P->packed = nullptr;
    P->gram = 0; P->gram_m = 0; P->gram_tail = 0; P->gram_top = 0; P->gram_D = 0; P->gram_table = nullptr;
    const uint64_t se = sigma > 2 ? (uint64_t)sigma : 2u;      // effective radix (a unary text still needs one bit)
    P->sigma = se;
    // kb_max < 64 (A/B): fewer key bits = fewer radix passes, more left to the rounds
    if ((se & (se - 1)) == 0) {                                // power of two: plain bit fields
        const int bits = bit_length(se - 1);
        P->bits = bits;
        P->k = kb_max / bits;
        const int used = P->k * bits;
        P->mask = used >= 64 ? ~0ull : ((1ull << used) - 1ull);
        P->top = 0;
        return used;
    }
    // otherwise pack as a base-sigma number: the largest k with sigma^k <= 2^64
    unsigned __int128 pw = 1;
    int k = 0;
    while (pw * se <= ((unsigned __int128)1 << kb_max)) { pw *= se; ++k; }
    P->bits = 0;
    P->k = k;
    P->mask = ~0ull;
    uint64_t top = 1;
    for (int i = 0; i + 1 < k; ++i) top *= se;
    P->top = top;
    const unsigned __int128 maxkey = pw - 1;                   // fits in 64 bits
    return bit_length((uint64_t)maxkey);
}

// Small device -> host read-backs (counts that steer the host loop) go through a pinned per-thread buffer:
// a 4-byte hipMemcpyAsync into pageable memory costs ~50-90 us per round trip, into pinned memory ~10.
// The buffer is a block of the process-wide pool: a short-lived worker thread (sa_amd_saca_batch) hands it back when it
// exits instead of paying hipHostMalloc / hipHostFree per call; a failed allocation is remembered, not retried per call.
struct PinnedWords {
    PinBlock b;
    bool failed = false;
    ~PinnedWords() { if (b.p) pool().release_pinned(b); }
};
static thread_local PinnedWords g_pinned;
static thread_local int g_readbacks = 0;        // blocking read-backs of the calling thread's current build (sa_amd_stats.readbacks)
static thread_local bool g_posted_off = false;  // SA_AMD_NO_POSTED_READBACK (set per build from the tuning)
static thread_local uint32_t g_post_seq = 0;

// A read-back as a POSTED write: one tiny kernel stores the words into the (mapped) pinned block, every 64-byte line tagged with
// a sequence number, and the host spins on the tags -- instead of a copy command plus hipStreamSynchronize, whose wake-up costs
// more than the kernel (measured, tools/readback_probe.hip: kernel + copy + synchronise 15.0 us, kernel + post kernel + spin
// 10.2 us, kernel + synchronise alone 11.4 us).  Line q of the block = [tag, words 15q .. 15q + 14]; the tag sits in the same
// line as the data it vouches for, so a line is either old or complete whatever the order the lines arrive in.
constexpr int POST_LINE = 16;
__global__ __launch_bounds__(256) void k_post_words(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int words, uint32_t seq)
{
    const int lines = (words + POST_LINE - 2) / (POST_LINE - 1);
    for (int i = threadIdx.x; i < lines * POST_LINE; i += 256) {
        const int q = i / POST_LINE, j = i % POST_LINE;
        if (j) { const int k = q * (POST_LINE - 1) + j - 1; dst[i] = k < words ? src[k] : 0u; }
    }
    __threadfence_system();
    __syncthreads();
    for (int q = threadIdx.x; q < lines; q += 256) __hip_atomic_store(dst + q * POST_LINE, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int read_words(void *dst, const void *dsrc, size_t bytes, hipStream_t st)     // bytes <= 3840, a multiple of 4; synchronises the stream
{
    ++g_readbacks;
    if (!g_pinned.b.p && !g_pinned.failed && pool().pinned(4096, -1, -1, &g_pinned.b) != SA_AMD_OK) g_pinned.failed = true;
    if (!g_pinned.b.p) {
        HIP_TRY(hipMemcpyAsync(dst, dsrc, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return SA_AMD_OK;
    }
    const int words = (int)(bytes / 4);
    const int lines = (words + POST_LINE - 2) / (POST_LINE - 1);
    if (!g_posted_off && (bytes & 3) == 0 && words > 0 && (size_t)lines * POST_LINE * 4 <= 4096) {
        uint32_t *host = (uint32_t *)g_pinned.b.p;
        if (++g_post_seq == 0) ++g_post_seq;
        const uint32_t seq = g_post_seq;
        // (the pool's pinned blocks are portable and mapped: the host address is valid on every device)
        hipLaunchKernelGGL(k_post_words, dim3(1), dim3(256), 0, st, (const uint32_t *)dsrc, host, words, seq);
        if (hipGetLastError() == hipSuccess) {
            const auto t0 = std::chrono::steady_clock::now();
            bool done = false, finished = false;
            for (unsigned spin = 0; !done; ++spin) {
                done = true;
                for (int q = lines - 1; q >= 0 && done; --q)
                    done = __atomic_load_n((volatile uint32_t *)(host + q * POST_LINE), __ATOMIC_ACQUIRE) == seq;
                if (done) break;
                if ((spin & 1023u) == 1023u) {
                    // the stream has drained and the tags are still not there (a second look after the query): the copy path decides
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    if (us > 200.0) {
                        if (finished) break;
                        const hipError_t q = hipStreamQuery(st);
                        if (q == hipSuccess) finished = true;
                        else if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
                    }
                }
            }
            if (done) {
                uint32_t *out = (uint32_t *)dst;
                for (int k = 0; k < words; ++k) out[k] = host[(k / (POST_LINE - 1)) * POST_LINE + 1 + k % (POST_LINE - 1)];
                return SA_AMD_OK;
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(g_pinned.b.p, dsrc, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(dst, g_pinned.b.p, bytes);
    return SA_AMD_OK;
}

// Binned ISA writes (one radix pass on the top bits of the suffix position, then a windowed scatter) pay off once the ISA is
// far larger than the caches: always for the full build (n entries), for a round's update only when it rewrites at least
// binned_min entries (measured on C3: direct stores of the CHANGED ranks win below 64 M pairs)
static bool binned(int64_t n, int64_t count, const Tuning &tn)
{
    if (tn.no_binned_isa) return false;
    if (tn.binned_isa_always) return count > 0;      // tests: exercise the path at small sizes
    return n >= ((int64_t)1 << 25) && (count >= n || count >= tn.binned_min);
}

// (suffix, rank) pairs (both 32 bits) -> ISA.  Large batches: two radix passes on the top 16 bits of the suffix position, then
// windows of the ISA are assembled in LDS and stored in address order (k_scatter_windows); smaller ones: one pass on the top
// 8 bits and a plain scatter inside 2^(nb-8)-entry windows.  SA_AMD_SCATTER_LEVELS = 1 / 2 forces either.
// iota: the value of pair i is i and pk is READ-ONLY (the suffix array itself): the passes then go pk -> (altk, altv) ->
// (altk2, altv2) and never write into pk.
static int scatter_binned(uint32_t *pk, uint32_t *pv, uint32_t *altk, uint32_t *altv, int64_t count, int64_t n,
                          const Workspace &w, hipStream_t st, sa_amd_stats *local, const Tuning &tn, bool iota = false,
                          uint32_t *altk2 = nullptr, uint32_t *altv2 = nullptr)
{
    const int nb = bit_length((uint64_t)(n > 1 ? n : 1));          // (the sentinel value n may be among the keys)
    int wlog = nb - 16;                                              // window = 2^wlog ISA entries, assembled in LDS
    if (wlog < 10) wlog = 10;                                        // small texts: one pass covers the bits above the window
    if (wlog > 15) wlog = 15;                                        // (n < 2^31, so nb <= 31)
    // (a text of fewer than 2^10 bytes has no bits above the window: the one-pass form below does it)
    const bool two = nb > wlog && (tn.scatter_levels == 2 || (tn.scatter_levels == 0 && count >= ((int64_t)1 << 25)));
    SortResult32 pr;
    if (two) {
        int rc;
        if (iota) {
            const int mid = wlog + RADIX_BITS < nb ? wlog + RADIX_BITS : nb;
            rc = sort_pairs32(pk, nullptr, altk, altv, count, wlog, mid, w.ss, nullptr, st, &pr, tn, true);
            if (rc) return rc;
            if (pr.passes != 1) return SA_AMD_EINTERNAL;            // (cannot happen: nb > wlog whenever count > 1)
            local->sort_passes += 1; local->sorted_elements += count;
            if (mid < nb) {
                rc = sort_pairs32(altk, altv, altk2, altv2, count, mid, nb, w.ss, nullptr, st, &pr, tn);
                if (rc) return rc;
                local->sort_passes += pr.passes; local->sorted_elements += (int64_t)pr.passes * count;
            }
        } else {
            rc = sort_pairs32(pk, pv, altk, altv, count, wlog, nb, w.ss, nullptr, st, &pr, tn);
            if (rc) return rc;
            local->sort_passes += pr.passes; local->sorted_elements += (int64_t)pr.passes * count;
        }
        const unsigned grid = (unsigned)ceil_div(count, SW_CHUNK);
        if (wlog <= 13)
            PROF(KC_SCATTER, count, st, hipLaunchKernelGGL((k_scatter_windows<13>), dim3(grid), dim3(SW_THREADS), 0, st, (const uint32_t *)pr.keys,
                                                           (const uint32_t *)pr.vals, w.isa, count, (uint32_t)n, wlog));
        else
            PROF(KC_SCATTER, count, st, hipLaunchKernelGGL((k_scatter_windows<15>), dim3(grid), dim3(SW_THREADS), 0, st, (const uint32_t *)pr.keys,
                                                           (const uint32_t *)pr.vals, w.isa, count, (uint32_t)n, wlog));
        return SA_AMD_OK;
    }
    const int shift = nb > RADIX_BITS ? nb - RADIX_BITS : 0;
    int rc = sort_pairs32(pk, pv, altk, altv, count, shift, shift + RADIX_BITS, w.ss, nullptr, st, &pr, tn, iota);
    if (rc) return rc;
    local->sort_passes += pr.passes; local->sorted_elements += (int64_t)pr.passes * count;
    PROF(KC_SCATTER, count, st, hipLaunchKernelGGL((k_scatter_pairs<uint32_t>), dim3((unsigned)ceil_div(count, 1024)), dim3(256), 0, st,
                                                   (const uint32_t *)pr.keys, (const uint32_t *)pr.vals, w.isa, count, (uint32_t)n));
    return SA_AMD_OK;
}

struct Refined { const uint64_t *keys; const uint32_t *vals; uint32_t *vnext; int64_t m_global; };   // m_global: members ordered by the global sort

// Read-backs of a refinement round (DeviceBuild::doubling_rounds).  A round used to block twice: once in the middle for the number
// of members the local pass could not order (they go through the global sort before the re-rank) and once at its end for the
// number still tied.  From the second round of a kind on nearly every round has NO such members, so the caller may ask
// refine_list to DEFER the first question: the local pass and its counting kernels are launched, the count stays on the device
// (w.total[RC_FLAGGED]) and the re-rank kernels behind it are launched GATED on it -- they do nothing when it is not zero.
// One read-back at the end of the round then answers both; in the rare round that did have such members the caller calls
// refine_list again with the words it read (resume_tot), which runs the global sort of those members, and launches the
// re-rank kernels ungated.
constexpr int RC_FLAGGED = 16, RC_GROUPS = 8, RC_BIG_LISTED = 12, RC_BIG_ORDERED = 13, RC_WORDS = 17;      // words of w.total
constexpr int RC_BIG_LISTED_SAVED = 14, RC_BIG_ORDERED_SAVED = 15;     // ... where k_rr_scan_round puts the two big-group counters before it zeroes them for the next round
struct RoundCtl {
    bool defer = false;                    // in: do not read the local pass's counts back, return right behind its launches
    const uint32_t *resume_tot = nullptr;  // in: the counts (RC_WORDS words of w.total) of a deferred call that did have flagged members
    bool skip_big = false;                 // in: no group can be larger than GS_CAP any more: k_group_sort_big is not launched
    bool counters_clear = false;           // in: the previous round's k_rr_scan_round has zeroed w.total[RC_BIG_LISTED .. +1]
    bool deferred = false;                 // out: the counts were left on the device
    int64_t m_flagged = -1;                // out: members the local pass left to the global sort (-1: no local pass ran)
    int64_t big_listed = -1;               // out: groups listed for k_group_sort_big (-1: not looked for)
};

// Three-way split of giant groups around their majority key (kernels/refine.hpp, k_split_*): the keys in rkA carry the dense
// group index above bit kb, w.ft_cnt the exclusive group-start counts per tile.  *taken = false: the count pass found more
// than an eighth of the members off their group's pivot key (or the scratch buffers too small) -- nothing has been changed,
// the caller sorts the list with the radix sort.  scratchU / scratchG: two free 4n-byte buffers.
static int split_giant_groups(uint64_t *rkA, uint64_t *rkB, uint32_t *Vcur, uint32_t *Valt, const uint32_t *Ucur, const uint32_t *Gcur,
                              uint32_t *scratchU, uint32_t *scratchG, int64_t m, int64_t n, uint32_t groups, int kb, int sort_bits,
                              const Workspace &w, hipStream_t st, sa_amd_stats *local, Refined *out, const Tuning &tn, bool *taken,
                              bool starts_ready)            // the gather has written the table of group starts (first array in scratchG)
{
    *taken = false;
    const int64_t tiles = ceil_div(m, RR_TILE);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    char *sg = (char *)scratchG;
    uint32_t *starts = (uint32_t *)sg;            sg += up(((size_t)groups + 1) * 4);     // list index of every group's first member (+ m)
    uint32_t *mps = (uint32_t *)sg;               sg += up(((size_t)groups + 1) * 4);     // minority members in front of it (+ their total)
    uint32_t *Lless = (uint32_t *)sg;             sg += up((size_t)groups * 4);           // minority members of the group below its pivot
    uint64_t *pivot = (uint64_t *)sg;             sg += up((size_t)groups * 8);
    const size_t cap = (size_t)m / 8 + 1;                                                  // (minority members when the split is taken)
    uint32_t *mv = (uint32_t *)sg, *mv_alt = mv + ((cap + 63) & ~(size_t)63);
    uint64_t *mk = (uint64_t *)scratchU, *mk_alt = mk + ((cap + 31) & ~(size_t)31);
    const size_t need_g = (size_t)(sg - (char *)scratchG) + 2 * ((cap + 63) & ~(size_t)63) * 4;
    const size_t need_u = 2 * ((cap + 31) & ~(size_t)31) * 8;
    if (need_g > (size_t)n * 4 || need_u > (size_t)n * 4) return SA_AMD_OK;
    if (!starts_ready)
        PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_group_starts), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, Ucur, Gcur, m,
                                                    (const uint32_t *)w.ft_cnt, groups, starts));
    PROF(KC_MISC, groups, st, hipLaunchKernelGGL((k_split_pivots), dim3((unsigned)ceil_div((int64_t)groups, 256)), dim3(256), 0, st,
                                             (const uint64_t *)rkA, (const uint32_t *)starts, groups, kb, pivot));
    PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_split_count), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, (const uint64_t *)rkA, m, kb,
                                                (const uint64_t *)pivot, w.tcnt));
    PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
    uint32_t minor = 0;
    { const int rcw = read_words(&minor, w.total, 4, st); if (rcw) return rcw; }
    if ((int64_t)minor * 8 > m) return SA_AMD_OK;
    *taken = true;
    out->m_global = m;
    if (minor == 0) {                                  // every member carries its group's pivot key: the order stands
        out->keys = rkA; out->vals = Vcur; out->vnext = Valt;
        return SA_AMD_OK;
    }
    PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_split_pass<false>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                (const uint64_t *)rkA, (const uint32_t *)Vcur, m, kb, (const uint64_t *)pivot,
                                                (const uint32_t *)w.tcnt, (const uint32_t *)starts, groups, mps, (const uint32_t *)w.total,
                                                mk, mv, (const uint32_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr));
    SortResult s2;
    const int rc = sort_pairs(mk, mv, mk_alt, mv_alt, minor, 0, sort_bits, w.ss, nullptr, st, &s2, tn);
    if (rc) return rc;
    local->sort_passes += s2.passes; local->sorted_elements += (int64_t)s2.passes * minor;
    PROF(KC_MISC, groups, st, hipLaunchKernelGGL((k_split_less), dim3((unsigned)ceil_div((int64_t)groups, 256)), dim3(256), 0, st,
                                             (const uint64_t *)s2.keys, (const uint32_t *)mps, (const uint64_t *)pivot, groups, kb, Lless));
    PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_split_pass<true>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                (const uint64_t *)rkA, (const uint32_t *)Vcur, m, kb, (const uint64_t *)pivot,
                                                (const uint32_t *)w.tcnt, (const uint32_t *)starts, groups, mps, (const uint32_t *)w.total,
                                                (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)Lless, rkB, Valt));
    PROF(KC_SCATTER, minor, st, hipLaunchKernelGGL((k_split_place_minor), dim3((unsigned)ceil_div((int64_t)minor, 256)), dim3(256), 0, st,
                                                   (const uint64_t *)s2.keys, (const uint32_t *)s2.vals, (int64_t)minor, kb,
                                                   (const uint32_t *)starts, (const uint32_t *)mps, (const uint32_t *)Lless, rkB, Valt));
    out->keys = rkB; out->vals = Valt; out->vnext = Vcur;
    return SA_AMD_OK;
}

// One refinement round of the tied list with a secondary key taken from the text (KeySrc): afterwards every
// group is ordered by (group head << kb) | key2.  Small groups: gather fused with the in-LDS group sort
// (k_group_sort); groups no tile owns, or everything when *local_ok is off: plain gather + global radix sort.
// scratchU / scratchG: two free 4n-byte buffers.
static int refine_list(uint64_t *rkA, uint64_t *rkB, uint32_t *Vcur, uint32_t *Valt, const uint32_t *Ucur, const uint32_t *Gcur,
                       uint32_t *scratchU, uint32_t *scratchG, int64_t m, int64_t n, const uint8_t *dT, const KeyParams &P,
                       const KeySrc &K_in, int g_bits, bool *local_ok, const Workspace &w, hipStream_t st, sa_amd_stats *local,
                       Refined *out, const Tuning &tn, bool retry_local = false, int *split_rest = nullptr,   // split_rest: rounds the three-way split sits out after its count pass found no majority
                       RoundCtl *ctl = nullptr)
{
    KeySrc K = K_in;
    K.net_min = tn.network_min;
    const bool resume = ctl && ctl->resume_tot;
    const int64_t tiles = ceil_div(m, RR_TILE);
    const int kb = K.kb;
    SortResult sr;
    int rc;
    // The local pass was given up because (nearly) every member sat in a group no tile can own.  Lists large enough for the
    // dense group index count their groups anyway: when the average group has come down to about what a tile can own
    // (a Fibonacci word's groups shrink while the list does not), the local pass is tried again -- in this round
    bool counted = false;
    uint32_t groups = 0;
    if (!*local_ok && retry_local && !tn.no_local_sort && m < tn.dense_rekey_min) *local_ok = true;     // (small lists do not count their groups)
    if (!*local_ok && retry_local && !tn.no_local_sort && m >= tn.dense_rekey_min) {
        PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_flag_count), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                    (const uint8_t *)nullptr, Ucur, Gcur, m, w.tcnt, w.ft_cnt));
        PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.ft_cnt, w.thead, tiles, w.total + 8));
        { const int rcw = read_words(&groups, w.total + 8, 4, st); if (rcw) return rcw; }
        counted = true;
        if ((int64_t)groups * 2 * tn.group_cap >= m) *local_ok = true;
    }
    const bool had_local_pass = *local_ok;            // (then the keys are in rkA already when the whole list goes through the global sort after all)
    if (*local_ok) {
        uint8_t *flags = (uint8_t *)scratchG;
        const unsigned gs_blocks = (unsigned)ceil_div(m, GS_TILE);
        const int cap = tn.group_cap;                    // largest group ordered in LDS (C3: 1024 beats 512 by 1%)
        // groups of up to GB_CAP (8 192) members whose keys fit 32 bits: one workgroup each, in LDS (k_group_sort_big), on the list of
        // their first members that k_group_sort writes (scratchU is free until k_flag_gather); the rest goes through the global sort
        const bool big_local = !tn.no_big_group_sort && kb <= 32 && m > GS_CAP && !(ctl && ctl->skip_big);
        uint32_t *bheads = big_local ? scratchU : (uint32_t *)nullptr, *bcount = big_local ? w.total + RC_BIG_LISTED : (uint32_t *)nullptr;
        if (!resume) {
        if (!(ctl && ctl->counters_clear))
            HIP_TRY(hipMemsetAsync(w.total + RC_BIG_LISTED, 0, 8, st));           // [12] listed first members, [13] members ordered by k_group_sort_big
        if (K.mode == KS_TEXT)
            PROF(KC_LOCAL, m, st, hipLaunchKernelGGL((k_group_sort<KS_TEXT>), dim3(gs_blocks), dim3(GS_THREADS), 0, st, (const uint32_t *)Vcur,
                                                     Gcur, Ucur, dT, P, m, n, K, rkA, Vcur, flags, cap, bheads, bcount));
        else if (K.mode == KS_LOWKEY)
            PROF(KC_LOCAL, m, st, hipLaunchKernelGGL((k_group_sort<KS_LOWKEY>), dim3(gs_blocks), dim3(GS_THREADS), 0, st, (const uint32_t *)Vcur,
                                                     Gcur, Ucur, dT, P, m, n, K, rkA, Vcur, flags, cap, bheads, bcount));
        else if (K.mode == KS_RANK)
            PROF(KC_LOCAL, m, st, hipLaunchKernelGGL((k_group_sort<KS_RANK>), dim3(gs_blocks), dim3(GS_THREADS), 0, st, (const uint32_t *)Vcur,
                                                     Gcur, Ucur, dT, P, m, n, K, rkA, Vcur, flags, cap, bheads, bcount));
        else if (K.mode == KS_CHASE)
            PROF(KC_LOCAL, m, st, hipLaunchKernelGGL((k_group_sort<KS_CHASE>), dim3(gs_blocks), dim3(GS_THREADS), 0, st, (const uint32_t *)Vcur,
                                                     Gcur, Ucur, dT, P, m, n, K, rkA, Vcur, flags, cap, bheads, bcount));
        else {
            // sparse look-up: its own kernel, one suffix per thread (a chain of ~60 dependent loads each), then the sort on those keys
            int64_t gblocks = ceil_div(m, GK_THREADS);
            if (gblocks > 8192) gblocks = 8192;
            PROF(KC_GATHER, m, st, hipLaunchKernelGGL((k_gather_textkey<KS_SPARSE>), dim3((unsigned)gblocks), dim3(GK_THREADS), 0, st,
                                                      (const uint32_t *)Vcur, Gcur, dT, P, m, n, K, rkA));
            PROF(KC_LOCAL, m, st, hipLaunchKernelGGL((k_group_sort<KS_PRE>), dim3(gs_blocks), dim3(GS_THREADS), 0, st, (const uint32_t *)Vcur,
                                                     Gcur, Ucur, dT, P, m, n, K, rkA, Vcur, flags, cap, bheads, bcount));
        }
        if (gs_blocks > 1)
            PROF(KC_LOCAL, 0, st, hipLaunchKernelGGL((k_group_sort_straddle), dim3(gs_blocks - 1), dim3(GX_THREADS), 0, st, rkA, Vcur, Gcur,
                                                     Ucur, m, flags, cap, K, n));
        if (big_local) {
            const int lo = cap > GS_CAP ? cap : GS_CAP;
            PROF(KC_LOCAL, 0, st, hipLaunchKernelGGL((k_group_sort_big<256>), dim3((unsigned)(8 * cu_count())), dim3(256), 0, st, rkA, Vcur, Gcur, m, kb, lo,
                                                     (const uint32_t *)bheads, (const uint32_t *)bcount, flags, w.total + 13));
            PROF(KC_LOCAL, 0, st, hipLaunchKernelGGL((k_group_sort_big<512>), dim3((unsigned)(4 * cu_count())), dim3(512), 0, st, rkA, Vcur, Gcur, m, kb,
                                                     lo > GB_CAP_SMALL ? lo : GB_CAP_SMALL, (const uint32_t *)bheads, (const uint32_t *)bcount, flags, w.total + 13));
        }
        PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_flag_count), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                    (const uint8_t *)flags, Ucur, Gcur, m, w.tcnt, w.ft_cnt));
        // (flagged members -> w.total[RC_FLAGGED], flagged groups -> w.total[RC_GROUPS]; one launch)
        PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan_pair), dim3(2), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, w.total + RC_FLAGGED,
                                                       w.ft_cnt, (uint32_t *)nullptr, w.total + RC_GROUPS, tiles));
        }
        if (ctl && ctl->defer && !resume) {
            // the counts stay on the device: the caller launches the re-rank kernels gated on w.total[RC_FLAGGED] and reads everything at once
            ctl->deferred = true;
            out->keys = rkA; out->vals = Vcur; out->vnext = Valt; out->m_global = 0;
            return SA_AMD_OK;
        }
        uint32_t tot9[RC_WORDS] = { 0 };                      // [RC_FLAGGED] flagged members, [RC_GROUPS] flagged groups, [RC_BIG_ORDERED] members k_group_sort_big ordered
        if (resume) memcpy(tot9, ctl->resume_tot, sizeof(tot9));
        else {
            const int rcw = read_words(tot9, w.total, sizeof(tot9), st); if (rcw) return rcw;
            }
        const int64_t m_big = tot9[RC_FLAGGED], m_big_local = tot9[RC_BIG_ORDERED];
        if (ctl) { ctl->m_flagged = m_big; ctl->big_listed = big_local ? (int64_t)tot9[RC_BIG_LISTED] : -1; }
        // the flagged members are sorted in the first `half` entries of rkB / Valt with the second half as the alternate
        // buffers: half is even (16-byte aligned 8-byte keys) and half + m_big never exceeds the n entries the slabs hold
        const size_t half = ((size_t)n / 2) & ~(size_t)1;
        if ((size_t)m_big <= half && half + (size_t)m_big <= (size_t)n) {
            if (m_big > 0) {
                // groups no tile owns: global sort of (index of the group among them, key2), then back to their list positions
                // (a text that is one long run has ONE such group: no index bits at all, four passes instead of eight)
                const int idx_bits = bit_length((uint64_t)(tot9[RC_GROUPS] > 0 ? tot9[RC_GROUPS] - 1 : 0));
                PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_flag_gather), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                            (const uint8_t *)flags, (const uint64_t *)rkA, (const uint32_t *)Vcur, Ucur, Gcur, m,
                                                            (const uint32_t *)w.tcnt, (const uint32_t *)w.ft_cnt, kb, rkB, Valt, scratchU));
                rc = sort_pairs(rkB, Valt, rkB + half, Valt + half, m_big, 0, kb + idx_bits, w.ss, nullptr, st, &sr, tn, false, true);
                if (rc) return rc;
                local->sort_passes += sr.passes; local->sorted_elements += (int64_t)sr.passes * m_big;
                PROF(KC_SCATTER, m_big, st, hipLaunchKernelGGL((k_scatter_back), dim3((unsigned)ceil_div(m_big, 256)), dim3(256), 0, st,
                                                               (const uint64_t *)sr.keys, (const uint32_t *)sr.vals,
                                                               (const uint32_t *)scratchU, Gcur, kb, m_big, rkA, Vcur));
            }
            out->keys = rkA; out->vals = Vcur; out->vnext = Valt; out->m_global = m_big + m_big_local;      // (neither kind has been chased)
            local->locally_sorted += m - m_big;
            if (m_big * 2 > m) *local_ok = false;           // mostly large groups: not worth another local pass
            return SA_AMD_OK;
        }
        if (m_big * 10 >= m * 9) *local_ok = false;     // (nearly) the whole list sits in groups no tile can own (runs, periodic texts): the next rounds skip the local pass
    }
    // the whole list through the global sort.  Large lists are keyed by (index of the group in the list, key2) instead of
    // (28-bit slot of the group head, key2) when that saves radix passes: count the group heads first, then gather the keys in
    // that form (k_gather_keyed) or re-key the ones the local pass left (k_rekey_dense).  The head slots are not put back
    // after the sort: the re-rank kernels only compare neighbouring keys.
    int sort_bits = kb + g_bits;
    bool rekeyed = false;
    if (m >= tn.dense_rekey_min) {
        if (!counted || had_local_pass) {              // (the local pass has used the count arrays for its own compaction)
            PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_flag_count), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                        (const uint8_t *)nullptr, Ucur, Gcur, m, w.tcnt, w.ft_cnt));
            PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.ft_cnt, w.thead, tiles, w.total + 8));
            { const int rcw = read_words(&groups, w.total + 8, 4, st); if (rcw) return rcw; }
            }
        const int idx_bits = bit_length((uint64_t)(groups > 0 ? groups - 1 : 0));
        if (ceil_div(kb + idx_bits, RADIX_BITS) < ceil_div(kb + g_bits, RADIX_BITS)) {
            sort_bits = kb + idx_bits;
            rekeyed = true;
        }
    }
    const bool split_wanted = rekeyed && !tn.no_split && m >= tn.split_min && groups > 0 && (int64_t)groups * tn.split_group_min <= m &&
                              !(split_rest && *split_rest > 0);
    bool starts_ready = false;
    if (had_local_pass) {
        if (rekeyed)
            PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rekey_dense), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, rkA, Ucur, Gcur, m,
                                                        (const uint32_t *)w.ft_cnt, kb));
    } else if (K.mode == KS_SPARSE) {
        // (a chain of ~60 dependent loads per suffix: one suffix per thread)
        int64_t gblocks = ceil_div(m, GK_THREADS);
        if (gblocks > 8192) gblocks = 8192;
        PROF(KC_GATHER, m, st, hipLaunchKernelGGL((k_gather_textkey<KS_SPARSE>), dim3((unsigned)gblocks), dim3(GK_THREADS), 0, st,
                                                  (const uint32_t *)Vcur, Gcur, dT, P, m, n, K, rkA));
        if (rekeyed)
            PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rekey_dense), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, rkA, Ucur, Gcur, m,
                                                        (const uint32_t *)w.ft_cnt, kb));
    } else {
        const uint32_t *th = rekeyed ? (const uint32_t *)w.ft_cnt : (const uint32_t *)nullptr;
        // (when the three-way split may follow, the gather also writes its table of group starts: the first array in scratchG)
        uint32_t *gs = split_wanted ? scratchG : (uint32_t *)nullptr;
        starts_ready = gs != nullptr;
        if (K.mode == KS_TEXT)
            PROF(KC_GATHER, m, st, hipLaunchKernelGGL((k_gather_keyed<KS_TEXT>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                      (const uint32_t *)Vcur, Ucur, Gcur, dT, P, m, n, K, th, rkA, gs, groups));
        else if (K.mode == KS_LOWKEY)
            PROF(KC_GATHER, m, st, hipLaunchKernelGGL((k_gather_keyed<KS_LOWKEY>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                      (const uint32_t *)Vcur, Ucur, Gcur, dT, P, m, n, K, th, rkA, gs, groups));
        else
            PROF(KC_GATHER, m, st, hipLaunchKernelGGL((k_gather_keyed<KS_RANK>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                      (const uint32_t *)Vcur, Ucur, Gcur, dT, P, m, n, K, th, rkA, gs, groups));
    }
    // Giant groups (runs, periodic texts, long repeats): all but a few members of a group carry the same key, so the few
    // are pulled out and sorted on their own and the rest only shifts (split_giant_groups) -- if its count pass finds that
    // they are few indeed; otherwise the radix sort below.
    if (split_rest && *split_rest > 0) --*split_rest;
    else if (split_wanted) {
        bool taken = false;
        rc = split_giant_groups(rkA, rkB, Vcur, Valt, Ucur, Gcur, scratchU, scratchG, m, n, groups, kb, sort_bits, w, st, local, out, tn, &taken,
                                starts_ready);
        if (rc || taken) return rc;
        if (split_rest) *split_rest = 3;              // (a Fibonacci word's groups fall into parts of similar size round after round)
    }
    rc = sort_pairs(rkA, Vcur, rkB, Valt, m, 0, sort_bits, w.ss, nullptr, st, &sr, tn, false, true);
    if (rc) return rc;
    local->sort_passes += sr.passes; local->sorted_elements += (int64_t)sr.passes * m;
    out->keys = sr.keys; out->vals = sr.vals; out->m_global = m;
    out->vnext = (sr.vals == Vcur) ? Valt : Vcur;
    return SA_AMD_OK;
}

// Gram keys: how many of the sigma^g possible g-grams occur?  A word-structured text uses a small part of them, so a key of
// dense gram ranks holds more symbols than the base-sigma form, often in fewer digits (C3, sigma = 57: 12 symbols in 7
// passes instead of 10 in 8).  Measured (flags, count, scan), then decided: *P and *key_bits change only when the gram form
// holds more symbols, or as many in fewer radix passes.
static int choose_gram_keys(const uint8_t *dT, int64_t n, KeyParams *Pp, int *key_bits, const Workspace &w, hipStream_t st,
                            const Tuning &tn, bool trace)
{
    KeyParams &P = *Pp;
    const uint64_t se = P.sigma;
    const uint64_t cap = (uint64_t)((size_t)n < GRAM_MAX_ENTRIES ? (size_t)n : GRAM_MAX_ENTRIES);
    int g = 0;
    uint64_t S = 1;
    for (int t = 1; t <= 8; ++t) {
        if (S * se > cap) break;
        S *= se; g = t;
        if (tn.gram_g >= 2 && g == tn.gram_g) break;
    }
    if (g < 2) return SA_AMD_OK;
    uint64_t top = 1;
    for (int t = 0; t + 1 < g; ++t) top *= se;
    const int64_t gtiles = ceil_div((int64_t)S, GT_TILE);
    HIP_TRY(hipMemsetAsync(w.gram_flags, 0, (size_t)S, st));
    PROF(KC_MISC, n, st, hipLaunchKernelGGL((k_gram_mark), dim3((unsigned)ceil_div(ceil_div(n, KB_TILE), GM_TPW)), dim3(KB_THREADS), 0, st, dT, n, P, g,
                                            (uint32_t)top, w.gram_flags));
    PROF(KC_MISC, (int64_t)S, st, hipLaunchKernelGGL((k_gram_count), dim3((unsigned)gtiles), dim3(GT_THREADS), 0, st,
                                            (const uint8_t *)w.gram_flags, (int64_t)S, w.tcnt));
    PROF(KC_RR_SCAN, gtiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, gtiles, w.total));
    uint32_t D = 0;
    { const int rcw = read_words(&D, w.total, 4, st); if (rcw) return rcw; }
    int m = 0;
    unsigned __int128 pw = 1;
    while (D >= 2 && (m + 1) * g <= 64 && pw * D <= ((unsigned __int128)1 << 64)) { pw *= D; ++m; }
    // the bits a whole further gram does not fit into take plain symbols (base-sigma digits behind the gram ranks): C3's
    // 3 four-grams leave 9.7 of the 64 bits idle, one symbol of 5.8 bits makes the key 13 symbols deep for one more pass
    int tail = 0;
    while (m > 0 && tail < tn.gram_tail_max && m * g + tail + 1 <= 64 && pw * se <= ((unsigned __int128)1 << 64)) { pw *= se; ++tail; }
    const int gram_bits = m > 0 ? bit_length((uint64_t)(pw - 1)) : 0;
    const int plain_passes = (int)ceil_div(*key_bits, RADIX_BITS), gram_passes = (int)ceil_div(gram_bits, RADIX_BITS);
    const int syms = m * g + tail;
    const bool better = m > 0 && (syms > P.k || (syms == P.k && gram_passes < plain_passes));
    if (trace) fprintf(stderr, "suffix_array_amd: gram keys: %u of %llu %d-grams occur -> %d symbols (%d grams + %d plain) in %d bits (plain: %d in %d) -> %s\n",
                       D, (unsigned long long)S, g, syms, m, tail, gram_bits, P.k, *key_bits, better ? "gram keys" : "plain keys");
    if (!better) return SA_AMD_OK;
    PROF(KC_MISC, (int64_t)S, st, hipLaunchKernelGGL((k_gram_table), dim3((unsigned)gtiles), dim3(GT_THREADS), 0, st,
                                            (const uint8_t *)w.gram_flags, (int64_t)S, (const uint32_t *)w.tcnt, w.gram_table));
    P.gram = g; P.gram_m = m; P.gram_tail = tail; P.gram_top = (uint32_t)top; P.gram_D = D; P.gram_table = w.gram_table;
    P.bits = 0; P.k = syms; P.mask = ~0ull;
    *key_bits = gram_bits;
    return SA_AMD_OK;
}

// Early download (host/host_path.hpp): what the host-pointer entry points hand to the build so that the suffix array can start
// its way over PCIe while the last refinement rounds run.  The build calls start() once, when the tied list has come down to
// `threshold` suffixes -- by then it has marked the snapshot of the tied slots (k_early_mark) and written SA[0] --, and stop()
// when the rounds are done; stop() returns how many entries of the array (from its front) were or are being copied, and the
// build compacts the final values of the marked entries among them (d_holes, in entry order; d_tile_off = where each tile of
// EARLY_TILE entries starts in d_holes; d_bits = the marks).
struct EarlyDownload {
    int off = 0;                          // in: entry of slot 0 in the downloaded array (1: the array starts with the sentinel entry SA[0] = n)
    int64_t threshold = 0;                // in: start when at most this many suffixes are tied (0: never)
    std::function<void(hipEvent_t)> start;   // in: begin copying the array from its front; the copy stream must wait for the event first
    std::function<int64_t()> stop;        // in: no further chunk is started; returns the entries [0, x) that were handed to the copy
    bool started = false;
    int64_t m_snap = 0;                   // out: tied suffixes at the snapshot
    int64_t covered = 0;                  // out: stop()'s answer
    int64_t holes = 0;                    // out: marked entries among the covered ones
    const uint32_t *d_bits = nullptr, *d_tile_off = nullptr, *d_holes = nullptr;      // out (device pointers into the workspace)
    hipEvent_t ev = nullptr;
};

// One device-resident build, phase by phase.  The members are what the phases hand to each other; every phase returns an
// SA_AMD_* status.  The blocking 4-byte read-backs that steer the host (how many suffixes are still tied, how many groups,
// how many ranks changed) are the read_words calls inside the phases: each one names what it reads.
struct DeviceBuild {
    // ---- inputs ----
    const uint8_t *dT;
    uint32_t *dSA, *SA;                 // SA = dSA + 1: the n sorted suffixes behind the sentinel slot
    int64_t n;
    hipStream_t st;
    Tuning tn;
    Workspace w;
    sa_amd_stats local;
    EarlyDownload *early = nullptr;     // host-pointer callers: the array starts travelling before the last rounds are done
    bool trace = false;
    double trace_t = 0;
    double lap() { const double t = now_ms(), d = t - trace_t; trace_t = t; return d; }
    bool timing_only() const            // diag library only: stop after the initial sort (array NOT finished)
    {
#ifdef SA_AMD_DIAG
        return tn.timing_only_initial_sort;
#else
        return false;
#endif
    }
    // ---- key geometry and route (geometry_and_probes) ----
    KeyParams P, Ptext;
    int sigma = 0, key_bits = 0, g_bits = 0, top_shift = 0;
    bool force_dense = false, text_ok = false, local_ok = false, probe_dense = false;
    // ---- initial order (initial_sort) ----
    SortResult sr;
    const uint32_t *sorted32 = nullptr; // top-32 stage: the sorted 32-bit keys (no 64-bit sorted array exists)
    int bucket_top_bits = 0;            // key bits the two global passes in front of the bucket sort order (0: four global passes)
    bool flat_text = false;             // short text whose byte values are equally frequent: narrower initial keys (geometry_and_probes)
    bool bucket_finished = false;       // ... and k_bucket_sort has already ordered the suffixes tied on those 32 bits by their low key bits
    uint64_t *sorted0 = nullptr;        // the initial keys in SA order (kept for the rank look-ups)
    // ---- the tied list and its buffers (from first_round_from_sorted_keys on) ----
    uint32_t *Ucur = nullptr, *Unext = nullptr, *Gcur = nullptr, *Gnext = nullptr, *Vcur = nullptr;
    uint64_t *rkA = nullptr, *rkB = nullptr;
    int64_t tiles = 0, m = 0, depth = 0;
    uint32_t m32 = 0;
    bool lists_ready = false, finished32 = false, fused64 = false, sparse = false;
    bool isa_tail_ranks = false;        // the rank set-up of the dense route wrote tail ranks (k_rr_apply FTAIL)
    int s_sym = 0, tkb = 0, key2_bits = 0;
    bool unary_done = false;            // a text of one byte value: the array was written directly (geometry_and_probes), nothing else runs
    bool prev_clean = true;             // the last refinement round's local pass ordered every member (optimistic for the first one: a miss costs three empty launches)
    int deferred_misses = 0;            // rounds that deferred their mid-round read-back and did have members for the global sort (RoundCtl)


    // Early download: called wherever (Ucur, m) is the current tied list and every slot outside it is final
    int early_maybe_start()
    {
        if (!early || early->started || m <= 0 || m > early->threshold || !early->start) return SA_AMD_OK;
        if (!early->ev && hipEventCreateWithFlags(&early->ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return SA_AMD_OK; }
        hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, dSA, (uint32_t)n);   // reference src/saca.rs:13 (the first chunk carries it)
        LAUNCH_CHECK(st);
        HIP_TRY(hipMemsetAsync(w.early_bits, 0, (((size_t)n + 1 + 31) / 32 + EARLY_THREADS) * 4, st));
        int64_t blocks = ceil_div(m, 256 * 8);
        if (blocks > 65536) blocks = 65536;
        if ((((uintptr_t)Ucur) & 15) != 0) return SA_AMD_OK;       // (the list slabs are 256-byte aligned: cannot happen)
        PROF(KC_MISC, m, st, hipLaunchKernelGGL((k_early_mark), dim3((unsigned)blocks), dim3(256), 0, st, (const uint32_t *)Ucur, m, (uint32_t)early->off, w.early_bits));
        HIP_TRY(hipEventRecord(early->ev, st));
        early->started = true;
        early->m_snap = m;
        if (trace) fprintf(stderr, "suffix_array_amd: early download starts with %lld suffixes still tied\n", (long long)m);
        early->start(early->ev);
        return SA_AMD_OK;
    }
    // ... and when the array is complete: the final values of the marked entries among those the copy has taken
    int early_finish()
    {
        if (!early || !early->started) return SA_AMD_OK;
        early->covered = early->stop ? early->stop() : 0;
        if (early->covered > n + early->off) early->covered = n + early->off;
        early->holes = 0;
        if (early->covered <= 0) return SA_AMD_OK;
        const int64_t tiles_e = ceil_div(early->covered, EARLY_TILE);
        const int64_t words = ceil_div(early->covered, 32);
        // (bits beyond the covered entries inside the last word belong to entries that are downloaded later: harmless, they are
        // patched with their final values too when the host walks whole words -- the host stops at `covered`)
        PROF(KC_MISC, early->covered, st, hipLaunchKernelGGL((k_early_count), dim3((unsigned)tiles_e), dim3(EARLY_THREADS), 0, st,
                                                             (const uint32_t *)w.early_bits, words, w.early_cnt));
        PROF(KC_RR_SCAN, tiles_e, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.early_cnt, (uint32_t *)nullptr, tiles_e, w.early_cnt + tiles_e));
        uint32_t *holes = (uint32_t *)w.keysC;
        PROF(KC_MISC, early->covered, st, hipLaunchKernelGGL((k_early_gather), dim3((unsigned)tiles_e), dim3(EARLY_THREADS), 0, st,
                                                             (const uint32_t *)w.early_bits, words, (const uint32_t *)w.early_cnt,
                                                             (const uint32_t *)(dSA + 1 - early->off), holes));
        uint32_t total = 0;
        { const int rcw = read_words(&total, w.early_cnt + tiles_e, 4, st); if (rcw) return rcw; }
        early->holes = total;
        early->d_bits = w.early_bits; early->d_tile_off = w.early_cnt; early->d_holes = holes;
        return SA_AMD_OK;
    }

    // Bucket sort of the 32-bit first stage: which key bits do the global passes order?  16 (two passes of 8 bits) when an average
    // bucket fits the 10-pairs-per-thread shapes (n = 2^28: 4096 pairs; 2^29: 8192), 18 (two passes of NINE bits) for larger texts
    // (2^30: 2^18 buckets of 4096), 0 = four global passes.
    int choose_bucket_bits() const
    {
        if (tn.no_bucket_sort || n < tn.bucket_min_n || n <= 1) return 0;
        const bool wide_ok = onesweep_on(w.ss, tn) && tn.onesweep32_shape == 0;      // (nine-bit digits: single-pass engine, default tile)
        if (tn.bucket_bits == 18 && wide_ok) return 18;
        if (tn.bucket_bits == 16) return 16;
        // (measured, random bytes and DNA: with the two key bits that travel in the values (initial_sort_top32) 18 bits win from about
        // 4.9e8 suffixes on -- 2^29: 9.09 against 9.36 ms --, without them only where 16 bits no longer fit)
        const bool value_bits = wide_ok && !tn.no_value_bits && !tn.no_text_keys && g_bits <= 30 &&
                                ((sigma == 256 && P.bits == 8) || (sigma == 4 && P.bits == 2 && !tn.no_packed_text));
        if (value_bits && (n >> 16) > 7500 && (n >> 18) * 10 <= bucket_cap_max() * 9) return 18;
        if ((n >> 16) * 10 <= bucket_cap(2) * 9) return 16;
        if (wide_ok && (n >> 18) * 10 <= bucket_cap_max() * 9) return 18;
        if ((n >> 16) * 10 <= bucket_cap_max() * 9) return 16;
        return 0;
    }

    bool flat_rule_applies() const { return n >= 2 && n < tn.top32_probe_min_n && n < ((int64_t)1 << 24) && !tn.no_flat_rule; }

    // 1-2. which byte values occur -> symbol codes and key geometry; entropy probe, repeat probe, gram keys (read-backs: the 256 presence flags, two duplicate counts, the number of grams in use)
    int geometry_and_probes()
    {
        int rc = SA_AMD_OK; (void)rc;
        // 1. sigma = 256 histogram -> symbol codes, bits per symbol, symbols per key
        HIP_TRY(hipMemsetAsync(w.hist, 0, 256 * 4, st));
        {
            int64_t blocks = ceil_div(ceil_div(n, 16), BH_THREADS);
            if (blocks > 2048) blocks = 2048;
            if (blocks < 1) blocks = 1;
            // (texts below the entropy probe's size: with counts, for the flat-histogram rule below)
            const bool counts = flat_rule_applies();
            if (counts) PROF(KC_BYTE_HIST, n, st, hipLaunchKernelGGL((k_byte_hist_counts), dim3((unsigned)(blocks > 256 ? 256 : blocks)), dim3(BH_THREADS), 0, st, dT, n, w.hist));
            else PROF(KC_BYTE_HIST, n, st, hipLaunchKernelGGL((k_byte_hist), dim3((unsigned)blocks), dim3(BH_THREADS), 0, st, dT, n, w.hist));
        }
        uint32_t hist[256];
        { const int rcw = read_words(hist, w.hist, sizeof(hist), st); if (rcw) return rcw; }
        // Texts too short for the entropy probe (a sampling pass and a read-back of its own): byte values that are all equally
        // frequent (order-0 entropy = log2 of their number: random bytes, random DNA) say "no structure at order 0", and then
        // log2 n + 20 key bits separate nearly every suffix -- five radix passes instead of eight at 1 MiB (random bytes 0.36 ->
        // 0.25 ms; English-like text, whose histogram is anything but flat, would pay 0.50 -> 0.71 for the same keys and keeps all
        // 64 bits).  A flat text that does repeat (a period, a block copied twice) is ordered by the rounds as before.
        int kb_max = tn.key_bits_max;
        if (flat_rule_applies()) {
            double h0 = 0.0; int used = 0;
            for (int c = 0; c < 256; ++c)
                if (hist[c]) { const double pc = (double)hist[c] / (double)n; h0 -= pc * std::log2(pc); ++used; }
            if (used >= 2 && h0 > std::log2((double)used) - 0.01) {
                int kb = (bit_length((uint64_t)(n - 1)) + 20 + 7) / 8 * 8;
                if (kb < 32) kb = 32;
                if (kb < kb_max) kb_max = kb;
                flat_text = true;
            }
        }
        key_bits = make_key_params(hist, &P, &sigma, kb_max);      // (gram keys, step 2c, may shorten it)
        local.sigma = sigma; local.bits_per_symbol = P.bits; local.symbols_per_key = P.k;
        if (sigma == 1 && !tn.no_unary_shortcut) {
            // one byte value repeated: every suffix is a proper prefix of every longer one, the order is by length.  (The general
            // path gets there too -- 23 doubling rounds over one group of n members, 214 ms at 256 MiB -- and stays tested.)
            int64_t blocks = ceil_div(n, 256 * 16);
            if (blocks > 16384) blocks = 16384;
            PROF(KC_MISC, n, st, hipLaunchKernelGGL((k_fill_descending), dim3((unsigned)blocks), dim3(256), 0, st, SA, n));
            unary_done = true;
            return SA_AMD_OK;
        }

        g_bits = bit_length((uint64_t)(n - 1 > 0 ? n - 1 : 1));
        bucket_top_bits = choose_bucket_bits();
        force_dense = tn.force_dense;
        text_ok = !force_dense && !tn.no_text_rounds;
        local_ok = !tn.no_local_sort;

        // 2. entropy probe: do the top 32 key bits already separate (almost) all suffixes?  Then the initial
        //    sort only needs those 4 digits and a cheap round on the low bits finishes the few ties.
        top_shift = 0;
        if (text_ok && local_ok && key_bits > 32 && !tn.no_top32) {
            bool use = tn.force_top32;
            if (!use && n >= tn.top32_probe_min_n) {
                // 2^20 samples, fewer for texts below 8 Mi suffixes (a power of two: the duplicate count hashes into 4 S slots)
                int64_t S = (int64_t)1 << 20;
                while (S > 1024 && S * 8 > n) S >>= 1;
                PROF(KC_MISC, S, st, hipLaunchKernelGGL((k_sample_keys), dim3((unsigned)ceil_div(S, GK_THREADS)), dim3(GK_THREADS), 0, st, dT, P, n, S,
                                                        key_bits - 32, w.keysA));
                // duplicates counted in a hash table (4 entries per sample, in the other key buffer) instead of sorting the sample
                const uint32_t H = (uint32_t)S * 4u;
                HIP_TRY(hipMemsetAsync(w.keysB, 0xff, (size_t)H * 8, st));
                HIP_TRY(hipMemsetAsync(w.total, 0, 8, st));
                PROF(KC_MISC, S, st, hipLaunchKernelGGL((k_count_sample_dups), dim3((unsigned)(ceil_div(S, 256) < 4096 ? ceil_div(S, 256) : 4096)), dim3(256), 0, st, (const uint64_t *)w.keysA, S,
                                                        (unsigned long long *)w.keysB, H - 1u, w.total));
                // the same samples, counted per bucket of the bucket sort: an estimate of its largest bucket (word 1 of the read-back)
                const int tb = choose_bucket_bits();
                if (tb) {
                    HIP_TRY(hipMemsetAsync(w.keysC, 0, ((size_t)1 << tb) * 4, st));
                    // (a quarter of the samples is enough to tell a bucket with a tenth of the text from a flat one, and a quarter of the atomics)
                    PROF(KC_MISC, S / 4, st, hipLaunchKernelGGL((k_sample_bucket_hist), dim3((unsigned)ceil_div(S / 4, BK_STARTS_THREADS)), dim3(BK_STARTS_THREADS), 0, st,
                                                            (const uint64_t *)w.keysA, S / 4, tb, (uint32_t *)w.keysC));
                    PROF(KC_MISC, S, st, hipLaunchKernelGGL((k_u32_max), dim3((unsigned)(((size_t)1 << tb) / BK_STARTS_THREADS)), dim3(BK_STARTS_THREADS), 0, st,
                                                            (const uint32_t *)w.keysC, 1u << tb, w.total + 1));
                }
                uint32_t dups = 0;
                {
                    uint32_t two[2] = { 0, 0 };
                    const int rcw = read_words(two, w.total, 8, st); if (rcw) return rcw;
                    dups = two[0];
                    // (a flat text of 2^28 bytes: 4 samples per bucket on average, the fullest holds about 14 -> 14 336 estimated, 4400 true)
                    const double est = (double)two[1] * (double)n / (double)(S / 4);
                    if (tb && est > 3.0 * (double)bucket_cap_max()) {
                        bucket_top_bits = 0;
                        if (trace) fprintf(stderr, "suffix_array_amd: bucket sort of the 32-bit first stage: largest bucket estimated at %.0f suffixes -> four global passes\n", est);
                    }
                }
                // c - 1 per value seen c times under-counts pairs only when values repeat often, which is the
                // "do not" case anyway; expected number of other suffixes sharing the top bits with a given one:
                const double p32 = (double)n * 2.0 * (double)dups / ((double)S * (double)S);
                use = p32 < (double)tn.top32_partners_x100 / 100.0;
                double p64 = -1.0;
                if (!use && p32 < 4.0 * (double)tn.top32_collisions_x100 / 100.0) {
                    // Not "almost all separated" -- but WHY do suffixes share their top 32 bits?  Chance collisions (a small or
                    // skewed alphabet: few partners each, told apart by the low key bits in one pass over the sorted keys) or
                    // repeats (they share the low bits too and go to the refinement rounds either way).  The same sample,
                    // whole keys: the partners a suffix has on all key bits.  Measured (tools/top32_threshold.py, 1 GiB): sigma 16
                    // skewed, 2.5 partners on 32 bits and none on 64: 42 ms against 74 ms with full keys; DNA with 20 % in
                    // repeats (0.72 / 0.35): 118 against 137 ms; 40 % in repeats (0.82 / 0.45): 154 against 143 ms -- hence 0.8 x the threshold for those.
                    PROF(KC_MISC, S, st, hipLaunchKernelGGL((k_sample_keys), dim3((unsigned)ceil_div(S, GK_THREADS)), dim3(GK_THREADS), 0, st, dT, P, n, S,
                                                            -1, w.keysA));
                    HIP_TRY(hipMemsetAsync(w.keysB, 0xff, (size_t)H * 8, st));
                    HIP_TRY(hipMemsetAsync(w.total, 0, 4, st));
                    PROF(KC_MISC, S, st, hipLaunchKernelGGL((k_count_sample_dups), dim3((unsigned)(ceil_div(S, 256) < 4096 ? ceil_div(S, 256) : 4096)), dim3(256), 0, st, (const uint64_t *)w.keysA, S,
                                                            (unsigned long long *)w.keysB, H - 1u, w.total));
                    uint32_t dups64 = 0;
                    { const int rcw = read_words(&dups64, w.total, 4, st); if (rcw) return rcw; }
                    const double chance = (double)S * (double)S / 8589934592.0;           // 32-bit hash collisions among S samples
                    const double d64 = (double)dups64 > chance ? (double)dups64 - chance : 0.0;
                    p64 = (double)n * 2.0 * d64 / ((double)S * (double)S);
                    use = p64 < 0.8 * (double)tn.top32_partners_x100 / 100.0 && p32 - p64 < (double)tn.top32_collisions_x100 / 100.0;
                }
                if (trace) fprintf(stderr, "suffix_array_amd: entropy probe: %u duplicates of the top 32 key bits among %lld samples -> %.3f expected partners per suffix "
                                           "(on all key bits: %.3f) -> %s\n", dups, (long long)S, p32, p64, use ? "32-bit first stage" : "full keys");
            }
            if (use) top_shift = key_bits - 32;
        }
        local.top32_first = top_shift ? 1 : 0;
        // 2b. repeat probe (texts the first probe did not send to the 32-bit route): the fraction of suffixes that share
        //     2k symbols with another suffix.  Many (copied passages, a corpus): the text-keyed rounds cannot finish, so rank
        //     doubling starts right after the initial sort (measured on C3: 64 ms against 68 ms); few: text-keyed rounds.
        probe_dense = false;
        if (text_ok && local_ok && !top_shift && !force_dense && !tn.no_repeat_probe && n >= ((int64_t)1 << 24)) {
            const int64_t S = (int64_t)1 << 20;
            PROF(KC_MISC, S, st, hipLaunchKernelGGL((k_sample_repeat_keys), dim3((unsigned)ceil_div(S, GK_THREADS)), dim3(GK_THREADS), 0, st, dT, P, n, S,
                                                    w.keysA));
            const uint32_t H = (uint32_t)S * 4u;
            HIP_TRY(hipMemsetAsync(w.keysB, 0xff, (size_t)H * 8, st));
            HIP_TRY(hipMemsetAsync(w.total, 0, 4, st));
            PROF(KC_MISC, S, st, hipLaunchKernelGGL((k_count_sample_dups), dim3((unsigned)(ceil_div(S, 256) < 4096 ? ceil_div(S, 256) : 4096)), dim3(256), 0, st, (const uint64_t *)w.keysA, S,
                                                    (unsigned long long *)w.keysB, H - 1u, w.total));
            uint32_t dups = 0;
            { const int rcw = read_words(&dups, w.total, 4, st); if (rcw) return rcw; }
            const double chance = (double)S * (double)S / 8589934592.0;           // 32-bit hash collisions among S samples
            const double frac = ((double)dups - chance) * (double)n / ((double)S * (double)S);
            probe_dense = frac > 0.04;
            if (trace) fprintf(stderr, "suffix_array_amd: repeat probe: %u duplicates among %lld samples (repeat index %.3f, threshold 0.04) -> %s\n",
                               dups, (long long)S, frac, probe_dense ? "rank doubling from the start" : "text-keyed rounds");
        }

        // 2c. gram keys (texts that did not take the 32-bit route): choose_gram_keys measures and decides; the probes above keep
        //     using the plain key
        if (!top_shift && !tn.no_gram_keys && n >= tn.gram_min_n && n >= 2 && sigma >= 2 && tn.key_bits_max == 64) {
            const int rcg = choose_gram_keys(dT, n, &P, &key_bits, w, st, tn, trace);
            if (rcg) return rcg;
            local.bits_per_symbol = P.bits; local.symbols_per_key = P.k;
        }

        return SA_AMD_OK;
    }

    // 4b. the 32-bit first stage: the top 32 key bits as (u32, u32) pairs -- two global passes + the in-LDS bucket sort, or four
    //     global passes (read-back: the largest bucket)
    int initial_sort_top32(uint32_t *vals0, uint8_t *packed_out, bool iota)
    {
        int rc = SA_AMD_OK; (void)rc;
        uint32_t *k32a = (uint32_t *)w.keysA, *k32b = (uint32_t *)w.keysB;
        // Two global passes over the top 16 key bits + one pass that orders every bucket (= value of those bits) in LDS, when an
        // average bucket fits a workgroup well (n = 2^28: 4096 pairs, 2^29: 8192); larger texts take two passes of NINE bits
        // (2^30: 2^18 buckets of 4096).  A text whose LARGEST bucket fits no workgroup -- known only once the two passes have
        // run -- builds its keys again and takes the four global passes.
        int top_bits = bucket_top_bits;      // 0: four global passes (choose_bucket_bits, the probe's estimate of the largest bucket)
        for (;;) {
            // the first radix pass's digit histogram comes out of k_build_keys (keys in registers there): one read of every key less
            const FirstCounts fc = sort_first_counts(w.ss, tn, n, true);
            const bool counted = n > 1;
            const int rbits = top_bits == 18 ? 9 : RADIX_BITS;
            // a text of all 256 byte values (code = byte, 8 symbols per key): the top 32 key bits are four text bytes -- the first
            // global pass reads them from the text (a quarter of the bytes) and no key array is built in front of it
            const bool text_route = top_bits != 0 && counted && iota && top_shift == 32 && onesweep_on(w.ss, tn) && tn.onesweep32_shape == 0 && !tn.no_text_keys;
            const bool text_keys = text_route && sigma == 256 && P.bits == 8 && packed_out == nullptr;
            // ... and with four symbols (DNA) the bit-packed text is the stream of keys: it is packed first (n / 4 bytes instead of a
            // key array of 4n), and counted and read by the first pass as it stands
            const bool packed_keys = text_route && sigma == 4 && P.bits == 2 && packed_out != nullptr;
            // Key bits for free: an index below 2^g_bits leaves 32 - g_bits bits of the value word unused, and the bucket sort stages
            // 16 bits per pair of which 32 - top_bits are low key bits.  When both have room (texts of 0.6 - 1 GiB: 18 top bits, 30-bit
            // indices) the first pass puts the two key bits BEHIND the 32 there and the bucket sort orders 34 bits: a quarter of the ties
            // on all 32 bits are left for its round on the low key bits (1 GiB of DNA: 22 % of the suffixes -> 6 %).
            int val_extra = 0;
            if ((text_keys || packed_keys) && !tn.no_value_bits) {
                val_extra = 32 - g_bits;
                if (val_extra > BK_MAX_LBITS - (32 - top_bits)) val_extra = BK_MAX_LBITS - (32 - top_bits);
                if (val_extra > 2) val_extra = 2;
                if (val_extra < 0) val_extra = 0;
            }
            if (counted) HIP_TRY(hipMemsetAsync(fc.zero_ptr, 0, fc.zero_bytes, st));
            if (text_keys || packed_keys) {
                int split = 2048 / fc.G;
                while (split > 1 && fc.chunk_elems / split < 16384) split /= 2;
                const int64_t sub = (ceil_div(fc.chunk_elems, split) + 15) & ~(int64_t)15;
                if (packed_keys) {
                    int64_t pblocks = ceil_div(ceil_div(n, 16), 256);
                    if (pblocks > 16384) pblocks = 16384;
                    PROF(KC_BUILD_KEYS, n, st, hipLaunchKernelGGL((k_pack_text2), dim3((unsigned)pblocks), dim3(256), 0, st, dT, n, P, packed_out));
                    PROF(KC_BUILD_KEYS, n, st, hipLaunchKernelGGL((k_packed2_upsweep32), dim3((unsigned)(fc.G * split)), dim3(SORT_THREADS), 0, st, (const uint8_t *)packed_out, n,
                                                                  fc.counts, 32 - top_bits, (1u << rbits) - 1u, fc.chunk_elems, fc.G, split, sub));
                } else
                PROF(KC_BUILD_KEYS, n, st, hipLaunchKernelGGL((k_text_upsweep32), dim3((unsigned)(fc.G * split)), dim3(SORT_THREADS), 0, st, dT, n, fc.counts,
                                                              32 - top_bits, (1u << rbits) - 1u, fc.chunk_elems, fc.G, split, sub));
            } else
            PROF(KC_BUILD_KEYS, n, st, hipLaunchKernelGGL((k_build_keys<true>), dim3((unsigned)ceil_div(ceil_div(n, KB_TILE), KB_TPW)), dim3(KB_THREADS), 0, st, dT, n, P,
                                                          (uint64_t *)nullptr, vals0, k32a, top_shift, packed_out,
                                                          counted ? fc.counts : (uint32_t *)nullptr, fc.chunk_elems, fc.G, (1u << rbits) - 1u, top_bits ? 32 - top_bits : 0));
            SortResult32 s32;
            if (top_bits) {
                rc = sort_pairs32(k32a, w.valsA, k32b, w.valsB, n, 32 - top_bits, 32, w.ss, nullptr, st, &s32, tn, iota, counted, rbits,
                                  text_keys ? dT : (packed_keys ? (const uint8_t *)packed_out : (const uint8_t *)nullptr), n, packed_keys ? 2 : 8, val_extra);
                if (rc) return rc;
                local.sort_passes += s32.passes; local.sorted_elements += (int64_t)s32.passes * n;
                uint32_t *kout = (s32.keys == k32a) ? k32b : k32a;
                bool done = false, fused = false;
                uint32_t largest = 0;
                // the round on the low key bits of the suffixes tied on all 32 (first_round_from_sorted_keys) in the same launch
                const bool fuse = local_ok && !tn.no_fused_finish && !tn.no_bucket_finish && !timing_only();
                BucketFinish F = BucketFinish();
                KeySrc K = KeySrc();
                if (fuse) {
                    const int64_t rr_tiles = ceil_div(n, RR_TILE);
                    HIP_TRY(hipMemsetAsync(w.surv_bits, 0, ((size_t)n + 31) / 32 * 4, st));
                    HIP_TRY(hipMemsetAsync(w.tcnt, 0, (size_t)rr_tiles * 4, st));
                    HIP_TRY(hipMemsetAsync(w.thead, 0, (size_t)rr_tiles * 4, st));
                    HIP_TRY(hipMemsetAsync(w.total, 0, 16, st));
                    HIP_TRY(hipMemsetAsync(w.chg, 0, (size_t)RR_CHG_COUNTERS * 32 * 4, st));
                    F.T = dT; F.n = n; F.cap = tn.group_cap; F.surv_bits = w.surv_bits; F.surv_head = w.isa; F.tile_cnt = w.tcnt; F.counters = w.total;
                    K.mode = KS_LOWKEY; K.kb = top_shift;
                }
                KeyParams Pf = P;
                Pf.packed = packed_out;             // (written by now: the fused round's text look-ups take the bit-packed text, a quarter of the lines)
                rc = bucket_sort32(s32.keys, s32.vals, kout, SA, n, top_bits, w.bk_start, w.os_err + 2, st, tn, &done, &largest,
                                   fuse ? &F : nullptr, &Pf, &K, &fused, val_extra);
                if (rc) return rc;
                bucket_finished = done && fused;
                if (trace) fprintf(stderr, "suffix_array_amd: 32-bit first stage: %d global passes over the top %d key bits, largest bucket %u -> %s\n", s32.passes, top_bits, largest,
                                   done ? (fused ? "low bits and ties on all 32 ordered bucket by bucket in LDS" : "low bits ordered bucket by bucket in LDS")
                                        : "too large: keys rebuilt, four global passes");
                if (!done) { top_bits = 0; continue; }      // (nothing was launched: the finish buffers are zeroed again by whoever uses them)
                local.sort_passes += 1; local.sorted_elements += n;
                sorted32 = kout;
                sr.vals = SA; sr.passes = s32.passes + 1;
                sr.keys = (kout == k32a) ? w.keysA : w.keysB;
                break;
            }
            rc = sort_pairs32(k32a, w.valsA, k32b, w.valsB, n, 0, 32, w.ss, SA, st, &s32, tn, iota, counted);
            if (rc) return rc;
            local.sort_passes += s32.passes; local.sorted_elements += (int64_t)s32.passes * n;
            sorted32 = s32.keys;
            sr.vals = s32.vals; sr.passes = s32.passes;
            sr.keys = (s32.keys == k32a) ? w.keysA : w.keysB;      // the 8n-byte buffer that now holds the sorted 32-bit keys
            break;
        }
        return SA_AMD_OK;
    }

    // 3-4. packed keys and the initial LSD sort, the last pass writing straight into SA
    int initial_sort()
    {
        int rc = SA_AMD_OK; (void)rc;
        // 3. packed keys, 4. initial sort: all key bits as (u64 key, u32 suffix) pairs, or only the top 32 bits as
        //    (u32, u32) pairs in 12 Ki-element tiles -- two thirds of the bytes per pass and half the passes
        sr.keys = w.keysA; sr.vals = w.valsA; sr.passes = 0;
        sorted32 = nullptr;               // top-32 stage: the sorted 32-bit keys (no 64-bit sorted array exists)
        bucket_finished = false;
        // value of pair i = i: not stored by k_build_keys, the first sort pass takes the index (saves 8 B / suffix)
        const bool iota = n >= 2 && key_bits > 0;
        uint32_t *vals0 = iota ? (uint32_t *)nullptr : w.valsA;
        // alphabets of 2, 4 or 16 symbols: k_build_keys also writes the text as bit-packed codes, which every later random
        // read of the text uses instead (a key becomes a bit field of two words; DNA shrinks to a quarter: cache-resident)
        uint8_t *packed_out = nullptr;
        if ((P.bits == 1 || P.bits == 2 || P.bits == 4) && n >= 64 && !tn.no_packed_text) {
            packed_out = w.packed;
            HIP_TRY(hipMemsetAsync(packed_out + (size_t)(n >> 3) * P.bits, 0, 64, st));     // the padding behind the last whole group
        }
        if (top_shift) {
            rc = initial_sort_top32(vals0, packed_out, iota);
            if (rc) return rc;
        } else {
            const FirstCounts fc = sort_first_counts(w.ss, tn, n, false);
            const bool counted = n > 1 && key_bits > 0;
            const int nb0 = key_bits < RADIX_BITS ? key_bits : RADIX_BITS;
            if (counted) HIP_TRY(hipMemsetAsync(fc.zero_ptr, 0, fc.zero_bytes, st));
            if (P.gram > 0)
                PROF(KC_BUILD_KEYS, n, st, hipLaunchKernelGGL((k_build_keys<false, true>), dim3((unsigned)ceil_div(ceil_div(n, KB_TILE), KB_TPW)), dim3(KB_THREADS), 0, st, dT, n, P,
                                                              w.keysA, vals0, (uint32_t *)nullptr, 0, packed_out,
                                                              counted ? fc.counts : (uint32_t *)nullptr, fc.chunk_elems, fc.G, (1u << nb0) - 1u));
            else
                PROF(KC_BUILD_KEYS, n, st, hipLaunchKernelGGL((k_build_keys<false>), dim3((unsigned)ceil_div(ceil_div(n, KB_TILE), KB_TPW)), dim3(KB_THREADS), 0, st, dT, n, P,
                                                              w.keysA, vals0, (uint32_t *)nullptr, 0, packed_out,
                                                              counted ? fc.counts : (uint32_t *)nullptr, fc.chunk_elems, fc.G, (1u << nb0) - 1u));
            // (a text of ONE byte value -- a zero-filled file -- has the same key everywhere but at its end: its passes are the identity
            // and are looked for; any other text does not pay the read-backs)
            // diagnostic library, SA_AMD_SAMPLE_SORT=1 (a measured dead end, profiles/r04_sample_sort_64.txt): the sample sort -- two
            // distribution passes over quantile digits + every bucket ordered in LDS -- instead of one LSD pass per digit
            bool ss_done = false;
#ifdef SA_AMD_DIAG
            if (iota && counted && tn.sample_sort && n >= tn.sample_sort_min_n && key_bits > 32 && onesweep_on(w.ss, tn) && !timing_only()) {
                const size_t need_big = ((size_t)ceil_div(n, SS_TILE) + SS_WAYS) * SS_IDS2 * 4;
                if (need_big <= (size_t)n * 4 && (size_t)n * 4 >= ((size_t)4 << 20)) {
                    rc = sample_sort64(w.keysA, w.keysB, w.keysC, w.valsA, w.valsB, w.U0, w.U1, w.isa, w.G0, w.bk_start, SA, n, key_bits, w.ss, st, &local, tn,
                                       &ss_done, trace);
                    if (rc) return rc;
                    if (ss_done) { sr.keys = w.keysB; sr.vals = SA; sr.passes = 3; sr.skipped = 0; }
                    else {
                        // (cannot sort this text in workgroup-sized buckets: the keys again, then the LSD engine)
                        HIP_TRY(hipMemsetAsync(fc.zero_ptr, 0, fc.zero_bytes, st));
                        if (P.gram > 0)
                            PROF(KC_BUILD_KEYS, n, st, hipLaunchKernelGGL((k_build_keys<false, true>), dim3((unsigned)ceil_div(ceil_div(n, KB_TILE), KB_TPW)), dim3(KB_THREADS), 0, st, dT, n, P,
                                                                          w.keysA, vals0, (uint32_t *)nullptr, 0, packed_out, fc.counts, fc.chunk_elems, fc.G, (1u << nb0) - 1u));
                        else
                            PROF(KC_BUILD_KEYS, n, st, hipLaunchKernelGGL((k_build_keys<false>), dim3((unsigned)ceil_div(ceil_div(n, KB_TILE), KB_TPW)), dim3(KB_THREADS), 0, st, dT, n, P,
                                                                          w.keysA, vals0, (uint32_t *)nullptr, 0, packed_out, fc.counts, fc.chunk_elems, fc.G, (1u << nb0) - 1u));
                    }
                }
            }
#endif
            if (!ss_done) {
            rc = sort_pairs(w.keysA, w.valsA, w.keysB, w.valsB, n, 0, key_bits, w.ss, SA, st, &sr, tn, iota, sigma == 1, counted);
            if (rc) return rc;
            local.sort_passes += sr.passes; local.sorted_elements += (int64_t)sr.passes * n;
            }
        }
        P.packed = packed_out;
        if (sr.vals != SA) {   // n == 1: no pass ran, the values are still in the input buffer
            PROF(KC_MISC, n, st, hipLaunchKernelGGL((k_copy_u32), dim3(1), dim3(256), 0, st, sr.vals, SA, n));
        }

        return SA_AMD_OK;
    }

    // list buffers; the fast finish of the 32-bit first stage / the opt-in fused first text round (read-backs: tied counts)
    int first_round_from_sorted_keys()
    {
        int rc = SA_AMD_OK; (void)rc;
        // 4. group heads of the initial order; how many suffixes are still tied with a neighbour
        Ucur = w.U0; Unext = w.U1; Gcur = w.G0; Gnext = w.G1;
        Vcur = w.valsA;
        tiles = ceil_div(n, RR_TILE);
        m32 = 0;
        m = 0;
        rkA = w.keysA; rkB = w.keysB;          // key buffers of the refinement rounds
        sorted0 = sr.keys;                      // the initial keys in SA order (kept for the rank look-ups)
        lists_ready = false;                         // (Ucur, Gcur, Vcur) already hold the tied suffixes
        depth = P.k;                               // symbols the current order is sorted by
        // Text-keyed rounds pack their symbols as bit fields of ceil(log2 sigma) bits whatever the alphabet: a secondary key only
        // has to preserve the order inside one round, and the base-sigma form costs a 64-bit multiply per symbol in kernels
        // that are instruction-bound (k_group_sort: 26 ps per suffix however small the text).  English-like sigma = 56: six
        // symbols in 36 bits either way.
        Ptext = P;
        if (P.bits == 0) Ptext.bits = bit_length(P.sigma - 1);
        Ptext.gram = 0;                                    // (gram ranks are the initial keys' business only)
        s_sym = 0; tkb = 0;                            // symbols per round, bits of their packed key
        {
            const int room = 64 - g_bits;                   // bits left below the group head
            s_sym = room / Ptext.bits;
            if (s_sym > 64) s_sym = 64;
            tkb = s_sym * Ptext.bits;
        }
        finished32 = false; fused64 = false;
        if (top_shift && local_ok && !tn.no_fused_finish && !timing_only()) {
            // fast finish of the 32-bit first stage: one pass orders every small group by its low key bits in place
            // (k_finish_sorted); only if some group is too large for it does the general path below run instead
            const int cap = tn.group_cap;
            uint32_t *surv_bits = w.surv_bits, *surv_head = w.isa;  // (the ISA is not in use before the doubling rounds)
            if (!bucket_finished) {      // (else k_bucket_sort has done this round while it had the buckets in LDS, into the same records)
            HIP_TRY(hipMemsetAsync(surv_bits, 0, ((size_t)n + 31) / 32 * 4, st));
            HIP_TRY(hipMemsetAsync(w.tcnt, 0, (size_t)tiles * 4, st));
            HIP_TRY(hipMemsetAsync(w.thead, 0, (size_t)tiles * 4, st));
            HIP_TRY(hipMemsetAsync(w.total, 0, 16, st));
            HIP_TRY(hipMemsetAsync(w.chg, 0, (size_t)RR_CHG_COUNTERS * 32 * 4, st));
            KeySrc K = KeySrc(); K.mode = KS_LOWKEY; K.kb = top_shift;
            PROF(KC_FINISH, n, st, hipLaunchKernelGGL((k_finish_sorted<uint32_t, KS_LOWKEY, false>), dim3((unsigned)ceil_div(n, FT_TILE)), dim3(FT_THREADS),
                                                     0, st, sorted32, SA, dT, P, n, K, cap, surv_bits, surv_head, w.tcnt, w.total, (uint32_t *)nullptr,
                                                     (uint32_t *)nullptr, (uint32_t *)nullptr));
            }
            // (k_bucket_sort counts its survivors itself: the scan of the tiles' counts runs only if there are any)
            if (!bucket_finished) PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
            uint32_t cnt3[3] = { 0, 0, 0 };                       // still tied on 64 bits, members of groups nobody owned, tied on 32 bits
            {
                uint32_t words[64 + RR_CHG_COUNTERS * 32];         // (the tied-slot counts are spread over w.chg, directly behind w.total)
                const int rcw = read_words(words, w.total, sizeof(words), st); if (rcw) return rcw;
                cnt3[0] = words[0]; cnt3[1] = words[1];
                for (int c = 0; c < RR_CHG_COUNTERS; ++c) cnt3[2] += words[64 + c * 32];
                if (bucket_finished) {
                    cnt3[0] = 0;
                    for (int c = 0; c < RR_CHG_COUNTERS; ++c) cnt3[0] += words[64 + c * 32 + 1];
                    if (cnt3[0] != 0 && cnt3[1] == 0)
                        PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
                }
            }
            if (cnt3[1] == 0) {
                finished32 = true;
                m = cnt3[0];
                local.locally_sorted += cnt3[2];
                local.unresolved_after_initial = m;
                if (m > 0) {
                    PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_surv_compact), dim3((unsigned)tiles), dim3(256), 0, st, (const uint32_t *)surv_bits,
                                                                (const uint32_t *)surv_head, (const uint32_t *)SA, n, (const uint32_t *)w.tcnt,
                                                                (const uint32_t *)w.total, Ucur, Gcur, Vcur));
                    rkA = (sr.keys == w.keysA) ? w.keysB : w.keysA;
                    rkB = w.keysC;
                    lists_ready = true;
                }
            }
        }
        // (SA_AMD_SPARSE_DIV moves the text-round / doubling boundary for the tests: then the general route decides, as before)
        // Opt-in (SA_AMD_FUSED64=1): measured on C3 the one-pass round costs 8.2 ms against the 4.3 ms of k_group_sort on the tied
        // list -- with 68 % of the slots tied the work list is six entries per thread -- and the whole build 31.0 instead of 28.8 ms.
        if (!top_shift && text_ok && local_ok && s_sym > 0 && tn.fused64 && !tn.no_fused_finish && !tn.sparse_div_set && !timing_only()) {
            // the first text-keyed round straight from the sorted keys (k_finish_sorted): groups of up to `cap` members are
            // ordered in place by the next s_sym symbols, their still-tied members recorded by slot; the members of larger
            // groups are listed (k_todo_compact) and take the general route (refine_list + re-rank), joining the same record;
            // k_surv_compact then lists everything that is still tied, in slot order, for the second round
            const int cap = tn.group_cap;
            const int64_t ft_tiles = ceil_div(n, FT_TILE);
            uint32_t *surv_head = w.isa;
            HIP_TRY(hipMemsetAsync(w.surv_bits, 0, ((size_t)n + 31) / 32 * 4, st));
            HIP_TRY(hipMemsetAsync(w.todo_bits, 0, ((size_t)n + 31) / 32 * 4, st));
            HIP_TRY(hipMemsetAsync(w.ft_cnt, 0, (size_t)(ft_tiles + 1) * 4, st));
            HIP_TRY(hipMemsetAsync(w.ft_head, 0, (size_t)(ft_tiles + 1) * 4, st));
            HIP_TRY(hipMemsetAsync(w.surv_cnt, 0, (size_t)tiles * 4, st));
            HIP_TRY(hipMemsetAsync(w.thead, 0, (size_t)tiles * 4, st));
            HIP_TRY(hipMemsetAsync(w.total, 0, 32, st));
            HIP_TRY(hipMemsetAsync(w.chg, 0, (size_t)RR_CHG_COUNTERS * 32 * 4, st));
            KeySrc K = KeySrc(); K.mode = KS_TEXT; K.h = depth; K.s = s_sym; K.kb = tkb;
            PROF(KC_FINISH, n, st, hipLaunchKernelGGL((k_finish_sorted<uint64_t, KS_TEXT, true>), dim3((unsigned)ft_tiles), dim3(FT_THREADS), 0, st,
                                                     (const uint64_t *)sorted0, SA, dT, Ptext, n, K, cap, w.surv_bits, surv_head, w.surv_cnt, w.total,
                                                     w.todo_bits, w.ft_cnt, w.ft_head));
            PROF(KC_RR_SCAN, ft_tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.ft_cnt, w.ft_head, ft_tiles, w.total + 3));
            uint32_t cnt4[4] = { 0, 0, 0, 0 };                    // [2] tied after the initial sort, [3] members left to the general route
            {
                uint32_t words[64 + RR_CHG_COUNTERS * 32];
                const int rcw = read_words(words, w.total, sizeof(words), st); if (rcw) return rcw;
                cnt4[0] = words[0]; cnt4[1] = words[1]; cnt4[3] = words[3];
                for (int c = 0; c < RR_CHG_COUNTERS; ++c) cnt4[2] += words[64 + c * 32];
            }
            local.unresolved_after_initial = cnt4[2];
            const int64_t m_todo = cnt4[3];
            local.locally_sorted += (int64_t)cnt4[2] - m_todo;
            rkA = (sr.keys == w.keysA) ? w.keysB : w.keysA;
            rkB = w.keysC;
            if (m_todo > 0) {
                PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_todo_compact<uint64_t>), dim3((unsigned)ft_tiles), dim3(FT_THREADS), 0, st,
                                                            (const uint64_t *)sorted0, (const uint32_t *)SA, n, (const uint32_t *)w.todo_bits,
                                                            (const uint32_t *)w.ft_cnt, (const uint32_t *)w.ft_head, (const uint32_t *)(w.total + 3),
                                                            Ucur, Gcur, Vcur));
                uint32_t *Valt = (Vcur == w.valsA) ? w.valsB : w.valsA;
                Refined rf;
                bool big_local = true;                             // (large groups: the global sort does the work either way)
                rc = refine_list(rkA, rkB, Vcur, Valt, Ucur, Gcur, Unext, Gnext, m_todo, n, dT, Ptext, K, g_bits, &big_local, w, st, &local, &rf, tn);
                if (rc) return rc;
                const int64_t tt = ceil_div(m_todo, RR_TILE);
                PROF(KC_RR_COUNT, m_todo, st, hipLaunchKernelGGL((k_rr_count<false>), dim3((unsigned)tt), dim3(RR_THREADS), 0, st, rf.keys,
                                                            (const uint32_t *)Ucur, m_todo, w.ft_cnt, w.ft_head, 0, (uint32_t *)nullptr, 0));
                PROF(KC_RR_SCAN, tt, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.ft_cnt, w.ft_head, tt, w.total + 4));
                PROF(KC_RR_APPLY, m_todo, st, hipLaunchKernelGGL((k_rr_apply<false, true, 4>), dim3((unsigned)tt), dim3(RR_THREADS), 0, st,
                                                            rf.keys, rf.vals, (const uint32_t *)Ucur, m_todo, (const uint32_t *)w.ft_cnt,
                                                            (const uint32_t *)w.ft_head, SA, surv_head, Unext, Gnext, rf.vnext, (uint32_t)n,
                                                            w.surv_bits, 0, (uint64_t *)nullptr, w.surv_cnt, (const uint32_t *)(w.total + 4), 0, (const uint32_t *)nullptr, 0, (uint32_t *)nullptr));
            }
            PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.surv_cnt, w.thead, tiles, w.total));
            { const int rcw = read_words(&m32, w.total, 4, st); if (rcw) return rcw; }
            m = m32;
            Ucur = w.U0; Gcur = w.G0; Vcur = w.valsA; Unext = w.U1; Gnext = w.G1;
            if (m > 0)
                PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_surv_compact), dim3((unsigned)tiles), dim3(256), 0, st, (const uint32_t *)w.surv_bits,
                                                            (const uint32_t *)surv_head, (const uint32_t *)SA, n, (const uint32_t *)w.surv_cnt,
                                                            (const uint32_t *)w.total, Ucur, Gcur, Vcur));
            fused64 = true;
            lists_ready = true;
            depth += s_sym;
            local.text_rounds++;
            local.rounds++;
        }
        return SA_AMD_OK;
    }

    // 4. group heads of the initial order: how many suffixes are still tied (read-back: that count)
    int group_heads()
    {
        int rc = SA_AMD_OK; (void)rc;
        if (!finished32 && !fused64) {
        if (top_shift)
            PROF(KC_RR_COUNT, n, st, hipLaunchKernelGGL((k_rr_count<true, uint32_t>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, sorted32,
                                                        (const uint32_t *)nullptr, n, w.tcnt, w.thead, 0, (uint32_t *)nullptr, 0));
        else
            // (also every tile's first group start: the dense route's first ranks are tail ranks, k_rr_apply FTAIL)
            PROF(KC_RR_COUNT, n, st, hipLaunchKernelGGL((k_rr_count<true, uint64_t, true>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, (const uint64_t *)sr.keys,
                                                        (const uint32_t *)nullptr, n, w.tcnt, w.thead, 0, w.tnext, 0));
        PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
        { const int rcw = read_words(&m32, w.total, 4, st); if (rcw) return rcw; }
        m = m32;
        local.unresolved_after_initial = m;
        if (timing_only()) m = 0;   // diag library: ablation kernels produce wrong orders; stop here
        }
        return SA_AMD_OK;
    }

    // the suffixes tied on the top 32 key bits are ordered by their low key bits (read-back: tied on all 64 bits)
    int finish_top32_ties()
    {
        int rc = SA_AMD_OK; (void)rc;
        if (!finished32 && top_shift && m > 0) {
            // finish the initial sort: the suffixes tied on the top 32 bits are ordered by their low key bits
            rkA = (sr.keys == w.keysA) ? w.keysB : w.keysA;
            rkB = w.keysC;
            PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_rr_apply<true, false, 1, uint32_t>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                        sorted32, (const uint32_t *)SA, (const uint32_t *)nullptr, n,
                                                        (const uint32_t *)w.tcnt, (const uint32_t *)w.thead, SA, w.isa, Ucur, Gcur, Vcur, 0u,
                                                        w.has_isa, 0, (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)w.total, 0, (const uint32_t *)nullptr, 0, (uint32_t *)nullptr));
            uint32_t *Valt = (Vcur == w.valsA) ? w.valsB : w.valsA;
            Refined rf;
            KeySrc K = KeySrc(); K.mode = KS_LOWKEY; K.kb = top_shift;
            rc = refine_list(rkA, rkB, Vcur, Valt, Ucur, Gcur, Unext, Gnext, m, n, dT, P, K, g_bits, &local_ok, w, st, &local, &rf, tn);
            if (rc) return rc;
            tiles = ceil_div(m, RR_TILE);
            PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_rr_count<false>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, rf.keys,
                                                        (const uint32_t *)Ucur, m, w.tcnt, w.thead, 0, (uint32_t *)nullptr, 0));
            PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
            PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rr_apply<false, true, 3>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                        rf.keys, rf.vals, (const uint32_t *)Ucur, m, (const uint32_t *)w.tcnt,
                                                        (const uint32_t *)w.thead, SA, w.isa, Unext, Gnext, rf.vnext, (uint32_t)n,
                                                        (uint32_t *)nullptr, 0, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                                        (const uint32_t *)w.total, 0, (const uint32_t *)nullptr, 0, (uint32_t *)nullptr));
            { const int rcw = read_words(&m32, w.total, 4, st); if (rcw) return rcw; }
            m = m32;
            uint32_t *t;
            t = Ucur; Ucur = Unext; Unext = t;
            t = Gcur; Gcur = Gnext; Gnext = t;
            Vcur = rf.vnext;
            lists_ready = true;
            local.unresolved_after_initial = m;           // now: tied on the whole 64-bit key, as after a full sort
        }
        return SA_AMD_OK;
    }

    // 5. dense route: ranks + ISA; otherwise compaction and the text-keyed rounds (read-back per round: tied count)
    int rank_setup_and_text_rounds()
    {
        int rc = SA_AMD_OK; (void)rc;
        // 5. refinement of the tied suffixes.  Three regimes:
        //   text rounds  while more than n / SPARSE_DIV suffixes are tied: secondary key = the next symbols of
        //                the text itself (no rank array needed yet), depth grows by s symbols per round;
        //   sparse       few tied suffixes: prefix doubling, ranks looked up without an ISA (sparse_key2);
        //   dense        prefix doubling with a full ISA (repetitive texts, or forced for A/B measurements).
        key2_bits = bit_length((uint64_t)(2 * n));
        const int64_t sparse_div = tn.sparse_div;      // (SA_AMD_SPARSE_DIV moves the boundary for tests / A-B)
        const int64_t sparse_limit = n / sparse_div;
        sparse = false;
        const bool dense_first = m > 0 && !lists_ready && (force_dense || (!text_ok && m > sparse_limit) || (probe_dense && m > sparse_limit));
        if (m > 0 && dense_first) {
            // ranks (ISA scatter) + compaction of the tied suffixes; SA already holds the sorted order
            // the first ranks are TAIL ranks (a group's last slot + 1), the form the dense rounds keep (k_rr_apply): the first doubling
            // round then writes only the ranks that change (SA_AMD_NO_FIRST_TAIL=1: head ranks, everything rewritten in round 1)
            isa_tail_ranks = !tn.no_first_tail;
            if (isa_tail_ranks)
                PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan_next), dim3(1), dim3(SPINE_THREADS), 0, st, w.tnext, tiles, (uint32_t *)nullptr));
            if (binned(n, n, tn)) {
                uint64_t *pk = (sr.keys == w.keysA) ? w.keysB : w.keysA;
                if (isa_tail_ranks)
                    PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_rr_apply<true, false, 2, uint64_t, true>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                                sr.keys, (const uint32_t *)SA, (const uint32_t *)nullptr, n, w.tcnt, w.thead,
                                                                SA, w.isa, Ucur, Gcur, Vcur, (uint32_t)n, (uint32_t *)nullptr, 0, pk, w.U1, (const uint32_t *)w.total, 0, (const uint32_t *)w.tnext, 0, (uint32_t *)nullptr));
                else
                PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_rr_apply<true, false, 2>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                            sr.keys, (const uint32_t *)SA, (const uint32_t *)nullptr, n, w.tcnt, w.thead,
                                                            SA, w.isa, Ucur, Gcur, Vcur, (uint32_t)n, (uint32_t *)nullptr, 0, pk, w.U1, (const uint32_t *)w.total, 0, (const uint32_t *)nullptr, 0, (uint32_t *)nullptr));
                rc = scatter_binned((uint32_t *)pk, w.U1, (uint32_t *)sr.keys, w.G1, n, n, w, st, &local, tn);
                if (rc) return rc;
            } else if (isa_tail_ranks) {
                PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_rr_apply<true, false, 0, uint64_t, true>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                            sr.keys, (const uint32_t *)SA, (const uint32_t *)nullptr, n, w.tcnt, w.thead,
                                                            SA, w.isa, Ucur, Gcur, Vcur, (uint32_t)n, (uint32_t *)nullptr, 0,
                                                            (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)w.total, 0, (const uint32_t *)w.tnext, 0, (uint32_t *)nullptr));
            } else {
                PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_rr_apply<true, false, 0>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                            sr.keys, (const uint32_t *)SA, (const uint32_t *)nullptr, n, w.tcnt, w.thead,
                                                            SA, w.isa, Ucur, Gcur, Vcur, (uint32_t)n, (uint32_t *)nullptr, 0,
                                                            (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)w.total, 0, (const uint32_t *)nullptr, 0, (uint32_t *)nullptr));
            }
        } else if (m > 0) {
            // compaction only; the sorted initial keys stay intact for the rank look-ups
            HIP_TRY(hipMemsetAsync(w.has_isa, 0, ((size_t)n + 31) / 32 * 4, st));
            if (!lists_ready) {
                rkA = (sr.keys == w.keysA) ? w.keysB : w.keysA;
                rkB = w.keysC;
                PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_rr_apply<true, false, 1>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                            (const uint64_t *)sorted0, (const uint32_t *)SA, (const uint32_t *)nullptr, n,
                                                            (const uint32_t *)w.tcnt, (const uint32_t *)w.thead, SA, w.isa, Ucur, Gcur, Vcur, 0u,
                                                            w.has_isa, 0, (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)w.total, 0, (const uint32_t *)nullptr, 0, (uint32_t *)nullptr));
            }
            // ---- text-keyed rounds ----
            bool progressing = true;     // a text round that resolves little (runs, long repeats) is the last one

            bool counters_clear = false;      // (RoundCtl: one read-back per round while the rounds have nothing for the global sort)
            while (text_ok && s_sym > 0 && m > sparse_limit && local.text_rounds < tn.max_text_rounds && progressing) {
                if ((rc = early_maybe_start())) return rc;
                const int64_t m_before = m;
                uint32_t *Valt = (Vcur == w.valsA) ? w.valsB : w.valsA;
                Refined rf;
                KeySrc K = KeySrc(); K.mode = KS_TEXT; K.h = depth; K.s = s_sym; K.kb = tkb;
                RoundCtl ctl;
                ctl.defer = prev_clean && local_ok && !tn.no_defer;
                ctl.counters_clear = counters_clear;
                rc = refine_list(rkA, rkB, Vcur, Valt, Ucur, Gcur, Unext, Gnext, m, n, dT, Ptext, K, g_bits, &local_ok, w, st, &local, &rf, tn, false, nullptr, &ctl);
                if (rc) return rc;
                uint32_t words[RC_WORDS];
                for (int attempt = 0; ; ++attempt) {
                    const uint32_t *gate = ctl.deferred ? (const uint32_t *)(w.total + RC_FLAGGED) : (const uint32_t *)nullptr;
                    tiles = ceil_div(m, RR_TILE);
                    PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_rr_count<false>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, rf.keys,
                                                                (const uint32_t *)Ucur, m, w.tcnt, w.thead, 0, (uint32_t *)nullptr, 0, gate));
                    PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan_round), dim3(2), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total,
                                                                   (uint32_t *)nullptr, (uint32_t *)nullptr, w.total + RC_BIG_LISTED, gate));
                    PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rr_apply<false, true, 3>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                                rf.keys, rf.vals, (const uint32_t *)Ucur, m, (const uint32_t *)w.tcnt,
                                                                (const uint32_t *)w.thead, SA, w.isa, Unext, Gnext, rf.vnext,
                                                                (uint32_t)n, (uint32_t *)nullptr, 0, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                                                (const uint32_t *)w.total, 0, (const uint32_t *)nullptr, 0, (uint32_t *)nullptr, gate));
                    { const int rcw = read_words(words, w.total, sizeof(words), st); if (rcw) return rcw; }
                    if (!ctl.deferred) break;
                    if (words[RC_FLAGGED] == 0) { ctl.m_flagged = 0; local.locally_sorted += m; break; }
                    if (attempt > 0) return SA_AMD_EINTERNAL;
                    uint32_t tot[RC_WORDS];
                    memcpy(tot, words, sizeof(tot));
                    ctl.resume_tot = tot; ctl.defer = false; ctl.deferred = false;
                    rc = refine_list(rkA, rkB, Vcur, Valt, Ucur, Gcur, Unext, Gnext, m, n, dT, Ptext, K, g_bits, &local_ok, w, st, &local, &rf, tn, false, nullptr, &ctl);
                    if (rc) return rc;
                    ctl.resume_tot = nullptr;
                }
                prev_clean = ctl.m_flagged == 0;
                counters_clear = true;
                uint32_t *Vnext = rf.vnext;
                m32 = words[0];
                m = m32;
                uint32_t *t;
                t = Ucur; Ucur = Unext; Unext = t;
                t = Gcur; Gcur = Gnext; Gnext = t;
                Vcur = Vnext;
                depth += s_sym;
                local.text_rounds++;
                local.rounds++;
                progressing = m * 4 <= m_before * 3;
                if (trace) fprintf(stderr, "suffix_array_amd: text round %d depth %lld: tied %lld -> %lld  (%.2f ms)\n", local.text_rounds, (long long)depth, (long long)m_before, (long long)m, lap());
            }
            if (m > sparse_limit) {
                // still many ties (repetitive text): build the ISA of the current order and double densely
                int64_t blocks = ceil_div(n, 256);
                if (blocks > 16384) blocks = 16384;
                if (binned(n, n, tn) && (((uintptr_t)dSA) & 15) == 0) {
                    // inverse permutation without n random 4-byte stores: dSA[0 .. n] itself is the key array (dSA[0] = n, the
                    // sentinel, is skipped by the scatter), the value is the index = rank; one 32-bit radix pass bins the pairs by
                    // the top 8 bits of the suffix position, the scatter then works window by window (10.0 -> ~2.5 ms at 256 MiB)
                    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, dSA, (uint32_t)n);
                    LAUNCH_CHECK(st);
                    const size_t H = ((size_t)n + 1 + 3) & ~(size_t)3;      // keys in the first half of an 8(n + 64)-byte buffer, values in the second
                    rc = scatter_binned(dSA, nullptr, (uint32_t *)rkB, (uint32_t *)rkB + H, n + 1, n, w, st, &local, tn, true,
                                        (uint32_t *)rkA, (uint32_t *)rkA + H);
                    if (rc) return rc;
                } else
                    PROF(KC_SCATTER, n, st, hipLaunchKernelGGL((k_isa_from_sa), dim3((unsigned)blocks), dim3(256), 0, st, (const uint32_t *)SA, w.isa, n));
                blocks = ceil_div(m, 256);
                if (blocks > 16384) blocks = 16384;
                PROF(KC_SCATTER, m, st, hipLaunchKernelGGL((k_isa_tied), dim3((unsigned)blocks), dim3(256), 0, st, (const uint32_t *)Vcur,
                                                           (const uint32_t *)Gcur, w.isa, m, n));
            } else {
                sparse = m > 0;
            }
        }
        local.sparse_mode = sparse ? 1 : 0;
        if (trace) fprintf(stderr, "suffix_array_amd: initial sort + text rounds + rank set-up done, %lld tied (%.2f ms since the last line)\n", (long long)m, lap());

        return SA_AMD_OK;
    }

    // prefix doubling on what is still tied, dense (ISA) or sparse (read-backs per round: tied count, ranks written)
    int doubling_rounds()
    {
        int rc = SA_AMD_OK; (void)rc;
        // prefix doubling on what is still tied; `depth` symbols are sorted, so the first offset is `depth`
        const int64_t depth_text = depth;
        int64_t h = depth;
        bool chase_ok = false;
        int split_rest = 0;                               // rounds the three-way split sits out (refine_list)
        bool parent_tail = isa_tail_ranks;                // the ranks in the ISA are tail ranks (set by the first dense round, or by the rank set-up of the dense route)
        int64_t changed_prev = isa_tail_ranks ? m : 0;    // ranks the last dense round wrote (none yet: expect every rank to change)
        int64_t m_local_off = m;                          // size of the tied list when the local pass was last in use
        int rounds_local_off = 0;
        // read-backs (RoundCtl): a round whose predecessor had nothing for the global sort defers that question and blocks ONCE
        bool all_small = false;                           // ... and listed no group for k_group_sort_big: no group is larger than GS_CAP any more
        bool counters_clear = false;                      // the previous round's k_rr_scan_round zeroed the big-group counters
        while (m > 0) {
            if (local.rounds >= 48) return SA_AMD_EINTERNAL;
            if ((rc = early_maybe_start())) return rc;     // (every slot outside the tied list is final from here on)
            uint32_t *Valt = (Vcur == w.valsA) ? w.valsB : w.valsA;
            // the same refinement machinery as the text rounds, keyed by ranks -- small groups (a long repeat gives millions of
            // pairs) are ordered in LDS, only large groups go through the global sort.  Dense: ranks from the ISA; sparse:
            // looked up without one (sparse_key2)
            KeySrc K = KeySrc();
            K.mode = sparse ? KS_SPARSE : KS_RANK; K.h = h; K.kb = key2_bits; K.isa = w.isa;
            // dense rounds chase (up to `chase` rank look-ups per member inside one launch) once a round has had no group left
            // for the global sort: from then on every surviving group is known to share (iters + 1) * h symbols
            // the local pass was given up because (nearly) every member sat in a group no tile can own: it is tried again when the
            // list has halved, and every third round -- if refine_list then finds the average group small enough for a tile (large
            // lists count their groups anyway; a Fibonacci word's groups shrink while the list does not, a periodic text's never do)
            if (!local_ok && !tn.no_local_sort && m * 2 < m_local_off) local_ok = true;
            const bool retry_local = !local_ok && ++rounds_local_off >= 3;
            K.iters = (!sparse && local_ok && chase_ok) ? (m >= tn.chase_big_min ? tn.chase_big : tn.chase) : 1;
            if (K.iters > 1) K.mode = KS_CHASE;
            K.has_isa = w.has_isa; K.sorted_keys = sorted0; K.sorted_top32 = sorted32; K.sa = SA; K.depth = depth_text; K.top_shift = top_shift;
            Refined rf;
            if (local_ok) { m_local_off = m; rounds_local_off = 0; }
            // binned or direct ISA stores: by the number of ranks this round is expected to write -- all of them when the parents'
            // ranks are not tail ranks yet, otherwise about as many as the round before wrote
            const int64_t expect = (!parent_tail || m < changed_prev) ? m : changed_prev;
            const bool bin = !sparse && binned(n, expect, tn);
            RoundCtl ctl;
            ctl.defer = prev_clean && !bin && local_ok && !tn.no_defer;
            ctl.skip_big = all_small;
            ctl.counters_clear = counters_clear;
            rc = refine_list(rkA, rkB, Vcur, Valt, Ucur, Gcur, Unext, Gnext, m, n, dT, P, K, g_bits, &local_ok, w, st, &local, &rf, tn, retry_local, &split_rest, &ctl);
            if (rc) return rc;
            uint32_t words[64 + RR_CHG_COUNTERS * 32];
            for (int attempt = 0; ; ++attempt) {
                if (retry_local && rf.m_global < m) { m_local_off = m; rounds_local_off = 0; }      // (the local pass ran again)
                const uint64_t *keysS = rf.keys; const uint32_t *valsS = rf.vals; uint32_t *Vnext = rf.vnext;
                const uint32_t *gate = ctl.deferred ? (const uint32_t *)(w.total + RC_FLAGGED) : (const uint32_t *)nullptr;
                tiles = ceil_div(m, RR_TILE);
                // dense rounds: a group's rank is its last slot + 1 and a parent's last subgroup keeps it (k_rr_apply, TAIL); the tiles
                // then also need the first group start BEHIND them (k_rr_scan_next, second block of k_rr_scan_round)
                if (sparse)
                    PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_rr_count<false>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, keysS, Ucur, m, w.tcnt,
                                       w.thead, 0, (uint32_t *)nullptr, 0, gate));
                else
                    PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_rr_count<false, uint64_t, true>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, keysS, Ucur, m,
                                       w.tcnt, w.thead, 0, w.tnext, key2_bits, gate));
                PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan_round), dim3(2), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total,
                                                               sparse ? (uint32_t *)nullptr : w.tnext, sparse ? (uint32_t *)nullptr : w.chg, w.total + RC_BIG_LISTED, gate));
                if (sparse) {
                    PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rr_apply<false, true, 1>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                                keysS, valsS, Ucur, m, w.tcnt, w.thead, SA, w.isa, Unext, Gnext, Vnext,
                                                                (uint32_t)n, w.has_isa, key2_bits, (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)w.total, 0, (const uint32_t *)nullptr, 0, (uint32_t *)nullptr, gate));
                } else if (bin) {
                    // Gcur has been consumed by the gather, the other key buffer by nothing: they take the pairs
                    uint64_t *pk = (keysS == rkA) ? rkB : rkA;
                    PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rr_apply<false, true, 2>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                                keysS, valsS, Ucur, m, w.tcnt, w.thead, SA, w.isa, Unext, Gnext, Vnext,
                                                                (uint32_t)n, (uint32_t *)nullptr, key2_bits, pk, Gcur, (const uint32_t *)w.total, 0,
                                                                (const uint32_t *)w.tnext, parent_tail ? 1 : 0, w.chg));
                    // (only the ranks that change became pairs; their number is in the counters -- read together with the round's
                    // other results: the one read-back of this round)
                    { const int rcw = read_words(words, w.total, sizeof(words), st); if (rcw) return rcw; }
                    int64_t pairs = 0;
                    for (int c = 0; c < RR_CHG_COUNTERS; ++c) pairs += words[64 + c * 32];
                    if (pairs > 0) {
                        rc = scatter_binned((uint32_t *)pk, Gcur, (uint32_t *)keysS, (uint32_t *)valsS, pairs, n, w, st, &local, tn);
                        if (rc) return rc;
                    }
                    break;
                } else {
                    PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rr_apply<false, true, 0>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                                keysS, valsS, Ucur, m, w.tcnt, w.thead, SA, w.isa, Unext, Gnext, Vnext,
                                                                (uint32_t)n, (uint32_t *)nullptr, key2_bits, (uint64_t *)nullptr,
                                                                (uint32_t *)nullptr, (const uint32_t *)w.total, 0,
                                                                (const uint32_t *)w.tnext, parent_tail ? 1 : 0, w.chg, gate));
                }
                // w.total (64 words) and the changed-rank counters behind it (w.chg) in one read-back
                { const int rcw = read_words(words, w.total, sparse ? (size_t)RC_WORDS * 4 : sizeof(words), st); if (rcw) return rcw; }
                if (!ctl.deferred) break;
                if (words[RC_FLAGGED] == 0) {
                    // as expected: the local pass ordered everything (what k_group_sort_big ordered was not chased: one look-up)
                    ctl.m_flagged = 0;
                    ctl.big_listed = ctl.skip_big ? -1 : (int64_t)words[RC_BIG_LISTED_SAVED];
                    rf.m_global = words[RC_BIG_ORDERED_SAVED];
                    local.locally_sorted += m;
                    break;
                }
                // the gated launches did nothing: the flagged members go through the global sort now (refine_list picks up behind its
                // local pass with the counts just read), then the same re-rank kernels run ungated
                if (attempt > 0) return SA_AMD_EINTERNAL;
                uint32_t tot[RC_WORDS];
                memcpy(tot, words, sizeof(tot));
                ctl.resume_tot = tot; ctl.defer = false; ctl.deferred = false;
                rc = refine_list(rkA, rkB, Vcur, Valt, Ucur, Gcur, Unext, Gnext, m, n, dT, P, K, g_bits, &local_ok, w, st, &local, &rf, tn, retry_local, &split_rest, &ctl);
                if (rc) return rc;
                ctl.resume_tot = nullptr;
                ++deferred_misses;
            }
            uint32_t *Vnext = rf.vnext;
            prev_clean = ctl.m_flagged == 0;
            if (prev_clean && (ctl.big_listed == 0 || m <= GS_CAP)) all_small = true;
            counters_clear = true;                        // (k_rr_scan_round ran ungated in the end)
            {
                m32 = words[0];
                if (!sparse) {
                    changed_prev = 0;
                    for (int c = 0; c < RR_CHG_COUNTERS; ++c) changed_prev += words[64 + c * 32];
                    parent_tail = true;
                }
            }
            if (trace) fprintf(stderr, "suffix_array_amd: doubling round %d h %lld (%s, %d look-ups, %lld through the global sort): tied %lld -> %u, %lld ranks written\n", local.rounds + 1, (long long)h, sparse ? "sparse" : "dense", K.iters, (long long)rf.m_global, (long long)m, m32, sparse ? -1ll : (long long)changed_prev), fprintf(stderr, "    (%.2f ms)\n", lap());
            m = m32;
            uint32_t *t;
            t = Ucur; Ucur = Unext; Unext = t;
            t = Gcur; Gcur = Gnext; Gnext = t;
            Vcur = Vnext;
            // every group that is still tied went through K.iters look-ups -- unless some went through the global sort (one look-up)
            h *= (K.iters > 1 && rf.m_global == 0) ? (int64_t)(K.iters + 1) : 2;
            chase_ok = rf.m_global == 0;
            local.rounds++;
        }
        return SA_AMD_OK;
    }

};

static int build_device(const uint8_t *dT, uint32_t *dSA, int32_t n32, void *dWork, int64_t work_bytes, hipStream_t st,
                        sa_amd_stats *stats, EarlyDownload *early = nullptr,
                        void *dWork2 = nullptr, int64_t work2_bytes = 0)      // reduced-memory route: the slabs that do not fit work_bytes live here (device-visible pinned host memory)
{
    const int64_t n = n32;
    DeviceBuild B;
    B.dT = dT; B.dSA = dSA; B.SA = dSA + 1; B.n = n; B.st = st;
    B.early = early;
    B.tn = Tuning::from_env(N_SORT_VARIANTS, N_SORT32_VARIANTS, N_OS_SHAPES64, N_OS_SHAPES32);
    B.trace = env_int("SA_AMD_VERBOSE", 0, 0, 9) >= 3;      // one line per refinement round on stderr
    B.trace_t = now_ms();
    memset(&B.local, 0, sizeof(B.local));
    g_readbacks = 0;
    g_posted_off = B.tn.no_posted_readback;
    const Tuning &tn = B.tn;
    sa_amd_stats &local = B.local;
    if (n == 0) {
        PROF(KC_MISC, 1, st, hipLaunchKernelGGL((k_set_u32), dim3(1), dim3(1), 0, st, dSA, 0u));
        HIP_TRY(hipStreamSynchronize(st));
        if (stats) *stats = local;
        return SA_AMD_OK;
    }
    B.w = dWork2 ? carve(dWork, n, (size_t)work_bytes, dWork2) : carve(dWork, n);
    const Workspace &w = B.w;
    if ((int64_t)w.bytes > work_bytes || (int64_t)w.bytes2 > work2_bytes) return SA_AMD_EINVAL;
    if (n <= tn.small_max) {
        // small texts: the whole construction in one launch of one workgroup, everything in LDS (kernels/small.hpp)
        if (n <= SM_LITE_SINGLE_N)
            PROF(KC_MISC, n, st, hipLaunchKernelGGL((k_small_sa_lite), dim3(1), dim3(SM_LITE_THREADS), 0, st, dT, dSA, (int)n, w.total));
        else
            PROF(KC_MISC, n, st, hipLaunchKernelGGL((k_small_sa), dim3(1), dim3(SM_THREADS), 0, st, dT, dSA, (int)n, w.total));
        uint32_t rounds = 0;
        { const int rcw = read_words(&rounds, w.total, 4, st); if (rcw) return rcw; }
        local.rounds = (int)rounds;
        local.readbacks = g_readbacks;
        g_prof.resolve();
        g_last_stats = local;
        if (stats) *stats = local;
        return SA_AMD_OK;
    }
    HIP_TRY(hipMemsetAsync(w.os_err, 0, 16, st));          // look-back give-ups of the single-pass scatter: checked at the end
    int rc;
    if ((rc = B.geometry_and_probes())) return rc;
    if (!B.unary_done) {
    if ((rc = B.initial_sort())) return rc;
    if ((rc = B.first_round_from_sorted_keys())) return rc;
    if ((rc = B.group_heads())) return rc;
    if ((rc = B.finish_top32_ties())) return rc;
    if ((rc = B.rank_setup_and_text_rounds())) return rc;
    if ((rc = B.doubling_rounds())) return rc;
    }
    if ((rc = B.early_finish())) return rc;
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, dSA, (uint32_t)n);   // reference src/saca.rs:13
    LAUNCH_CHECK(st);
    {
        // (synchronises the stream) a look-back that gave up means a tile scatter wrote nothing useful: never a silent wrong array
        // (word 3: a bucket larger than the shape the host picked for k_bucket_sort)
        uint32_t gave_up[4] = { 0, 0, 0, 0 };
        const int rcw = read_words(gave_up, w.os_err, 16, st); if (rcw) return rcw;
        if (gave_up[0] || gave_up[3]) return SA_AMD_EINTERNAL;
    }
    local.readbacks = g_readbacks;
    if (B.trace) fprintf(stderr, "suffix_array_amd: %d blocking read-backs in this build (%d rounds deferred their mid-round count and had to run the global sort after all)\n", g_readbacks, B.deferred_misses);
    g_prof.resolve();
    g_last_stats = local;
    if (stats) *stats = local;
    return SA_AMD_OK;
}

}  // namespace sa
