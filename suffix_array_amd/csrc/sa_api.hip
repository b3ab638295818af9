// sa_api.hip -- host pipeline and C ABI (include/suffix_array_amd.h) of the MI355X-native
// suffix-array construction engine.  Replaces the body of `saca()` (reference
// src/saca.rs:9-15) and the C engine behind `cdivsufsort::sort_in_place` (src/saca.rs:14).
//
// There is deliberately no CPU fallback in this file: every entry point runs the HIP
// kernels of sa_kernels.hpp or returns an error code.
#include "sa_kernels.hpp"
#include "sa_extras.hpp"
#include "../../include/suffix_array_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <thread>
#include <vector>

namespace sa {

static bool debug_sync()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("SA_AMD_DEBUG_SYNC"); v = (e && *e && *e != '0') ? 1 : 0; }
    return v == 1;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            if (getenv("SA_AMD_VERBOSE"))                                                          \
                fprintf(stderr, "suffix_array_amd: %s -> %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? SA_AMD_ENOMEM : SA_AMD_EHIP;                        \
        }                                                                                          \
    } while (0)

#define LAUNCH_CHECK(st)                                                                           \
    do {                                                                                           \
        HIP_TRY(hipGetLastError());                                                                \
        if (debug_sync()) HIP_TRY(hipStreamSynchronize(st));                                       \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int bit_length(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline) ----
enum KClass { KC_BYTE_HIST = 0, KC_BUILD_KEYS, KC_UPSWEEP, KC_SPINE, KC_DOWNSWEEP, KC_RR_COUNT, KC_RR_SCAN, KC_RR_APPLY,
              KC_GATHER, KC_SCATTER, KC_LOCAL, KC_MISC, KC_UPSWEEP32, KC_DOWNSWEEP32, KC_COUNT };
static const char *const kclass_names[KC_COUNT] = { "k_byte_hist", "k_build_keys", "k_radix_upsweep", "k_spine_rows",
                                                    "k_radix_downsweep", "k_rr_count", "k_rr_scan", "k_rr_apply",
                                                    "k_gather_key2", "k_scatter_pairs", "k_group_sort", "misc",   // (k_gather_key2: the plain gathers; k_group_sort: all fused gather + sort kernels)
                                                    "k_radix_upsweep32", "k_radix_downsweep32" };
struct Profiler {
    bool on = false;
    uint64_t mask = ~0ull;      // kernel classes that get events (each pair costs a few microseconds of host time)
    struct Rec { int cls; hipEvent_t a, b; int64_t units; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double ms[KC_COUNT] = { 0 };
    int64_t launches[KC_COUNT] = { 0 }, units[KC_COUNT] = { 0 };
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void begin(int cls, int64_t u, hipStream_t st)
    {
        open = on && ((mask >> cls) & 1ull);
        if (!open) return;
        Rec r; r.cls = cls; r.units = u; r.a = get(); r.b = get();
        (void)hipEventRecord(r.a, st);
        recs.push_back(r);
    }
    bool open = false;
    void end(hipStream_t st) { if (open && !recs.empty()) (void)hipEventRecord(recs.back().b, st); open = false; }
    void resolve()   // call after the stream has been synchronised
    {
        for (auto &r : recs) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.cls] += t; launches[r.cls]++; units[r.cls] += r.units; }
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
    }
};
static thread_local Profiler g_prof;
static thread_local sa_amd_stats g_last_stats;
#define PROF(cls, units, st, launch_stmt)                                                          \
    do { g_prof.begin(cls, units, st); launch_stmt; g_prof.end(st); LAUNCH_CHECK(st); } while (0)

constexpr int MAX_TEXT_ROUNDS = 4;   // text-keyed rounds before falling back to rank doubling with a full ISA
constexpr int64_t SPARSE_DIV = 64;  // sparse refinement when at most n / 64 suffixes are tied after the initial sort
constexpr int SORT_MAX_WG = 1024;   // spine rows are scanned by one 1024-thread block

// downsweep configurations (threads, items per thread, min waves per SIMD); SA_AMD_SORT_VARIANT
// selects one at run time for A/B measurements, the default is the fastest measured on MI355X
typedef void (*DownsweepFn)(const uint64_t *, const uint32_t *, uint64_t *, uint32_t *, uint32_t *,
                            const uint32_t *, int64_t, int, uint32_t, int64_t, int);
struct SortVariant { int threads, items, wg_per_cu; DownsweepFn fn; const char *name; };
static const SortVariant sort_variants[] = {
    { 1024, 8, 2, k_radix_downsweep_wcl<1024, 8, 16, 1, false, uint64_t, 4>, "carry-completed lines 1024x8 + LDS prefetch of half of the next tile's keys (default)" },
    { 1024, 8, 2, k_radix_downsweep_wcl<1024, 8>, "carry-completed lines 1024x8" },
    { 1024, 8, 2, k_radix_downsweep<1024, 8, 4>, "plain tile scatter 1024x8 (first generation)" },
    { 1024, 8, 2, k_radix_downsweep<1024, 8, 4, 1>, "plain 1024x8 ABLATION sequential stores (wrong results)" },
    { 1024, 8, 2, k_radix_downsweep<1024, 8, 4, 33>, "plain 1024x8 ABLATION no ranking + sequential stores (wrong results)" },
    { 1024, 8, 2, k_radix_downsweep<1024, 8, 4, 16>, "plain 1024x8 ABLATION no stores (wrong results)" },
    { 1024, 8, 2, k_radix_downsweep<1024, 8, 4, 49>, "plain 1024x8 ABLATION no ranking, no stores (wrong results)" },
    { 1024, 8, 2, k_radix_downsweep_wcl<1024, 8, 16, 1, true>, "carry 1024x8 DIAGNOSTIC phase stamps (tools/phase_stamps.py)" },
    { 512, 16, 1, k_radix_downsweep_wcl<512, 16>, "carry-completed lines 512x16" },
    { 1024, 8, 2, k_radix_downsweep_wcl<1024, 8, 8>, "carry 1024x8, granule 8" },
};
constexpr int SORT_DEFAULT_VARIANT = 0;
static const SortVariant &sort_variant()
{
    const char *e = getenv("SA_AMD_SORT_VARIANT");          // read per sort: the tests switch it inside one process
    int v = e ? atoi(e) : SORT_DEFAULT_VARIANT;
    if (v < 0 || v >= (int)(sizeof(sort_variants) / sizeof(sort_variants[0]))) v = SORT_DEFAULT_VARIANT;
    return sort_variants[v];
}

struct SortGrid { int G; int64_t tiles_per_wg; int tile; };
static SortGrid sort_grid(int64_t count)
{
    const SortVariant &sv = sort_variant();
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

// device scratch layout for a text of n bytes
struct Workspace {
    uint64_t *keysA, *keysB, *keysC;
    uint32_t *valsA, *valsB, *isa, *U0, *U1, *G0, *G1;
    uint32_t *spine, *digit_tot, *tcnt, *thead, *hist, *total, *has_isa;
    uint8_t *packed;           // bit-packed text (alphabets of 2, 4 or 16 symbols): n / 2 + 64 bytes
    uint32_t *surv_bits, *surv_cnt, *todo_bits, *ft_cnt, *ft_head;   // first refinement round straight from the sorted keys (k_finish_sorted)
    size_t bytes;
};

static Workspace carve(void *base, int64_t n)
{
    Workspace w;
    const size_t N = (size_t)(n > 0 ? n : 1);
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = align_up(off + b, 256); return (char *)base + o; };
    w.keysA = (uint64_t *)take(N * 8);
    w.keysB = (uint64_t *)take(N * 8);
    w.keysC = (uint64_t *)take(N * 8);
    w.valsA = (uint32_t *)take(N * 4);
    w.valsB = (uint32_t *)take(N * 4);
    w.isa = (uint32_t *)take(N * 4);
    w.U0 = (uint32_t *)take(N * 4);
    w.U1 = (uint32_t *)take(N * 4);
    w.G0 = (uint32_t *)take(N * 4);
    w.G1 = (uint32_t *)take(N * 4);
    w.spine = (uint32_t *)take((size_t)RADIX * SORT_MAX_WG * 4);
    w.digit_tot = (uint32_t *)take(RADIX * 4);
    const size_t rr_tiles = (size_t)ceil_div((int64_t)N, RR_TILE);
    w.tcnt = (uint32_t *)take(rr_tiles * 4);
    w.thead = (uint32_t *)take(rr_tiles * 4);
    w.hist = (uint32_t *)take(256 * 4);
    w.total = (uint32_t *)take(256);
    w.has_isa = (uint32_t *)take((N + 31) / 32 * 4);
    w.packed = (uint8_t *)take(N / 2 + 64);
    w.surv_bits = (uint32_t *)take((N + 31) / 32 * 4);
    w.surv_cnt = (uint32_t *)take(rr_tiles * 4);          // (not tcnt: refine_list uses that one for its own compaction)
    w.todo_bits = (uint32_t *)take((N + 31) / 32 * 4);
    const size_t ft_tiles = (size_t)ceil_div((int64_t)N, FT_TILE) + 1;
    w.ft_cnt = (uint32_t *)take(ft_tiles * 4);
    w.ft_head = (uint32_t *)take(ft_tiles * 4);
    w.bytes = off;
    return w;
}

struct SortResult { uint64_t *keys; uint32_t *vals; int passes; };

// stable LSD sort of `count` pairs on key bits [begin_bit, end_bit); ping-pongs between in/alt.
// spine: RADIX * SORT_MAX_WG words, digit_tot: RADIX words.  final_vals (optional): the LAST pass
// writes its values there instead of into the ping-pong buffer (the initial sort delivers
// straight into SA this way).
static int sort_pairs(uint64_t *keys_in, uint32_t *vals_in, uint64_t *keys_alt, uint32_t *vals_alt, int64_t count,
                      int begin_bit, int end_bit, uint32_t *spine, uint32_t *digit_tot, uint32_t *final_vals,
                      hipStream_t st, SortResult *res, bool iota = false)   // iota: value i = index i, vals_in is scratch only
{
    res->keys = keys_in; res->vals = vals_in; res->passes = 0;
    if (count <= 1 || end_bit <= begin_bit) return SA_AMD_OK;
    const SortGrid g = sort_grid(count);
    const SortVariant &sv = sort_variant();
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
            if (split > 1 && res->passes == 0) HIP_TRY(hipMemsetAsync(spine, 0, (size_t)RADIX * g.G * 4, st));
            PROF(KC_UPSWEEP, count, st, hipLaunchKernelGGL((k_radix_upsweep), dim3(g.G * split), dim3(SORT_THREADS), 0, st, kin, spine,
                                                           count, shift, dmask, chunk, g.G, split, sub));
        }
        PROF(KC_SPINE, (int64_t)RADIX * g.G, st, hipLaunchKernelGGL((k_spine_rows), dim3(RADIX), dim3(SPINE_THREADS), 0, st,
                                                                    spine, digit_tot, g.G));
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
}

// 32-bit keys (two-stage initial sort): same three-kernel pass, 12 Ki-pair tiles by default (the LDS stage holds more 4-byte elements)
struct SortResult32 { uint32_t *keys; uint32_t *vals; int passes; };
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
};
static const Sort32Variant &sort32_variant()
{
    const char *e = getenv("SA_AMD_SORT32_VARIANT");
    int v = e ? atoi(e) : 0;          // 1024 x 12 with the next tile's keys prefetched into LDS: measured best on C4 / C5
    if (v < 0 || v > 4) v = 0;
    return sort32_variants[v];
}

static int sort_pairs32(uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_alt, uint32_t *vals_alt, int64_t count, int begin_bit,
                        int end_bit, uint32_t *spine, uint32_t *digit_tot, uint32_t *final_vals, hipStream_t st, SortResult32 *res,
                        bool iota = false)
{
    res->keys = keys_in; res->vals = vals_in; res->passes = 0;
    if (count <= 1 || end_bit <= begin_bit) return SA_AMD_OK;
    const Sort32Variant &sv = sort32_variant();
    const int64_t SORT32_TILE = (int64_t)SORT32_THREADS * sv.items;
    const int64_t tiles = ceil_div(count, SORT32_TILE);
    int64_t tiles_per_wg = ceil_div(tiles, 512);
    if (tiles_per_wg < 1) tiles_per_wg = 1;
    const int G = (int)ceil_div(tiles, tiles_per_wg);
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
            if (split > 1 && res->passes == 0) HIP_TRY(hipMemsetAsync(spine, 0, (size_t)RADIX * G * 4, st));
            PROF(KC_UPSWEEP32, count, st, hipLaunchKernelGGL((k_radix_upsweep32), dim3(G * split), dim3(SORT_THREADS), 0, st,
                                                           (const uint32_t *)kin, spine, count, shift, dmask, chunk, G, split, sub));
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
}

// symbol codes and key geometry from the sigma = 256 histogram; returns the number of key bits to sort
static int make_key_params(const uint32_t *hist, KeyParams *P, int *sigma_out)
{
    int sigma = 0;
    for (int c = 0; c < 256; ++c) {
        if (hist[c]) P->code[c] = (uint8_t)sigma++;
        else P->code[c] = 0;
    }
    *sigma_out = sigma;
    P->packed = nullptr;
    const uint64_t se = sigma > 2 ? (uint64_t)sigma : 2u;      // effective radix (a unary text still needs one bit)
    P->sigma = se;
    int kb_max = 64;                                           // A/B: fewer key bits = fewer radix passes, more left to the rounds
    if (const char *e = getenv("SA_AMD_KEY_BITS")) { kb_max = atoi(e); if (kb_max < 16) kb_max = 16; if (kb_max > 64) kb_max = 64; }
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
struct PinnedWords {
    uint32_t *p = nullptr;
    ~PinnedWords() { if (p) (void)hipHostFree(p); }
};
static thread_local PinnedWords g_pinned;
static int read_words(void *dst, const void *dsrc, size_t bytes, hipStream_t st)     // bytes <= 1024; synchronises the stream
{
    if (!g_pinned.p && hipHostMalloc((void **)&g_pinned.p, 1024, hipHostMallocDefault) != hipSuccess) {
        g_pinned.p = nullptr;
        (void)hipGetLastError();
        HIP_TRY(hipMemcpyAsync(dst, dsrc, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return SA_AMD_OK;
    }
    HIP_TRY(hipMemcpyAsync(g_pinned.p, dsrc, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(dst, g_pinned.p, bytes);
    return SA_AMD_OK;
}

// binned ISA update pays off once the ISA is far larger than the caches and there is enough to write
static bool binned(int64_t n, int64_t count)
{
    if (getenv("SA_AMD_NO_BINNED_ISA")) return false;
    if (getenv("SA_AMD_BINNED_ISA_ALWAYS")) return count > 0;      // tests: exercise the path at small sizes
    return n >= ((int64_t)1 << 25) && count >= ((int64_t)1 << 22);
}

// (suffix, rank) pairs -> one radix pass on the top 8 bits of the suffix position -> windowed scatter
static int scatter_binned(uint64_t *pk, uint32_t *pv, uint64_t *altk, uint32_t *altv, int64_t count, int64_t n,
                          const Workspace &w, hipStream_t st, sa_amd_stats *local)
{
    const int nb = bit_length((uint64_t)(n - 1));
    const int shift = nb > RADIX_BITS ? nb - RADIX_BITS : 0;
    SortResult pr;
    int rc = sort_pairs(pk, pv, altk, altv, count, shift, shift + RADIX_BITS, w.spine, w.digit_tot, nullptr, st, &pr);
    if (rc) return rc;
    local->sort_passes += pr.passes; local->sorted_elements += (int64_t)pr.passes * count;
    PROF(KC_SCATTER, count, st, hipLaunchKernelGGL((k_scatter_pairs), dim3((unsigned)ceil_div(count, 1024)), dim3(256), 0, st,
                                                   (const uint64_t *)pr.keys, (const uint32_t *)pr.vals, w.isa, count, (uint32_t)n));
    return SA_AMD_OK;
}

struct Refined { const uint64_t *keys; const uint32_t *vals; uint32_t *vnext; };

// One refinement round of the tied list with a secondary key taken from the text (KeySrc): afterwards every
// group is ordered by (group head << kb) | key2.  Small groups: gather fused with the in-LDS group sort
// (k_group_sort); groups no tile owns, or everything when *local_ok is off: plain gather + global radix sort.
// scratchU / scratchG: two free 4n-byte buffers.
static int refine_list(uint64_t *rkA, uint64_t *rkB, uint32_t *Vcur, uint32_t *Valt, const uint32_t *Ucur, const uint32_t *Gcur,
                       uint32_t *scratchU, uint32_t *scratchG, int64_t m, int64_t n, const uint8_t *dT, const KeyParams &P,
                       const KeySrc &K, int g_bits, bool *local_ok, const Workspace &w, hipStream_t st, sa_amd_stats *local,
                       Refined *out)
{
    const int64_t tiles = ceil_div(m, RR_TILE);
    const int kb = K.kb;
    SortResult sr;
    int rc;
    if (*local_ok) {
        uint8_t *flags = (uint8_t *)scratchG;
        const unsigned gs_blocks = (unsigned)ceil_div(m, GS_TILE);
        int cap = GS_CAP;                                // largest group ordered in LDS (C3: 1024 beats 512 by 1%)
        if (const char *e = getenv("SA_AMD_GROUP_CAP")) { cap = atoi(e); if (cap < 2) cap = 2; if (cap > GS_CAP) cap = GS_CAP; }
        if (K.mode == KS_TEXT)
            PROF(KC_LOCAL, m, st, hipLaunchKernelGGL((k_group_sort<KS_TEXT>), dim3(gs_blocks), dim3(GS_THREADS), 0, st, (const uint32_t *)Vcur,
                                                     Gcur, Ucur, dT, P, m, n, K, rkA, Vcur, flags, cap));
        else if (K.mode == KS_LOWKEY)
            PROF(KC_LOCAL, m, st, hipLaunchKernelGGL((k_group_sort<KS_LOWKEY>), dim3(gs_blocks), dim3(GS_THREADS), 0, st, (const uint32_t *)Vcur,
                                                     Gcur, Ucur, dT, P, m, n, K, rkA, Vcur, flags, cap));
        else if (K.mode == KS_RANK)
            PROF(KC_LOCAL, m, st, hipLaunchKernelGGL((k_group_sort<KS_RANK>), dim3(gs_blocks), dim3(GS_THREADS), 0, st, (const uint32_t *)Vcur,
                                                     Gcur, Ucur, dT, P, m, n, K, rkA, Vcur, flags, cap));
        else {
            // sparse look-up: its own kernel, one suffix per thread (a chain of ~60 dependent loads each), then the sort on those keys
            int64_t gblocks = ceil_div(m, GK_THREADS);
            if (gblocks > 8192) gblocks = 8192;
            PROF(KC_GATHER, m, st, hipLaunchKernelGGL((k_gather_textkey<KS_SPARSE>), dim3((unsigned)gblocks), dim3(GK_THREADS), 0, st,
                                                      (const uint32_t *)Vcur, Gcur, dT, P, m, n, K, rkA));
            PROF(KC_LOCAL, m, st, hipLaunchKernelGGL((k_group_sort<KS_PRE>), dim3(gs_blocks), dim3(GS_THREADS), 0, st, (const uint32_t *)Vcur,
                                                     Gcur, Ucur, dT, P, m, n, K, rkA, Vcur, flags, cap));
        }
        if (gs_blocks > 1)
            PROF(KC_LOCAL, 0, st, hipLaunchKernelGGL((k_group_sort_straddle), dim3(gs_blocks - 1), dim3(GX_THREADS), 0, st, rkA, Vcur, Gcur,
                                                     Ucur, m, flags, cap));
        PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_flag_count), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                    (const uint8_t *)flags, m, w.tcnt));
        PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
        uint32_t big32 = 0;
        { const int rcw = read_words(&big32, w.total, 4, st); if (rcw) return rcw; }
        const int64_t m_big = big32;
        const size_t half = (((size_t)n / 2 + 1) & ~(size_t)1);
        if ((size_t)m_big <= half) {
            if (m_big > 0) {
                // groups no tile owns: global sort of (group head, key2), then back to their list positions
                PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_flag_gather), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                            (const uint8_t *)flags, (const uint64_t *)rkA, (const uint32_t *)Vcur, m,
                                                            (const uint32_t *)w.tcnt, rkB, Valt, scratchU));
                rc = sort_pairs(rkB, Valt, rkB + half, Valt + half, m_big, 0, kb + g_bits, w.spine, w.digit_tot, nullptr, st, &sr);
                if (rc) return rc;
                local->sort_passes += sr.passes; local->sorted_elements += (int64_t)sr.passes * m_big;
                PROF(KC_SCATTER, m_big, st, hipLaunchKernelGGL((k_scatter_back), dim3((unsigned)ceil_div(m_big, 256)), dim3(256), 0, st,
                                                               (const uint64_t *)sr.keys, (const uint32_t *)sr.vals,
                                                               (const uint32_t *)scratchU, m_big, rkA, Vcur));
            }
            out->keys = rkA; out->vals = Vcur; out->vnext = Valt;
            local->locally_sorted += m - m_big;
            if (m_big * 2 > m) *local_ok = false;           // mostly large groups: not worth another local pass
            return SA_AMD_OK;
        }
    }
    else {
        int64_t gblocks = ceil_div(m, GK_THREADS);
        if (gblocks > 8192) gblocks = 8192;
        if (K.mode == KS_TEXT)
            PROF(KC_GATHER, m, st, hipLaunchKernelGGL((k_gather_textkey<KS_TEXT>), dim3((unsigned)gblocks), dim3(GK_THREADS), 0, st,
                                                      (const uint32_t *)Vcur, Gcur, dT, P, m, n, K, rkA));
        else if (K.mode == KS_LOWKEY)
            PROF(KC_GATHER, m, st, hipLaunchKernelGGL((k_gather_textkey<KS_LOWKEY>), dim3((unsigned)gblocks), dim3(GK_THREADS), 0, st,
                                                      (const uint32_t *)Vcur, Gcur, dT, P, m, n, K, rkA));
        else if (K.mode == KS_RANK)
            PROF(KC_GATHER, m, st, hipLaunchKernelGGL((k_gather_textkey<KS_RANK>), dim3((unsigned)gblocks), dim3(GK_THREADS), 0, st,
                                                      (const uint32_t *)Vcur, Gcur, dT, P, m, n, K, rkA));
        else
            PROF(KC_GATHER, m, st, hipLaunchKernelGGL((k_gather_textkey<KS_SPARSE>), dim3((unsigned)gblocks), dim3(GK_THREADS), 0, st,
                                                      (const uint32_t *)Vcur, Gcur, dT, P, m, n, K, rkA));
    }
    rc = sort_pairs(rkA, Vcur, rkB, Valt, m, 0, kb + g_bits, w.spine, w.digit_tot, nullptr, st, &sr);
    if (rc) return rc;
    local->sort_passes += sr.passes; local->sorted_elements += (int64_t)sr.passes * m;
    out->keys = sr.keys; out->vals = sr.vals;
    out->vnext = (sr.vals == Vcur) ? Valt : Vcur;
    return SA_AMD_OK;
}

static int build_device(const uint8_t *dT, uint32_t *dSA, int32_t n32, void *dWork, int64_t work_bytes, hipStream_t st,
                        sa_amd_stats *stats)
{
    const int64_t n = n32;
    sa_amd_stats local;
    memset(&local, 0, sizeof(local));
    if (n == 0) {
        PROF(KC_MISC, 1, st, hipLaunchKernelGGL((k_set_u32), dim3(1), dim3(1), 0, st, dSA, 0u));
        HIP_TRY(hipStreamSynchronize(st));
        if (stats) *stats = local;
        return SA_AMD_OK;
    }
    Workspace w = carve(dWork, n);
    if ((int64_t)w.bytes > work_bytes) return SA_AMD_EINVAL;
    uint32_t *SA = dSA + 1;

    // 1. sigma = 256 histogram -> symbol codes, bits per symbol, symbols per key
    HIP_TRY(hipMemsetAsync(w.hist, 0, 256 * 4, st));
    {
        int64_t blocks = ceil_div(ceil_div(n, 16), BH_THREADS);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        PROF(KC_BYTE_HIST, n, st, hipLaunchKernelGGL((k_byte_hist), dim3((unsigned)blocks), dim3(BH_THREADS), 0, st, dT, n, w.hist));
    }
    uint32_t hist[256];
    { const int rcw = read_words(hist, w.hist, sizeof(hist), st); if (rcw) return rcw; }
    KeyParams P;
    int sigma;
    const int key_bits = make_key_params(hist, &P, &sigma);
    local.sigma = sigma; local.bits_per_symbol = P.bits; local.symbols_per_key = P.k;

    const int g_bits = bit_length((uint64_t)(n - 1 > 0 ? n - 1 : 1));
    const bool force_dense = getenv("SA_AMD_FORCE_DENSE") != nullptr;
    const bool text_ok = !force_dense && !getenv("SA_AMD_NO_TEXT_ROUNDS");
    bool local_ok = !getenv("SA_AMD_NO_LOCAL_SORT");

    // 2. entropy probe: do the top 32 key bits already separate (almost) all suffixes?  Then the initial
    //    sort only needs those 4 digits and a cheap round on the low bits finishes the few ties.
    int top_shift = 0;
    if (text_ok && local_ok && key_bits > 32 && !getenv("SA_AMD_NO_TOP32")) {
        bool use = getenv("SA_AMD_FORCE_TOP32") != nullptr;
        if (!use && n >= ((int64_t)1 << 24)) {
            const int64_t S = (int64_t)1 << 20;
            PROF(KC_MISC, S, st, hipLaunchKernelGGL((k_sample_keys), dim3((unsigned)ceil_div(S, GK_THREADS)), dim3(GK_THREADS), 0, st, dT, P, n, S,
                                                    key_bits - 32, w.keysA));
            // duplicates counted in a hash table (4 entries per sample, in the other key buffer) instead of sorting the sample
            const uint32_t H = (uint32_t)S * 4u;
            HIP_TRY(hipMemsetAsync(w.keysB, 0xff, (size_t)H * 8, st));
            HIP_TRY(hipMemsetAsync(w.total, 0, 4, st));
            PROF(KC_MISC, S, st, hipLaunchKernelGGL((k_count_sample_dups), dim3(256), dim3(256), 0, st, (const uint64_t *)w.keysA, S,
                                                    (unsigned long long *)w.keysB, H - 1u, w.total));
            uint32_t dups = 0;
            { const int rcw = read_words(&dups, w.total, 4, st); if (rcw) return rcw; }
            // c - 1 per value seen c times under-counts pairs only when values repeat often, which is the
            // "do not" case anyway; expected number of other suffixes sharing the top bits with a given one:
            const double q = 2.0 * (double)dups / ((double)S * (double)S);
            use = (double)n * q < 0.5;
        }
        if (use) top_shift = key_bits - 32;
    }
    local.top32_first = top_shift ? 1 : 0;

    // 3. packed keys, 4. initial sort: all key bits as (u64 key, u32 suffix) pairs, or only the top 32 bits as
    //    (u32, u32) pairs in 12 Ki-element tiles -- two thirds of the bytes per pass and half the passes
    SortResult sr;
    sr.keys = w.keysA; sr.vals = w.valsA; sr.passes = 0;
    const uint32_t *sorted32 = nullptr;               // top-32 stage: the sorted 32-bit keys (no 64-bit sorted array exists)
    int rc;
    // value of pair i = i: not stored by k_build_keys, the first sort pass takes the index (saves 8 B / suffix)
    const bool iota = n >= 2 && key_bits > 0;
    uint32_t *vals0 = iota ? (uint32_t *)nullptr : w.valsA;
    // alphabets of 2, 4 or 16 symbols: k_build_keys also writes the text as bit-packed codes, which every later random
    // read of the text uses instead (a key becomes a bit field of two words; DNA shrinks to a quarter: cache-resident)
    uint8_t *packed_out = nullptr;
    if ((P.bits == 1 || P.bits == 2 || P.bits == 4) && n >= 64 && !getenv("SA_AMD_NO_PACKED_TEXT")) {
        packed_out = w.packed;
        HIP_TRY(hipMemsetAsync(packed_out + (size_t)(n >> 3) * P.bits, 0, 64, st));     // the padding behind the last whole group
    }
    if (top_shift) {
        uint32_t *k32a = (uint32_t *)w.keysA, *k32b = (uint32_t *)w.keysB;
        PROF(KC_BUILD_KEYS, n, st, hipLaunchKernelGGL((k_build_keys<true>), dim3((unsigned)ceil_div(n, KB_TILE)), dim3(KB_THREADS), 0, st, dT, n, P,
                                                      (uint64_t *)nullptr, vals0, k32a, top_shift, packed_out));
        SortResult32 s32;
        rc = sort_pairs32(k32a, w.valsA, k32b, w.valsB, n, 0, 32, w.spine, w.digit_tot, SA, st, &s32, iota);
        if (rc) return rc;
        local.sort_passes += s32.passes; local.sorted_elements += (int64_t)s32.passes * n;
        sorted32 = s32.keys;
        sr.vals = s32.vals; sr.passes = s32.passes;
        sr.keys = (s32.keys == k32a) ? w.keysA : w.keysB;      // the 8n-byte buffer that now holds the sorted 32-bit keys
    } else {
        PROF(KC_BUILD_KEYS, n, st, hipLaunchKernelGGL((k_build_keys<false>), dim3((unsigned)ceil_div(n, KB_TILE)), dim3(KB_THREADS), 0, st, dT, n, P,
                                                      w.keysA, vals0, (uint32_t *)nullptr, 0, packed_out));
        rc = sort_pairs(w.keysA, w.valsA, w.keysB, w.valsB, n, 0, key_bits, w.spine, w.digit_tot, SA, st, &sr, iota);
        if (rc) return rc;
        local.sort_passes += sr.passes; local.sorted_elements += (int64_t)sr.passes * n;
    }
    P.packed = packed_out;
    if (sr.vals != SA) {   // n == 1: no pass ran, the values are still in the input buffer
        PROF(KC_MISC, n, st, hipLaunchKernelGGL((k_copy_u32), dim3(1), dim3(256), 0, st, sr.vals, SA, n));
    }

    // 4. group heads of the initial order; how many suffixes are still tied with a neighbour
    uint32_t *Ucur = w.U0, *Unext = w.U1, *Gcur = w.G0, *Gnext = w.G1;
    uint32_t *Vcur = w.valsA;
    int64_t tiles = ceil_div(n, RR_TILE);
    uint32_t m32 = 0;
    int64_t m = 0;
    uint64_t *rkA = w.keysA, *rkB = w.keysB;          // key buffers of the refinement rounds
    uint64_t *sorted0 = sr.keys;                      // the initial keys in SA order (kept for the rank look-ups)
    bool lists_ready = false;                         // (Ucur, Gcur, Vcur) already hold the tied suffixes
    int64_t depth = P.k;                               // symbols the current order is sorted by
    // Text-keyed rounds pack their symbols as bit fields of ceil(log2 sigma) bits whatever the alphabet: a secondary key only
    // has to preserve the order inside one round, and the base-sigma form costs a 64-bit multiply per symbol in kernels
    // that are instruction-bound (k_group_sort: 26 ps per suffix however small the text).  English-like sigma = 56: six
    // symbols in 36 bits either way.
    KeyParams Ptext = P;
    if (P.bits == 0) Ptext.bits = bit_length(P.sigma - 1);
    int s_sym = 0, tkb = 0;                            // symbols per round, bits of their packed key
    {
        const int room = 64 - g_bits;                   // bits left below the group head
        s_sym = room / Ptext.bits;
        if (s_sym > 64) s_sym = 64;
        tkb = s_sym * Ptext.bits;
    }
    bool finished32 = false, fused64 = false;
    if (top_shift && local_ok && !getenv("SA_AMD_NO_FUSED_FINISH") && !getenv("SA_AMD_TIMING_ONLY_INITIAL_SORT")) {
        // fast finish of the 32-bit first stage: one pass orders every small group by its low key bits in place
        // (k_finish_sorted); only if some group is too large for it does the general path below run instead
        int cap = GS_CAP;
        if (const char *e = getenv("SA_AMD_GROUP_CAP")) { cap = atoi(e); if (cap < 2) cap = 2; if (cap > GS_CAP) cap = GS_CAP; }
        uint32_t *surv_bits = w.surv_bits, *surv_head = w.isa;  // (the ISA is not in use before the doubling rounds)
        HIP_TRY(hipMemsetAsync(surv_bits, 0, ((size_t)n + 31) / 32 * 4, st));
        HIP_TRY(hipMemsetAsync(w.tcnt, 0, (size_t)tiles * 4, st));
        HIP_TRY(hipMemsetAsync(w.thead, 0, (size_t)tiles * 4, st));
        HIP_TRY(hipMemsetAsync(w.total, 0, 16, st));
        KeySrc K = KeySrc(); K.mode = KS_LOWKEY; K.kb = top_shift;
        PROF(KC_LOCAL, n, st, hipLaunchKernelGGL((k_finish_sorted<uint32_t, KS_LOWKEY, false>), dim3((unsigned)ceil_div(n, FT_TILE)), dim3(FT_THREADS),
                                                 0, st, sorted32, SA, dT, P, n, K, cap, surv_bits, surv_head, w.tcnt, w.total, (uint32_t *)nullptr,
                                                 (uint32_t *)nullptr, (uint32_t *)nullptr));
        PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
        uint32_t cnt3[3] = { 0, 0, 0 };                       // still tied on 64 bits, members of groups nobody owned, tied on 32 bits
        { const int rcw = read_words(cnt3, w.total, 12, st); if (rcw) return rcw; }
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
    if (!top_shift && text_ok && local_ok && s_sym > 0 && getenv("SA_AMD_FUSED64") && !getenv("SA_AMD_NO_FUSED_FINISH") &&
        !getenv("SA_AMD_SPARSE_DIV") &&
        !getenv("SA_AMD_TIMING_ONLY_INITIAL_SORT")) {
        // the first text-keyed round straight from the sorted keys (k_finish_sorted): groups of up to `cap` members are
        // ordered in place by the next s_sym symbols, their still-tied members recorded by slot; the members of larger
        // groups are listed (k_todo_compact) and take the general route (refine_list + re-rank), joining the same record;
        // k_surv_compact then lists everything that is still tied, in slot order, for the second round
        int cap = GS_CAP;
        if (const char *e = getenv("SA_AMD_GROUP_CAP")) { cap = atoi(e); if (cap < 2) cap = 2; if (cap > GS_CAP) cap = GS_CAP; }
        const int64_t ft_tiles = ceil_div(n, FT_TILE);
        uint32_t *surv_head = w.isa;
        HIP_TRY(hipMemsetAsync(w.surv_bits, 0, ((size_t)n + 31) / 32 * 4, st));
        HIP_TRY(hipMemsetAsync(w.todo_bits, 0, ((size_t)n + 31) / 32 * 4, st));
        HIP_TRY(hipMemsetAsync(w.ft_cnt, 0, (size_t)(ft_tiles + 1) * 4, st));
        HIP_TRY(hipMemsetAsync(w.ft_head, 0, (size_t)(ft_tiles + 1) * 4, st));
        HIP_TRY(hipMemsetAsync(w.surv_cnt, 0, (size_t)tiles * 4, st));
        HIP_TRY(hipMemsetAsync(w.thead, 0, (size_t)tiles * 4, st));
        HIP_TRY(hipMemsetAsync(w.total, 0, 32, st));
        KeySrc K = KeySrc(); K.mode = KS_TEXT; K.h = depth; K.s = s_sym; K.kb = tkb;
        PROF(KC_LOCAL, n, st, hipLaunchKernelGGL((k_finish_sorted<uint64_t, KS_TEXT, true>), dim3((unsigned)ft_tiles), dim3(FT_THREADS), 0, st,
                                                 (const uint64_t *)sorted0, SA, dT, Ptext, n, K, cap, w.surv_bits, surv_head, w.surv_cnt, w.total,
                                                 w.todo_bits, w.ft_cnt, w.ft_head));
        PROF(KC_RR_SCAN, ft_tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.ft_cnt, w.ft_head, ft_tiles, w.total + 3));
        uint32_t cnt4[4] = { 0, 0, 0, 0 };                    // [2] tied after the initial sort, [3] members left to the general route
        { const int rcw = read_words(cnt4, w.total, 16, st); if (rcw) return rcw; }
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
            rc = refine_list(rkA, rkB, Vcur, Valt, Ucur, Gcur, Unext, Gnext, m_todo, n, dT, Ptext, K, g_bits, &big_local, w, st, &local, &rf);
            if (rc) return rc;
            const int64_t tt = ceil_div(m_todo, RR_TILE);
            PROF(KC_RR_COUNT, m_todo, st, hipLaunchKernelGGL((k_rr_count<false>), dim3((unsigned)tt), dim3(RR_THREADS), 0, st, rf.keys,
                                                        (const uint32_t *)Ucur, m_todo, w.ft_cnt, w.ft_head, 0));
            PROF(KC_RR_SCAN, tt, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.ft_cnt, w.ft_head, tt, w.total + 4));
            PROF(KC_RR_APPLY, m_todo, st, hipLaunchKernelGGL((k_rr_apply<false, true, 4>), dim3((unsigned)tt), dim3(RR_THREADS), 0, st,
                                                        rf.keys, rf.vals, (const uint32_t *)Ucur, m_todo, (const uint32_t *)w.ft_cnt,
                                                        (const uint32_t *)w.ft_head, SA, surv_head, Unext, Gnext, rf.vnext, (uint32_t)n,
                                                        w.surv_bits, 0, (uint64_t *)nullptr, w.surv_cnt, (const uint32_t *)(w.total + 4), 0));
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
    if (!finished32 && !fused64) {
    if (top_shift)
        PROF(KC_RR_COUNT, n, st, hipLaunchKernelGGL((k_rr_count<true, uint32_t>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, sorted32,
                                                    (const uint32_t *)nullptr, n, w.tcnt, w.thead, 0));
    else
        PROF(KC_RR_COUNT, n, st, hipLaunchKernelGGL((k_rr_count<true>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, (const uint64_t *)sr.keys,
                                                    (const uint32_t *)nullptr, n, w.tcnt, w.thead, 0));
    PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
    { const int rcw = read_words(&m32, w.total, 4, st); if (rcw) return rcw; }
    m = m32;
    local.unresolved_after_initial = m;
    if (getenv("SA_AMD_TIMING_ONLY_INITIAL_SORT")) m = 0;   // ablation builds produce wrong orders; stop here
    }
    if (!finished32 && top_shift && m > 0) {
        // finish the initial sort: the suffixes tied on the top 32 bits are ordered by their low key bits
        rkA = (sr.keys == w.keysA) ? w.keysB : w.keysA;
        rkB = w.keysC;
        PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_rr_apply<true, false, 1, uint32_t>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                    sorted32, (const uint32_t *)SA, (const uint32_t *)nullptr, n,
                                                    (const uint32_t *)w.tcnt, (const uint32_t *)w.thead, SA, w.isa, Ucur, Gcur, Vcur, 0u,
                                                    w.has_isa, 0, (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)w.total, 0));
        uint32_t *Valt = (Vcur == w.valsA) ? w.valsB : w.valsA;
        Refined rf;
        KeySrc K = KeySrc(); K.mode = KS_LOWKEY; K.kb = top_shift;
        rc = refine_list(rkA, rkB, Vcur, Valt, Ucur, Gcur, Unext, Gnext, m, n, dT, P, K, g_bits, &local_ok, w, st, &local, &rf);
        if (rc) return rc;
        tiles = ceil_div(m, RR_TILE);
        PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_rr_count<false>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, rf.keys,
                                                    (const uint32_t *)Ucur, m, w.tcnt, w.thead, 0));
        PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
        PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rr_apply<false, true, 3>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                    rf.keys, rf.vals, (const uint32_t *)Ucur, m, (const uint32_t *)w.tcnt,
                                                    (const uint32_t *)w.thead, SA, w.isa, Unext, Gnext, rf.vnext, (uint32_t)n,
                                                    (uint32_t *)nullptr, 0, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                                    (const uint32_t *)w.total, 0));
        { const int rcw = read_words(&m32, w.total, 4, st); if (rcw) return rcw; }
        m = m32;
        uint32_t *t;
        t = Ucur; Ucur = Unext; Unext = t;
        t = Gcur; Gcur = Gnext; Gnext = t;
        Vcur = rf.vnext;
        lists_ready = true;
        local.unresolved_after_initial = m;           // now: tied on the whole 64-bit key, as after a full sort
    }
    // 5. refinement of the tied suffixes.  Three regimes:
    //   text rounds  while more than n / SPARSE_DIV suffixes are tied: secondary key = the next symbols of
    //                the text itself (no rank array needed yet), depth grows by s symbols per round;
    //   sparse       few tied suffixes: prefix doubling, ranks looked up without an ISA (sparse_key2);
    //   dense        prefix doubling with a full ISA (repetitive texts, or forced for A/B measurements).
    const int key2_bits = bit_length((uint64_t)(2 * n));
    int64_t sparse_div = SPARSE_DIV;
    if (const char *e = getenv("SA_AMD_SPARSE_DIV")) { sparse_div = atoll(e); if (sparse_div < 1) sparse_div = 1; }   // tests / A-B
    const int64_t sparse_limit = n / sparse_div;
    bool sparse = false;
    const bool dense_first = m > 0 && !lists_ready && (force_dense || (!text_ok && m > sparse_limit));
    if (m > 0 && dense_first) {
        // ranks (ISA scatter) + compaction of the tied suffixes; SA already holds the sorted order
        if (binned(n, n)) {
            uint64_t *pk = (sr.keys == w.keysA) ? w.keysB : w.keysA;
            PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_rr_apply<true, false, 2>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                        sr.keys, (const uint32_t *)SA, (const uint32_t *)nullptr, n, w.tcnt, w.thead,
                                                        SA, w.isa, Ucur, Gcur, Vcur, (uint32_t)n, (uint32_t *)nullptr, 0, pk, w.U1, (const uint32_t *)w.total, 0));
            rc = scatter_binned(pk, w.U1, sr.keys, w.G1, n, n, w, st, &local);
            if (rc) return rc;
        } else {
            PROF(KC_RR_APPLY, n, st, hipLaunchKernelGGL((k_rr_apply<true, false, 0>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                        sr.keys, (const uint32_t *)SA, (const uint32_t *)nullptr, n, w.tcnt, w.thead,
                                                        SA, w.isa, Ucur, Gcur, Vcur, (uint32_t)n, (uint32_t *)nullptr, 0,
                                                        (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)w.total, 0));
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
                                                        w.has_isa, 0, (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)w.total, 0));
        }
        // ---- text-keyed rounds ----
        bool progressing = true;     // a text round that resolves little (runs, long repeats) is the last one

        while (text_ok && s_sym > 0 && m > sparse_limit && local.text_rounds < MAX_TEXT_ROUNDS && progressing) {
            const int64_t m_before = m;
            uint32_t *Valt = (Vcur == w.valsA) ? w.valsB : w.valsA;
            Refined rf;
            KeySrc K = KeySrc(); K.mode = KS_TEXT; K.h = depth; K.s = s_sym; K.kb = tkb;
            rc = refine_list(rkA, rkB, Vcur, Valt, Ucur, Gcur, Unext, Gnext, m, n, dT, Ptext, K, g_bits, &local_ok, w, st, &local, &rf);
            if (rc) return rc;
            const uint64_t *keysS = rf.keys;                  // (group, text key) pairs ordered inside every group
            const uint32_t *valsS = rf.vals;
            uint32_t *Vnext = rf.vnext;
            tiles = ceil_div(m, RR_TILE);
            PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_rr_count<false>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, keysS,
                                                        (const uint32_t *)Ucur, m, w.tcnt, w.thead, 0));
            PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
            PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rr_apply<false, true, 3>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                        keysS, valsS, (const uint32_t *)Ucur, m, (const uint32_t *)w.tcnt,
                                                        (const uint32_t *)w.thead, SA, w.isa, Unext, Gnext, Vnext,
                                                        (uint32_t)n, (uint32_t *)nullptr, 0, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                                        (const uint32_t *)w.total, 0));
            { const int rcw = read_words(&m32, w.total, 4, st); if (rcw) return rcw; }
            m = m32;
            uint32_t *t;
            t = Ucur; Ucur = Unext; Unext = t;
            t = Gcur; Gcur = Gnext; Gnext = t;
            Vcur = Vnext;
            depth += s_sym;
            local.text_rounds++;
            local.rounds++;
            progressing = m * 4 <= m_before * 3;
        }
        if (m > sparse_limit) {
            // still many ties (repetitive text): build the ISA of the current order and double densely
            int64_t blocks = ceil_div(n, 256);
            if (blocks > 16384) blocks = 16384;
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

    // prefix doubling on what is still tied; `depth` symbols are sorted, so the first offset is `depth`
    const int64_t depth_text = depth;
    int64_t h = depth;
    while (m > 0) {
        if (local.rounds >= 48) return SA_AMD_EINTERNAL;
        uint32_t *Valt = (Vcur == w.valsA) ? w.valsB : w.valsA;
        // the same refinement machinery as the text rounds, keyed by ranks -- small groups (a long repeat gives millions of
        // pairs) are ordered in LDS, only large groups go through the global sort.  Dense: ranks from the ISA; sparse:
        // looked up without one (sparse_key2)
        KeySrc K = KeySrc();
        K.mode = sparse ? KS_SPARSE : KS_RANK; K.h = h; K.kb = key2_bits; K.isa = w.isa;
        K.has_isa = w.has_isa; K.sorted_keys = sorted0; K.sorted_top32 = sorted32; K.sa = SA; K.depth = depth_text; K.top_shift = top_shift;
        Refined rf;
        rc = refine_list(rkA, rkB, Vcur, Valt, Ucur, Gcur, Unext, Gnext, m, n, dT, P, K, g_bits, &local_ok, w, st, &local, &rf);
        if (rc) return rc;
        const uint64_t *keysS = rf.keys; const uint32_t *valsS = rf.vals; uint32_t *Vnext = rf.vnext;
        tiles = ceil_div(m, RR_TILE);
        PROF(KC_RR_COUNT, m, st, hipLaunchKernelGGL((k_rr_count<false>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st, keysS, Ucur, m, w.tcnt,
                           w.thead, 0));
        PROF(KC_RR_SCAN, tiles, st, hipLaunchKernelGGL((k_rr_scan), dim3(1), dim3(SPINE_THREADS), 0, st, w.tcnt, w.thead, tiles, w.total));
        if (sparse) {
            PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rr_apply<false, true, 1>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                        keysS, valsS, Ucur, m, w.tcnt, w.thead, SA, w.isa, Unext, Gnext, Vnext,
                                                        (uint32_t)n, w.has_isa, key2_bits, (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)w.total, 0));
        } else if (binned(n, m)) {
            // Gcur has been consumed by the gather, the other key buffer by nothing: they take the pairs
            uint64_t *pk = (keysS == rkA) ? rkB : rkA;
            PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rr_apply<false, true, 2>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                        keysS, valsS, Ucur, m, w.tcnt, w.thead, SA, w.isa, Unext, Gnext, Vnext,
                                                        (uint32_t)n, (uint32_t *)nullptr, key2_bits, pk, Gcur, (const uint32_t *)w.total, 0));
            rc = scatter_binned(pk, Gcur, (uint64_t *)keysS, (uint32_t *)valsS, m, n, w, st, &local);
            if (rc) return rc;
        } else {
            PROF(KC_RR_APPLY, m, st, hipLaunchKernelGGL((k_rr_apply<false, true, 0>), dim3((unsigned)tiles), dim3(RR_THREADS), 0, st,
                                                        keysS, valsS, Ucur, m, w.tcnt, w.thead, SA, w.isa, Unext, Gnext, Vnext,
                                                        (uint32_t)n, (uint32_t *)nullptr, key2_bits, (uint64_t *)nullptr,
                                                        (uint32_t *)nullptr, (const uint32_t *)w.total, 0));
        }
        { const int rcw = read_words(&m32, w.total, 4, st); if (rcw) return rcw; }
        m = m32;
        uint32_t *t;
        t = Ucur; Ucur = Unext; Unext = t;
        t = Gcur; Gcur = Gnext; Gnext = t;
        Vcur = Vnext;
        h *= 2;
        local.rounds++;
    }
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, dSA, (uint32_t)n);   // reference src/saca.rs:13
    LAUNCH_CHECK(st);
    HIP_TRY(hipStreamSynchronize(st));
    g_prof.resolve();
    g_last_stats = local;
    if (stats) *stats = local;
    return SA_AMD_OK;
}

static int pick_device()
{
    const char *e = getenv("SA_AMD_DEVICE");
    return e ? atoi(e) : -1;   // -1: keep the calling thread's current device
}

// Per-thread device buffers of the host-pointer entry points.  Small and medium texts (the
// reference's own tests build thousands of arrays of < 4096 bytes, reference src/tests.rs:14) would
// otherwise pay three hipMalloc/hipFree pairs and a stream per call.  One grow-only block per
// (thread, device), kept while it is at most SA_AMD_CACHE_MAX_BYTES (default 1 GiB); larger
// requests are allocated and freed per call.  Thread-local, so the entry points stay re-entrant.
struct HostCache {
    int device = -1;
    void *block = nullptr;
    size_t bytes = 0;
    hipStream_t stream = nullptr;
    ~HostCache() { release(); }
    void release()
    {
        if (block) (void)hipFree(block);
        if (stream) (void)hipStreamDestroy(stream);
        block = nullptr; bytes = 0; stream = nullptr; device = -1;
    }
};
static thread_local HostCache g_cache;

static size_t cache_limit()
{
    const char *e = getenv("SA_AMD_CACHE_MAX_BYTES");
    return e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)1 << 30);
}

// host buffers in, host buffers out; with_sentinel writes SA[0] = n too (saca layout)
static int build_host(const uint8_t *T, uint32_t *SA_host, int32_t n, bool with_sentinel, int device)
{
    if (n < 0 || (n > 0 && (!T || !SA_host)) || (with_sentinel && !SA_host)) return SA_AMD_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SA_AMD_ENODEVICE;
    if (device >= ndev) return SA_AMD_EINVAL;
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    if (n == 0) { if (with_sentinel) SA_host[0] = 0; return SA_AMD_OK; }
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    const size_t wb = (size_t)sa_amd_workspace_bytes(n);
    const size_t tb = align_up((size_t)n, 256), sb = align_up(((size_t)n + 1) * 4, 256);
    const size_t need = tb + sb + wb;
    auto hip_rc = [&](hipError_t e) { return e == hipSuccess ? SA_AMD_OK : (e == hipErrorOutOfMemory ? SA_AMD_ENOMEM : SA_AMD_EHIP); };
    int rc = SA_AMD_OK;
    void *block = nullptr;
    hipStream_t st = nullptr;
    const bool cached = need <= cache_limit();
    if (cached) {
        if (g_cache.device != cur) g_cache.release();
        if (!g_cache.stream) { if ((rc = hip_rc(hipStreamCreate(&g_cache.stream)))) return rc; g_cache.device = cur; }
        if (g_cache.bytes < need) {
            if (g_cache.block) { (void)hipFree(g_cache.block); g_cache.block = nullptr; g_cache.bytes = 0; }
            size_t want = need + need / 4;                      // some slack so a growing series does not reallocate every call
            if (want > cache_limit()) want = need;
            if ((rc = hip_rc(hipMalloc(&g_cache.block, want)))) return rc;
            g_cache.bytes = want;
        }
        block = g_cache.block;
        st = g_cache.stream;
    } else {
        if ((rc = hip_rc(hipStreamCreate(&st)))) return rc;
        if ((rc = hip_rc(hipMalloc(&block, need)))) { (void)hipStreamDestroy(st); return rc; }
    }
    uint8_t *dT = (uint8_t *)block;
    uint32_t *dSA = (uint32_t *)((char *)block + tb);
    void *dW = (char *)block + tb + sb;
    rc = hip_rc(hipMemcpyAsync(dT, T, (size_t)n, hipMemcpyHostToDevice, st));
    if (rc == SA_AMD_OK) rc = build_device(dT, dSA, n, dW, (int64_t)wb, st, nullptr);
    if (rc == SA_AMD_OK) {
        if (with_sentinel) rc = hip_rc(hipMemcpyAsync(SA_host, dSA, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, st));
        else rc = hip_rc(hipMemcpyAsync(SA_host, dSA + 1, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    }
    const int rs = hip_rc(hipStreamSynchronize(st));       // also drains the stream after a failure
    if (rc == SA_AMD_OK) rc = rs;
    if (!cached) { (void)hipFree(block); (void)hipStreamDestroy(st); }
    return rc;
}

}  // namespace sa

extern "C" {

#define SA_EXPORT __attribute__((visibility("default")))

SA_EXPORT int32_t sa_amd_max_length(void) { return SA_AMD_MAX_LENGTH; }

SA_EXPORT int32_t sa_amd_divsufsort(const uint8_t *T, int32_t *SA, int32_t n)
{
    return sa::build_host(T, (uint32_t *)SA, n, false, sa::pick_device());
}

SA_EXPORT int32_t sa_amd_saca_u8(const uint8_t *T, uint32_t *SA, int32_t n)
{
    return sa::build_host(T, SA, n, true, sa::pick_device());
}

SA_EXPORT int32_t sa_amd_saca_batch(const uint8_t *const *T, uint32_t *const *SA, const int32_t *n, const int32_t *device,
                                    int32_t count, int32_t *status)
{
    if (count < 0 || (count > 0 && (!T || !SA || !n))) return SA_AMD_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SA_AMD_ENODEVICE;
    std::vector<int32_t> st((size_t)count, SA_AMD_OK);
    std::vector<std::vector<int>> per_dev((size_t)ndev);
    for (int i = 0; i < count; ++i) {
        const int d = device ? device[i] : i % ndev;
        if (d < 0 || d >= ndev) { st[(size_t)i] = SA_AMD_EINVAL; continue; }
        per_dev[(size_t)d].push_back(i);
    }
    // Two host threads per device (SA_AMD_BATCH_THREADS, 1..4), each with its own stream and device block, take the
    // device's items in turn: while one waits for its 4(n+1)-byte copy back over PCIe the other uploads and computes,
    // so the link and the GPU overlap instead of alternating.
    int per = 2;
    if (const char *e = getenv("SA_AMD_BATCH_THREADS")) { per = atoi(e); if (per < 1) per = 1; if (per > 4) per = 4; }
    std::vector<std::atomic<size_t>> next((size_t)ndev);
    for (auto &a : next) a.store(0);
    std::vector<std::thread> workers;
    for (int d = 0; d < ndev; ++d) {
        const size_t items = per_dev[(size_t)d].size();
        for (int k = 0; k < per && (size_t)k < items; ++k) {
            workers.emplace_back([&, d]() {
                for (;;) {
                    const size_t q = next[(size_t)d].fetch_add(1);
                    if (q >= per_dev[(size_t)d].size()) break;
                    const int i = per_dev[(size_t)d][q];
                    st[(size_t)i] = sa::build_host(T[i], SA[i], n[i], true, d);
                }
            });
        }
    }
    for (auto &t : workers) t.join();
    int32_t first = SA_AMD_OK;
    for (int i = 0; i < count; ++i) {
        if (status) status[i] = st[(size_t)i];
        if (first == SA_AMD_OK && st[(size_t)i] != SA_AMD_OK) first = st[(size_t)i];
    }
    return first;
}

SA_EXPORT int64_t sa_amd_workspace_bytes(int32_t n)
{
    if (n < 0) return -1;
    return (int64_t)sa::carve(nullptr, n).bytes;
}

SA_EXPORT int32_t sa_amd_saca_device(const uint8_t *dT, uint32_t *dSA, int32_t n, void *dWork, int64_t work_bytes,
                                     void *stream, sa_amd_stats *stats)
{
    if (n < 0 || !dSA || (n > 0 && (!dT || !dWork))) return SA_AMD_EINVAL;
    return sa::build_device(dT, dSA, n, dWork, work_bytes, (hipStream_t)stream, stats);
}

// ---- next rows: bucket table and integrity check on the device-resident arrays ----

SA_EXPORT int32_t sa_amd_bucket_table_device(const uint8_t *dT, const uint32_t *dSA, int32_t n, uint32_t *dBkt, void *stream)
{
    if (n < 0 || !dSA || !dBkt || (n > 0 && !dT)) return SA_AMD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sa::k_bucket_table, dim3((sa::BKT_LEN + 255) / 256), dim3(256), 0, st, dT, dSA, (int64_t)n, dBkt);
    if (hipGetLastError() != hipSuccess) return SA_AMD_EHIP;
    return hipStreamSynchronize(st) == hipSuccess ? SA_AMD_OK : SA_AMD_EHIP;
}

SA_EXPORT int32_t sa_amd_check_integrity_device(const uint8_t *dT, int32_t n, const uint32_t *dSA, void *dWork,
                                                int64_t work_bytes, void *stream)
{
    if (n < 0 || !dSA || !dWork || (n > 0 && !dT)) return SA_AMD_EINVAL;
    if (work_bytes < ((int64_t)n + 1) * 4 + 256) return SA_AMD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    uint32_t *flags = (uint32_t *)dWork;
    uint32_t *rank = (uint32_t *)((char *)dWork + 256);
    if (hipMemsetAsync(flags, 0, 4, st) != hipSuccess) return SA_AMD_EHIP;
    int64_t blocks = ((int64_t)n + 1 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(sa::k_ci_scatter, dim3((unsigned)blocks), dim3(256), 0, st, dSA, (int64_t)n, rank, flags);
    hipLaunchKernelGGL(sa::k_ci_check, dim3((unsigned)blocks), dim3(256), 0, st, dT, dSA, (int64_t)n, (const uint32_t *)rank, flags);
    if (hipGetLastError() != hipSuccess) return SA_AMD_EHIP;
    uint32_t f = 0;
    if (hipMemcpyAsync(&f, flags, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return SA_AMD_EHIP;
    if (hipStreamSynchronize(st) != hipSuccess) return SA_AMD_EHIP;
    if (f & 1u) return SA_AMD_ERANGE;
    return (f & 2u) ? 0 : 1;
}

// host buffers; which = 1: bucket table, 2: integrity check, 3: build SA (into SA, n + 1 entries) then bucket table
static int32_t extras_host(const uint8_t *T, int32_t n, uint32_t *SA, int64_t sa_len, uint32_t *bkt, int which)
{
    using namespace sa;
    if (n < 0 || !SA || (n > 0 && !T)) return SA_AMD_EINVAL;
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    if (which == 2 && sa_len != (int64_t)n + 1) return 0;          // reference src/sa.rs:73-75: false
    uint8_t *dT = nullptr; uint32_t *dSA = nullptr, *dB = nullptr; void *dW = nullptr;
    int32_t rc = SA_AMD_OK;
    const size_t N = (size_t)n;
    auto cleanup = [&]() { if (dT) (void)hipFree(dT); if (dSA) (void)hipFree(dSA); if (dB) (void)hipFree(dB); if (dW) (void)hipFree(dW); };
    auto hrc = [&](hipError_t e) { return e == hipSuccess ? SA_AMD_OK : (e == hipErrorOutOfMemory ? SA_AMD_ENOMEM : SA_AMD_EHIP); };
    if ((rc = hrc(hipMalloc((void **)&dT, N ? N : 1)))) { cleanup(); return rc; }
    if ((rc = hrc(hipMalloc((void **)&dSA, (N + 1) * 4)))) { cleanup(); return rc; }
    if (N && (rc = hrc(hipMemcpy(dT, T, N, hipMemcpyHostToDevice)))) { cleanup(); return rc; }
    if (which == 3) {
        const int64_t wb = sa_amd_workspace_bytes(n);
        if ((rc = hrc(hipMalloc(&dW, (size_t)wb)))) { cleanup(); return rc; }
        rc = build_device(dT, dSA, n, dW, wb, nullptr, nullptr);
        if (rc == SA_AMD_OK) rc = hrc(hipMemcpy(SA, dSA, (N + 1) * 4, hipMemcpyDeviceToHost));
        if (rc) { cleanup(); return rc; }
    } else if ((rc = hrc(hipMemcpy(dSA, SA, (N + 1) * 4, hipMemcpyHostToDevice)))) { cleanup(); return rc; }
    if (which == 1 || which == 3) {
        if ((rc = hrc(hipMalloc((void **)&dB, (size_t)BKT_LEN * 4)))) { cleanup(); return rc; }
        rc = sa_amd_bucket_table_device(dT, dSA, n, dB, nullptr);
        if (rc == SA_AMD_OK) rc = hrc(hipMemcpy(bkt, dB, (size_t)BKT_LEN * 4, hipMemcpyDeviceToHost));
    } else {
        const int64_t wb = ((int64_t)n + 1) * 4 + 256;
        if ((rc = hrc(hipMalloc(&dW, (size_t)wb)))) { cleanup(); return rc; }
        rc = sa_amd_check_integrity_device(dT, n, dSA, dW, wb, nullptr);
    }
    cleanup();
    return rc;
}

SA_EXPORT int32_t sa_amd_bucket_table(const uint8_t *T, int32_t n, const uint32_t *SA, uint32_t *bkt)
{
    if (!bkt) return SA_AMD_EINVAL;
    return extras_host(T, n, (uint32_t *)SA, (int64_t)n + 1, bkt, 1);
}

SA_EXPORT int32_t sa_amd_saca_u8_buckets(const uint8_t *T, uint32_t *SA, int32_t n, uint32_t *bkt)
{
    if (!bkt) return SA_AMD_EINVAL;
    return extras_host(T, n, SA, (int64_t)n + 1, bkt, 3);
}

SA_EXPORT int32_t sa_amd_check_integrity(const uint8_t *T, int32_t n, const uint32_t *SA, int64_t sa_len)
{
    return extras_host(T, n, (uint32_t *)SA, sa_len, nullptr, 2);
}

// ---- device-resident index: text + suffix array kept in HBM for bucket table, integrity check and batched search ----

struct sa_amd_index {
    int device;
    int32_t n;
    uint8_t *dT;
    uint32_t *dSA;
};

SA_EXPORT int32_t sa_amd_index_create(const uint8_t *T, int32_t n, const uint32_t *SA, sa_amd_index **out)
{
    using namespace sa;
    if (!out || n < 0 || (n > 0 && !T)) return SA_AMD_EINVAL;
    *out = nullptr;
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    auto hrc = [&](hipError_t e) { return e == hipSuccess ? SA_AMD_OK : (e == hipErrorOutOfMemory ? SA_AMD_ENOMEM : SA_AMD_EHIP); };
    sa_amd_index *ix = new (std::nothrow) sa_amd_index();
    if (!ix) return SA_AMD_ENOMEM;
    ix->n = n; ix->dT = nullptr; ix->dSA = nullptr; ix->device = 0;
    (void)hipGetDevice(&ix->device);
    const size_t N = (size_t)n;
    int32_t rc = hrc(hipMalloc((void **)&ix->dT, N ? N : 1));
    if (rc == SA_AMD_OK) rc = hrc(hipMalloc((void **)&ix->dSA, (N + 1) * 4));
    if (rc == SA_AMD_OK && N) rc = hrc(hipMemcpy(ix->dT, T, N, hipMemcpyHostToDevice));
    if (rc == SA_AMD_OK) {
        if (SA) rc = hrc(hipMemcpy(ix->dSA, SA, (N + 1) * 4, hipMemcpyHostToDevice));
        else {                                                   // SuffixArray::new on the device
            void *dW = nullptr;
            const int64_t wb = sa_amd_workspace_bytes(n);
            rc = hrc(hipMalloc(&dW, (size_t)wb));
            if (rc == SA_AMD_OK) rc = build_device(ix->dT, ix->dSA, n, dW, wb, nullptr, nullptr);
            if (dW) (void)hipFree(dW);
        }
    }
    if (rc != SA_AMD_OK) { if (ix->dT) (void)hipFree(ix->dT); if (ix->dSA) (void)hipFree(ix->dSA); delete ix; return rc; }
    *out = ix;
    return SA_AMD_OK;
}

SA_EXPORT void sa_amd_index_destroy(sa_amd_index *ix)
{
    if (!ix) return;
    if (ix->dT) (void)hipFree(ix->dT);
    if (ix->dSA) (void)hipFree(ix->dSA);
    delete ix;
}

SA_EXPORT int32_t sa_amd_index_sa(const sa_amd_index *ix, uint32_t *SA_out)
{
    if (!ix || !SA_out) return SA_AMD_EINVAL;
    return hipMemcpy(SA_out, ix->dSA, ((size_t)ix->n + 1) * 4, hipMemcpyDeviceToHost) == hipSuccess ? SA_AMD_OK : SA_AMD_EHIP;
}

SA_EXPORT int32_t sa_amd_index_buckets(const sa_amd_index *ix, uint32_t *bkt)
{
    if (!ix || !bkt) return SA_AMD_EINVAL;
    uint32_t *dB = nullptr;
    if (hipMalloc((void **)&dB, (size_t)sa::BKT_LEN * 4) != hipSuccess) return SA_AMD_ENOMEM;
    int32_t rc = sa_amd_bucket_table_device(ix->dT, ix->dSA, ix->n, dB, nullptr);
    if (rc == SA_AMD_OK && hipMemcpy(bkt, dB, (size_t)sa::BKT_LEN * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = SA_AMD_EHIP;
    (void)hipFree(dB);
    return rc;
}

SA_EXPORT int32_t sa_amd_index_check_integrity(const sa_amd_index *ix)
{
    if (!ix) return SA_AMD_EINVAL;
    void *dW = nullptr;
    const int64_t wb = ((int64_t)ix->n + 1) * 4 + 256;
    if (hipMalloc(&dW, (size_t)wb) != hipSuccess) return SA_AMD_ENOMEM;
    const int32_t rc = sa_amd_check_integrity_device(ix->dT, ix->n, ix->dSA, dW, wb, nullptr);
    (void)hipFree(dW);
    return rc;
}

SA_EXPORT int32_t sa_amd_index_search(const sa_amd_index *ix, const uint8_t *pat_data, const int64_t *pat_off, int32_t count,
                                      uint8_t *contains, uint32_t *range_lo, uint32_t *range_hi, uint32_t *lcp_start,
                                      uint32_t *lcp_len)
{
    using namespace sa;
    if (!ix || count < 0 || (count > 0 && !pat_off)) return SA_AMD_EINVAL;
    if (count == 0) return SA_AMD_OK;
    const int64_t total = pat_off[count];
    if (total < 0 || (total > 0 && !pat_data)) return SA_AMD_EINVAL;
    for (int32_t i = 0; i < count; ++i) if (pat_off[i + 1] < pat_off[i]) return SA_AMD_EINVAL;
    auto hrc = [&](hipError_t e) { return e == hipSuccess ? SA_AMD_OK : (e == hipErrorOutOfMemory ? SA_AMD_ENOMEM : SA_AMD_EHIP); };
    const size_t C = (size_t)count;
    uint8_t *dP = nullptr; int64_t *dO = nullptr; uint8_t *dC = nullptr; uint32_t *dR = nullptr;
    int32_t rc = hrc(hipMalloc((void **)&dP, total ? (size_t)total : 1));
    if (rc == SA_AMD_OK) rc = hrc(hipMalloc((void **)&dO, (C + 1) * 8));
    if (rc == SA_AMD_OK) rc = hrc(hipMalloc((void **)&dC, C));
    if (rc == SA_AMD_OK) rc = hrc(hipMalloc((void **)&dR, C * 4 * 4));
    if (rc == SA_AMD_OK && total) rc = hrc(hipMemcpy(dP, pat_data, (size_t)total, hipMemcpyHostToDevice));
    if (rc == SA_AMD_OK) rc = hrc(hipMemcpy(dO, pat_off, (C + 1) * 8, hipMemcpyHostToDevice));
    if (rc == SA_AMD_OK) {
        const int64_t threads = (int64_t)count * WAVE;
        hipLaunchKernelGGL(k_search_batch, dim3((unsigned)ceil_div(threads, SEARCH_THREADS)), dim3(SEARCH_THREADS), 0, nullptr,
                           (const uint8_t *)ix->dT, (const uint32_t *)ix->dSA, (int64_t)ix->n, (const uint8_t *)dP,
                           (const int64_t *)dO, count, dC, dR, dR + C, dR + 2 * C, dR + 3 * C);
        rc = hrc(hipGetLastError());
        if (rc == SA_AMD_OK) rc = hrc(hipDeviceSynchronize());
    }
    if (rc == SA_AMD_OK && contains) rc = hrc(hipMemcpy(contains, dC, C, hipMemcpyDeviceToHost));
    if (rc == SA_AMD_OK && range_lo) rc = hrc(hipMemcpy(range_lo, dR, C * 4, hipMemcpyDeviceToHost));
    if (rc == SA_AMD_OK && range_hi) rc = hrc(hipMemcpy(range_hi, dR + C, C * 4, hipMemcpyDeviceToHost));
    if (rc == SA_AMD_OK && lcp_start) rc = hrc(hipMemcpy(lcp_start, dR + 2 * C, C * 4, hipMemcpyDeviceToHost));
    if (rc == SA_AMD_OK && lcp_len) rc = hrc(hipMemcpy(lcp_len, dR + 3 * C, C * 4, hipMemcpyDeviceToHost));
    if (dP) (void)hipFree(dP);
    if (dO) (void)hipFree(dO);
    if (dC) (void)hipFree(dC);
    if (dR) (void)hipFree(dR);
    return rc;
}

// ---- packed format (reference src/packed_sa.rs); byte layout: u32 magic "SA4x" LE, u32 length, u64 data length
//      (bincode's Vec<u8> prefix), data ----

static int sa_bits_of(uint32_t length)          // reference src/packed_sa.rs:127-129
{
    const uint32_t v = length ? length - 1 : 0;
    return v ? sa::bit_length(v) : 0;
}

SA_EXPORT int64_t sa_amd_pack_bound(int64_t length)
{
    if (length < 0 || length > 0xffffffffLL) return -1;
    const int bits = sa_bits_of((uint32_t)length);
    return 16 + (int64_t)((length + 127) / 128) * bits * 16;
}

SA_EXPORT int32_t sa_amd_pack(const uint32_t *SA, int64_t length, uint8_t *out, int64_t capacity, int64_t *out_len)
{
    using namespace sa;
    if (!SA || !out || !out_len || length < 1 || length > 0xffffffffLL) return SA_AMD_EINVAL;
    if (capacity < sa_amd_pack_bound(length)) return SA_AMD_EINVAL;
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    const int bits = sa_bits_of((uint32_t)length);
    const int64_t blocks = (length + 127) / 128;
    const int64_t words = blocks * bits * 4;
    int64_t data_len = 0;
    if (bits > 0) {
        uint32_t *dS = nullptr, *dO = nullptr;
        HIP_TRY(hipMalloc((void **)&dS, (size_t)length * 4));
        HIP_TRY(hipMalloc((void **)&dO, (size_t)words * 4));
        HIP_TRY(hipMemcpy(dS, SA, (size_t)length * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_pack4x, dim3((unsigned)ceil_div(words, 256)), dim3(256), 0, nullptr, (const uint32_t *)dS, length, bits, dO, words);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out + 16, dO, (size_t)words * 4, hipMemcpyDeviceToHost));
        (void)hipFree(dS); (void)hipFree(dO);
        data_len = words * 4;
        if (length % 128) {                                   // a partial last block loses its trailing zero bytes (src/packed_sa.rs:41-45)
            const int64_t last = (blocks - 1) * bits * 16;
            while (data_len > last && out[16 + data_len - 1] == 0) --data_len;
        }
    }
    const uint32_t magic = 2016690515u, len32 = (uint32_t)length;   // src/packed_sa.rs:7
    const uint64_t dl = (uint64_t)data_len;
    memcpy(out, &magic, 4); memcpy(out + 4, &len32, 4); memcpy(out + 8, &dl, 8);
    *out_len = 16 + data_len;
    return SA_AMD_OK;
}

SA_EXPORT int32_t sa_amd_unpack(const uint8_t *bytes, int64_t nbytes, uint32_t *SA, int64_t capacity, int64_t *length)
{
    using namespace sa;
    if (!bytes || !length || nbytes < 16) return SA_AMD_EINVAL;
    uint32_t magic, len32; uint64_t dl;
    memcpy(&magic, bytes, 4); memcpy(&len32, bytes + 4, 4); memcpy(&dl, bytes + 8, 8);
    if (magic != 2016690515u || dl != (uint64_t)(nbytes - 16)) return SA_AMD_EINVAL;       // InvalidData in the reference
    *length = len32;
    if (!SA || capacity < (int64_t)len32) return SA_AMD_EINVAL;
    const int bits = sa_bits_of(len32);
    const int64_t blocks = ((int64_t)len32 + 127) / 128;
    if ((int64_t)dl > blocks * bits * 16) return SA_AMD_EINVAL;
    if (len32 == 0) return SA_AMD_OK;
    if (bits == 0) { SA[0] = 0; return SA_AMD_OK; }           // length 1: the reference's unpack loop does not terminate here
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    const int64_t in_words = ((int64_t)dl + 3) / 4;
    uint32_t *dI = nullptr, *dS = nullptr;
    HIP_TRY(hipMalloc((void **)&dI, (size_t)(in_words ? in_words : 1) * 4));
    HIP_TRY(hipMalloc((void **)&dS, (size_t)len32 * 4));
    HIP_TRY(hipMemset(dI, 0, (size_t)(in_words ? in_words : 1) * 4));
    if (dl) HIP_TRY(hipMemcpy(dI, bytes + 16, (size_t)dl, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_unpack4x, dim3((unsigned)ceil_div((int64_t)len32, 256)), dim3(256), 0, nullptr, (const uint32_t *)dI, in_words,
                       (int64_t)len32, bits, dS);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(SA, dS, (size_t)len32 * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dI); (void)hipFree(dS);
    return SA_AMD_OK;
}

SA_EXPORT int32_t sa_amd_debug_phase_cycles(uint64_t *out, int32_t count)
{
    unsigned long long h[16] = { 0 };
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(sa::g_phase_cycles), sizeof(h)) != hipSuccess) return SA_AMD_EHIP;
    for (int i = 0; i < count && i < 16; ++i) out[i] = h[i];
    unsigned long long z[16] = { 0 };
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sa::g_phase_cycles), z, sizeof(z));
    return 16;
}

SA_EXPORT int32_t sa_amd_debug_group_sort_stamps(int32_t on)
{
    const int v = on ? 1 : 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(sa::g_gs_stamp_on), &v, sizeof(v)) == hipSuccess ? SA_AMD_OK : SA_AMD_EHIP;
}

SA_EXPORT void sa_amd_release_cache(void) { sa::g_cache.release(); }

SA_EXPORT void sa_amd_last_stats(sa_amd_stats *out)
{
    if (out) *out = sa::g_last_stats;
}

SA_EXPORT int32_t sa_amd_device_count(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return 0;
    return ndev;
}

SA_EXPORT const char *sa_amd_strerror(int32_t code)
{
    switch (code) {
    case SA_AMD_OK: return "ok";
    case SA_AMD_EINVAL: return "invalid argument";
    case SA_AMD_ENOMEM: return "out of memory";
    case SA_AMD_EHIP: return "HIP runtime error";
    case SA_AMD_ENODEVICE: return "no HIP device";
    case SA_AMD_EINTERNAL: return "internal error: refinement did not converge";
    case SA_AMD_ERANGE: return "suffix offset out of range";
    default: return "unknown error";
    }
}

SA_EXPORT void sa_amd_profile_begin_classes(uint64_t class_mask)
{
    sa::Profiler &p = sa::g_prof;
    p.on = true;
    p.mask = class_mask;
    for (int i = 0; i < sa::KC_COUNT; ++i) { p.ms[i] = 0; p.launches[i] = 0; p.units[i] = 0; }
}

SA_EXPORT void sa_amd_profile_begin(void) { sa_amd_profile_begin_classes(~0ull); }

SA_EXPORT int32_t sa_amd_profile_end(double *ms, int64_t *launches, int64_t *units, int32_t capacity)
{
    sa::Profiler &p = sa::g_prof;
    p.on = false;
    const int cnt = capacity < sa::KC_COUNT ? capacity : sa::KC_COUNT;
    for (int i = 0; i < cnt; ++i) {
        if (ms) ms[i] = p.ms[i];
        if (launches) launches[i] = p.launches[i];
        if (units) units[i] = p.units[i];
    }
    return sa::KC_COUNT;
}

SA_EXPORT const char *sa_amd_profile_kernel_name(int32_t i)
{
    return (i >= 0 && i < sa::KC_COUNT) ? sa::kclass_names[i] : "";
}

SA_EXPORT const char *sa_amd_version(void) { return "suffix_array_amd 0.1.0 (gfx950)"; }

SA_EXPORT int32_t sa_amd_test_sort_pairs(uint64_t *keys, uint32_t *vals, int64_t count, int32_t begin_bit, int32_t end_bit)
{
    using namespace sa;
    if (count < 0 || (count > 0 && (!keys || !vals)) || begin_bit < 0 || end_bit > 64) return SA_AMD_EINVAL;
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    if (count == 0) return SA_AMD_OK;
    uint64_t *dk = nullptr; uint32_t *dv = nullptr, *spine = nullptr;
    const size_t N = (size_t)count;
    HIP_TRY(hipMalloc((void **)&dk, N * 8 * 2));
    HIP_TRY(hipMalloc((void **)&dv, N * 4 * 2));
    HIP_TRY(hipMalloc((void **)&spine, ((size_t)RADIX * SORT_MAX_WG + RADIX) * 4));
    HIP_TRY(hipMemcpy(dk, keys, N * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dv, vals, N * 4, hipMemcpyHostToDevice));
    SortResult sr;
    int rc = sort_pairs(dk, dv, dk + N, dv + N, count, begin_bit, end_bit, spine, spine + (size_t)RADIX * SORT_MAX_WG, nullptr, nullptr, &sr);
    if (rc == SA_AMD_OK) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(keys, sr.keys, N * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(vals, sr.vals, N * 4, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dk); (void)hipFree(dv); (void)hipFree(spine);
    return rc;
}

SA_EXPORT int32_t sa_amd_test_sort_pairs32(uint32_t *keys, uint32_t *vals, int64_t count, int32_t begin_bit, int32_t end_bit)
{
    using namespace sa;
    if (count < 0 || (count > 0 && (!keys || !vals)) || begin_bit < 0 || end_bit > 32) return SA_AMD_EINVAL;
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    if (count == 0) return SA_AMD_OK;
    uint32_t *dk = nullptr, *dv = nullptr, *spine = nullptr;
    const size_t N = ((size_t)count + 3) & ~(size_t)3;
    HIP_TRY(hipMalloc((void **)&dk, N * 4 * 2));
    HIP_TRY(hipMalloc((void **)&dv, N * 4 * 2));
    HIP_TRY(hipMalloc((void **)&spine, ((size_t)RADIX * SORT_MAX_WG + RADIX) * 4));
    HIP_TRY(hipMemcpy(dk, keys, (size_t)count * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dv, vals, (size_t)count * 4, hipMemcpyHostToDevice));
    SortResult32 sr;
    int rc = sort_pairs32(dk, dv, dk + N, dv + N, count, begin_bit, end_bit, spine, spine + (size_t)RADIX * SORT_MAX_WG, nullptr, nullptr, &sr);
    if (rc == SA_AMD_OK) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(keys, sr.keys, (size_t)count * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(vals, sr.vals, (size_t)count * 4, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dk); (void)hipFree(dv); (void)hipFree(spine);
    return rc;
}

SA_EXPORT int32_t sa_amd_test_build_keys(const uint8_t *T, int32_t n, uint64_t *keys, int32_t *bits, int32_t *k)
{
    using namespace sa;
    if (n <= 0 || !T || !keys) return SA_AMD_EINVAL;
    if (sa_amd_device_count() <= 0) return SA_AMD_ENODEVICE;
    uint8_t *dT = nullptr; uint64_t *dk = nullptr; uint32_t *dv = nullptr, *dh = nullptr;
    HIP_TRY(hipMalloc((void **)&dT, (size_t)n));
    HIP_TRY(hipMalloc((void **)&dk, (size_t)n * 8));
    HIP_TRY(hipMalloc((void **)&dv, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&dh, 1024));
    HIP_TRY(hipMemcpy(dT, T, (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dh, 0, 1024));
    int64_t blocks = ceil_div(ceil_div((int64_t)n, 16), BH_THREADS);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_byte_hist, dim3((unsigned)blocks), dim3(BH_THREADS), 0, nullptr, dT, (int64_t)n, dh);
    uint32_t hist[256];
    HIP_TRY(hipMemcpy(hist, dh, 1024, hipMemcpyDeviceToHost));
    KeyParams P; int sigma;
    make_key_params(hist, &P, &sigma);
    hipLaunchKernelGGL((k_build_keys<false>), dim3((unsigned)ceil_div((int64_t)n, KB_TILE)), dim3(KB_THREADS), 0, nullptr, dT,
                       (int64_t)n, P, dk, dv, (uint32_t *)nullptr, 0, (uint8_t *)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(keys, dk, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (bits) *bits = P.bits;
    if (k) *k = P.k;
    (void)hipFree(dT); (void)hipFree(dk); (void)hipFree(dv); (void)hipFree(dh);
    return SA_AMD_OK;
}

}  // extern "C"
