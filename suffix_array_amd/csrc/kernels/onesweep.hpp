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
//     A tile sums its predecessors' counts backwards until it meets an inclusive prefix; the workgroup reads the granules
//     of the 16 tiles in front of it in one go and the loads stay in flight while the tile's keys and values are staged.
//     A tile only ever waits for tiles with smaller tickets, which are held by running workgroups: no co-residency
//     assumption, no deadlock when another stream shares the device.
// Algorithmic traffic per pass: sizeof(key) + 4 read, the same written, + 16 bytes of granules per 256 / TILE elements.
#pragma once
#include "common.hpp"
#include "radix_sort.hpp"

namespace sa {

constexpr int OS_NSEG = 8;           // segments per pass (one per XCD; fewer when there are fewer tiles)
constexpr int OS_LB = 4;             // granule loads per thread and look-back round
constexpr int OS_LB_ROWS = 16;       // predecessors the whole workgroup reads in one go (OS_LB per thread, four threads per digit)
constexpr int OS_THREADS = 1024;

__device__ __forceinline__ unsigned os_xcc_id()
{
    return (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID, bits [3:0]
}

typedef __attribute__((address_space(1))) unsigned long long os_gu64;

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
    uint32_t flags;                 // bit 0: load the next tile's keys behind this tile's stores, not in front of them (A/B)
};

template <int ITEMS, typename KeyT>
__global__ __launch_bounds__(OS_THREADS) void k_onesweep(const KeyT *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                          KeyT *__restrict__ keys_out, uint32_t *__restrict__ vals_out, OnesweepPass P)
{
    constexpr int THREADS = OS_THREADS;
    constexpr int TILE = THREADS * ITEMS;
    constexpr int NWAVES = THREADS / WAVE;
    constexpr int WAVE_ELEMS = WAVE * ITEMS;
    static_assert(TILE < 65536, "16-bit tile-local counters");
    __shared__ __attribute__((aligned(16))) KeyT lds_k[TILE];
    __shared__ __attribute__((aligned(16))) uint32_t lds_v[TILE];
    __shared__ uint16_t wave_hist[NWAVES][RADIX];
    __shared__ uint32_t digit_base[RADIX];     // first stage slot of digit d
    __shared__ uint32_t goff[RADIX];           // global position = goff + stage slot
    __shared__ uint32_t seg0[RADIX];           // next pass: segment of the digit run's first element ...
    __shared__ uint32_t bnd[RADIX];            // ... and the first global position that belongs to the segment after it
    __shared__ uint32_t segbase[RADIX];        // where the current segment's run of digit d starts
    __shared__ uint32_t hist2[RADIX * OS_NSEG];
    __shared__ unsigned long long lbx[OS_LB_ROWS][RADIX];   // the granules of the last OS_LB_ROWS tiles, as the look-back found them
    __shared__ uint32_t scan_lds[NWAVES + 1];
    __shared__ int s_tile, s_seg;

    const int tid = threadIdx.x, l = lane_id(), w = wave_id();
    const unsigned xcc = os_xcc_id();
    const int nseg = P.nseg;
    const bool count_next = P.hist_next != nullptr;
    for (int i = tid; i < RADIX * OS_NSEG; i += THREADS) hist2[i] = 0;
    int cand = 0;                              // index into my list of candidate segments: my XCD's first, then all
    int cur_seg = -1;
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
    // Tickets are taken ONE TILE AHEAD and not waited for: the atomic for the tile after the next is issued in the middle of
    // a tile and looked at in the middle of the following one.  Every workgroup does so at the same point of its loop, so
    // ticket order is still the order in which tiles start.
    // The NEXT tile's keys are loaded (into the key registers, dead once the tile is staged in LDS) before this tile's
    // stores are issued: they travel while the stores drain and are there when the ranking of the next tile begins.
    int pend_s = -1, pend_k = 0x7fffffff;
    auto ask = [&]() {                                             // tid 0: issue the atomic, do not wait
        pend_s = cand < own + nseg ? cand_seg(cand) : -1;
        pend_k = (pend_s >= 0 && seg_tiles(pend_s) > 0) ? (int)atomicAdd(&P.tickets[pend_s], 1u) : 0x7fffffff;
    };
    auto answer = [&](int *sg) -> int {                            // tid 0: the tile of the pending atomic (or the next one there is)
        if (pend_s >= 0 && pend_k < seg_tiles(pend_s)) { *sg = pend_s; return pend_s * P.tiles_per_seg + pend_k; }
        if (pend_s >= 0) ++cand;
        return take_ticket(sg);
    };
    KeyT key[ITEMS];
    auto load_keys = [&](int tt) {
        const int64_t b = (int64_t)tt * TILE;
        const int vd = (P.n - b) >= TILE ? TILE : (int)(P.n - b);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int e = e0 + j * WAVE;
            key[j] = (vd == TILE || e < vd) ? keys_in[b + e] : (KeyT)~(KeyT)0;
        }
    };
    if (tid == 0) { int sg; s_tile = take_ticket(&sg); s_seg = sg; ask(); }
    __syncthreads();
    int t = s_tile, seg = s_seg;
    if (t >= 0) load_keys(t);
    while (t >= 0) {
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
            if (tid < RADIX) segbase[tid] = dstart + below;
            cur_seg = seg;
            __syncthreads();
        }
        const int seg_first = seg * P.tiles_per_seg;
        const int64_t base = (int64_t)t * TILE;
        const int valid = (P.n - base) >= TILE ? TILE : (int)(P.n - base);
        const bool full = valid == TILE;
        uint32_t val[ITEMS], pos[ITEMS];
        for (int i = tid; i < NWAVES * RADIX / 2; i += THREADS) ((uint32_t *)&wave_hist[0][0])[i] = 0;
        lds_barrier();                         // (the keys' loads are waited for by their first use, not here)
        stamp(0);      // segment switch, counters zeroed, barrier
        // ---- rank inside the wave: lanes with my digit below me (8 ballots + mbcnt), wave totals in LDS ----
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const bool ok = full || (e0 + j * WAVE) < valid;
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
            pos[j] = prior + below;
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(1);      // ranking (includes the wait for the keys)
        // the values are loaded only now (registers), and stay in flight across the LDS-only barriers below
        if (vals_in) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int e = e0 + j * WAVE;
                val[j] = (full || e < valid) ? vals_in[base + e] : 0u;
            }
        } else {                               // no values array: the value is the index itself
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) val[j] = (uint32_t)(base + e0 + j * WAVE);
        }
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
        // publish the count, start the look-back: the whole workgroup reads the granules of the OS_LB_ROWS tiles in front of
        // this one in one go (thread (q, d) takes rows q, q + 4, q + 8, q + 12 of digit d: 32 contiguous KiB); the loads stay in
        // flight while the tile is staged.  With 32 tiles of a segment in flight on an XCD the nearest inclusive prefix is
        // 10 to 20 tiles back: a walk of four tiles per round trip spent a fifth of the tile's time here.
        unsigned long long lb[OS_LB];
        const bool first_of_seg = t == seg_first;
        if (tid < RADIX && !first_of_seg)
            __hip_atomic_store(status + (int64_t)t * RADIX + tid, tagA | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int q = 0; q < OS_LB; ++q) {
            const int k = t - 1 - (q * 4 + (tid >> 8));
            lb[q] = k >= seg_first ? __hip_atomic_load(status + (int64_t)k * RADIX + (tid & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        uint32_t tile_total;
        const uint32_t dbase = block_excl_sum_b<THREADS, true>(tot, scan_lds, &tile_total);
        if (tid < RADIX) digit_base[tid] = dbase;
        lds_barrier();
        stamp(3);      // digit totals, count published, look-back loads issued, tile prefix (3 barriers)
        // ---- keys and values: stage in sorted order ----
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t d = digit_of(key[j], P.shift, P.dmask);
            pos[j] += digit_base[d] + my_hist[d];
            if (full || (e0 + j * WAVE) < valid) lds_k[pos[j]] = key[j];
        }
        // (the values second: their loads -- and, in the look-back waves, the granule loads behind them -- have had the time of
        // the key staging to arrive)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (full || (e0 + j * WAVE) < valid) lds_v[pos[j]] = val[j];
        stamp(4);      // keys and values staged in LDS (includes the wait for the values)
        // ---- tid 0: which tile comes next (the answer of the atomic issued a tile ago), and ask for the one after it ----
        if (tid == 0) { int sg; s_tile = answer(&sg); s_seg = sg; }
        // ---- the granules go to LDS, thread d sums its digit's column backwards until it meets an inclusive prefix ----
#pragma unroll
        for (int q = 0; q < OS_LB; ++q) lbx[q * 4 + (tid >> 8)][tid & 255] = lb[q];
        lds_barrier();
        if (tid < RADIX) {
            uint32_t g = segbase[tid];                              // (first tile of its segment: nothing in front)
            if (!first_of_seg) {
                uint32_t acc = 0;
                bool done = false;
                unsigned spins = 0;
                // a granule that was not there yet when the workgroup looked is polled by its digit's thread
                auto settle = [&](unsigned long long x, int k) -> bool {          // true: inclusive prefix met
                    for (;;) {
                        const unsigned long long tg = x & 0xffffffff00000000ull;
                        if (tg == tagI) { acc += (uint32_t)x; return true; }
                        if (tg == tagA) { acc += (uint32_t)x; return false; }
                        if (++spins > (1u << 22)) { atomicAdd(P.err, 1u); return true; }
                        __builtin_amdgcn_s_sleep(1);
                        x = __hip_atomic_load(status + (int64_t)k * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                };
                int k = t - 1;
                for (int r = 0; r < OS_LB_ROWS && !done && k >= seg_first; ++r, --k) done = settle(lbx[r][tid], k);
                while (!done) {                                     // (rare: further back, four tiles per round trip)
                    if (k < seg_first) { atomicAdd(P.err, 1u); break; }          // (cannot happen: the segment's first tile publishes an inclusive prefix)
                    unsigned long long y[OS_LB];
#pragma unroll
                    for (int q = 0; q < OS_LB; ++q)
                        y[q] = k - q >= seg_first ? __hip_atomic_load(status + (int64_t)(k - q) * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
                    for (int q = 0; q < OS_LB; ++q)
                        if (!done && k - q >= seg_first) done = settle(y[q], k - q);
                    k -= OS_LB;
                }
                g = acc;                                            // (the inclusive prefix the walk ended on carries the segment's start)
            }
            __hip_atomic_store(status + (int64_t)t * RADIX + tid, tagI | (unsigned long long)(g + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            goff[tid] = g - dbase;
            if (count_next) {
                const uint32_t s0 = g / seg_elems;
                seg0[tid] = s0;
                bnd[tid] = (s0 + 1u) * seg_elems;
            }
        }
        // (only now: the atomic's round trip must not sit in front of the look-back's loads in wave 0's memory counter)
        if (tid == 0) ask();
        lds_barrier();
        stamp(5);      // next ticket, look-back finished, barrier
        const int nt = s_tile, nsg = s_seg;
        if (nt >= 0 && !(P.flags & 1u)) load_keys(nt);            // in flight during the stores below
        // ---- LDS -> global, digit runs coalesced; the next pass's digit is counted on the way ----
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int idx = tid + j * THREADS;
            const bool ok = full || idx < valid;
            KeyT kx = 0;
            uint32_t d = 0, gp = 0;
            if (ok) {
                kx = lds_k[idx];
                d = digit_of(kx, P.shift, P.dmask);
                gp = goff[d] + (uint32_t)idx;
                keys_out[gp] = kx;
                vals_out[gp] = lds_v[idx];
            }
            if (count_next) {
                const uint32_t slot = ok ? digit_of(kx, P.shift_next, P.dmask_next) * OS_NSEG + seg0[d] + (gp >= bnd[d] ? 1u : 0u) : 0u;
                const uint64_t act = __ballot(ok);
                const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
                // one (digit, segment) for the whole wave (constant high digits, runs): one add, not 64 on one address
                // (lanes past the end of a ragged tile are the wave's last ones: lane 0 is valid whenever any lane is)
                if (__all(!ok || slot == f)) {
                    if (act && l == 0) atomicAdd(&hist2[f], (uint32_t)__popcll(act));
                } else if (ok) atomicAdd(&hist2[slot], 1u);
            }
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (four items' LDS reads in flight at a time: the next tile's keys hold ITEMS registers)
        }
        stamp(6);      // LDS -> global
        lds_barrier();                         // the LDS stage is free (s_tile / s_seg are rewritten two barriers from here)
        if (nt >= 0 && (P.flags & 1u)) load_keys(nt);             // (A/B: the keys only behind the stores)
        stamp(7);      // barrier + next tile's key loads issued
        t = nt; seg = nsg;
    }
    // ---- the next pass's counts: [digit][segment] ----
    if (count_next) {
        __syncthreads();
        for (int i = tid; i < RADIX * OS_NSEG; i += THREADS) {
            const uint32_t c = hist2[i];
            const int d = i / OS_NSEG, s = i % OS_NSEG;
            if (c && s < nseg) atomicAdd(&P.hist_next[d * nseg + s], c);
        }
    }
}

}  // namespace sa
