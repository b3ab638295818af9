// kernels/rerank.hpp -- group heads -> ranks, SA write-back, compaction of the suffixes still tied with a neighbour.
// Part of the MI355X-native suffix-array engine (gfx950 / CDNA4, wave64); see DESIGN.md section 3.
#pragma once
#include "common.hpp"

namespace sa {

// ------------------------------------------------------------------------------------------
// Re-rank: m sorted (key, suffix) pairs sitting in slots U[0..m) of SA (FIRST: U[j] = j).
// A group = maximal run of equal keys; its rank is (slot of its first element) + 1 -- in the dense
// doubling rounds (slot of its LAST element) + 1, see TAIL in k_rr_apply.
//   k_rr_count : per tile, how many elements stay tied with a neighbour, and the last group
//                head slot (+1) inside the tile; dense rounds: also the tile's first group start
//   k_rr_scan  : exclusive sum / exclusive max over the tiles (one workgroup)
//   k_rr_scan_next : dense rounds: for every tile the first group start behind it (one workgroup)
//   k_rr_apply : SA[U[j]] = V[j]; ISA[V[j]] = rank; compact (slot, group head, suffix) of the
//                elements that are still tied
// Algorithmic traffic per element: 12 B read twice (keys + vals [+ 4 B slot]), 4 B SA write,
// 4 B ISA scatter, 12 B per surviving element.
// ------------------------------------------------------------------------------------------
constexpr int RR_THREADS = 1024;
constexpr int RR_ITEMS = 8;
constexpr int RR_TILE = RR_THREADS * RR_ITEMS;   // 8192

constexpr int RR_WAVE_ELEMS = WAVE * RR_ITEMS;   // 512 consecutive elements per wave, item r of lane l = base + 64 r + l

// Head / tied masks of one wave's 512 elements.  Loads are wave-striped (512 contiguous bytes per
// instruction); the neighbour keys come from shuffles, so the whole classification is 8 ballots
// and scalar bit arithmetic: head[r] bit l = element (r, l) starts a group, tied[r] bit l = it is
// in a group of more than one element, valid[r] = it exists.
// PARENTS: phead[r] bit l = element (r, l) is the first one of its PARENT group (the group the round started from: a run
// of equal key >> g_shift) -- the dense doubling rounds label a group by its LAST slot and leave the ranks of a parent's
// last subgroup alone (see k_rr_apply)
struct WaveGroups { uint64_t head[RR_ITEMS], tied[RR_ITEMS], valid[RR_ITEMS], phead[RR_ITEMS]; };

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src, WAVE) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src, WAVE);
}

template <typename KeyT, bool PARENTS = false>
__device__ __forceinline__ WaveGroups rr_wave_classify(const KeyT *__restrict__ keys, int64_t m, int64_t wbase, int key_shift, int g_shift = 0,
                                                       uint64_t *kout = nullptr)      // kout: the wave's keys, [RR_ITEMS] = the key before the wave
{
    const int l = lane_id();
    uint64_t k[RR_ITEMS];
    WaveGroups g;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        k[r] = i < m ? ((uint64_t)keys[i] >> key_shift) : 0ull; // groups are runs of equal key >> key_shift
        g.valid[r] = __ballot(i < m);
    }
    const uint64_t before = (wbase > 0 && wbase - 1 < m) ? ((uint64_t)keys[wbase - 1] >> key_shift) : 0ull;
    const uint64_t after = (wbase + RR_WAVE_ELEMS < m) ? ((uint64_t)keys[wbase + RR_WAVE_ELEMS] >> key_shift) : 0ull;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        uint64_t up = shfl64(k[r], l ? l - 1 : 0);
        const uint64_t prev_last = r ? shfl64(k[r ? r - 1 : 0], 63) : before;   // executed by every lane (no shuffle under divergence)
        if (l == 0) up = prev_last;
        const int64_t i = wbase + 64 * r + l;
        g.head[r] = __ballot(i < m && (i == 0 || k[r] != up));
        g.phead[r] = PARENTS ? __ballot(i < m && (i == 0 || ((k[r] ^ up) >> g_shift) != 0ull)) : 0ull;
    }
    if (kout) {
#pragma unroll
        for (int r = 0; r < RR_ITEMS; ++r) kout[r] = k[r];
        kout[RR_ITEMS] = before;
    }
    const uint64_t last_key = shfl64(k[RR_ITEMS - 1], 63);
    const bool boundary_after = (wbase + RR_WAVE_ELEMS >= m) || after != last_key;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        // is the NEXT element a group boundary (a head, or past the end)?
        const uint64_t bnd = g.head[r] | ~g.valid[r];
        const uint64_t bnd_next0 = (r + 1 < RR_ITEMS) ? ((g.head[(r + 1) % RR_ITEMS] | ~g.valid[(r + 1) % RR_ITEMS]) & 1ull)
                                                      : (boundary_after ? 1ull : 0ull);
        const uint64_t next = (bnd >> 1) | (bnd_next0 << 63);
        g.tied[r] = g.valid[r] & ~(g.head[r] & next);
    }
    return g;
}

// "head code" of the dense doubling rounds: (list index of a group start << 1) | (it also starts a parent group);
// RR_NO_HEAD: there is none (list indices are below 2^31, so the code of index m with the parent bit set is the same word)
constexpr uint32_t RR_NO_HEAD = 0xffffffffu;
constexpr int RR_CHG_COUNTERS = 16;             // counters every workgroup of a large launch bumps (changed ranks of a dense round, tied slots of
                                                // k_finish_sorted) are spread over this many words, 32 words apart: a single word serialises in one L2 channel
__device__ __forceinline__ uint32_t rr_first_head_code(const WaveGroups &g, int64_t wbase, int r_from)
{
    uint32_t code = RR_NO_HEAD;
#pragma unroll
    for (int r = RR_ITEMS - 1; r >= 0; --r) {
        if (r >= r_from && g.head[r]) {
            const int b = __builtin_ctzll(g.head[r]);
            code = ((uint32_t)(wbase + 64 * r + b) << 1) | (uint32_t)((g.phead[r] >> b) & 1ull);
        }
    }
    return code;
}

// PARENTS (dense doubling rounds): also tile_first, the head code of the tile's first group start (g_shift: the parent
// group is the key above that bit)
template <bool FIRST, typename KeyT = uint64_t, bool PARENTS = false>
__global__ __launch_bounds__(RR_THREADS) void k_rr_count(const KeyT *__restrict__ keys,
                                                          const uint32_t *__restrict__ U, int64_t m,
                                                          uint32_t *__restrict__ tile_cnt,
                                                          uint32_t *__restrict__ tile_head, int key_shift,
                                                          uint32_t *__restrict__ tile_first, int g_shift,
                                                          const uint32_t *__restrict__ gate = nullptr)      // != nullptr: the launch does nothing when *gate != 0 (see RoundCtl, host/pipeline.hpp)
{
    __shared__ uint32_t wcnt[RR_THREADS / WAVE], whead[RR_THREADS / WAVE], wfirst[RR_THREADS / WAVE];
    if (gate && *gate) return;
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)wave_id() * RR_WAVE_ELEMS;
    uint64_t kk[RR_ITEMS + 1];
    const WaveGroups g = rr_wave_classify<KeyT, false>(keys, m, wbase, key_shift, 0, PARENTS ? kk : nullptr);
    uint32_t cnt = 0, lasthead = 0;
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        cnt += (uint32_t)__popcll(g.tied[r]);
        if (g.head[r]) {
            const int64_t i = wbase + 64 * r + (63 - __builtin_clzll(g.head[r]));
            lasthead = (FIRST ? (uint32_t)i : U[i]) + 1u;       // slots grow with the index: the last head wins
        }
    }
    // the wave's first group start, and whether its parent group starts there too: only that one element's key and its left
    // neighbour's are compared (wave-uniform item and lane: two shuffles, not a ballot per item)
    uint32_t first_code = RR_NO_HEAD;
    if (PARENTS) {
        bool found = false;
#pragma unroll
        for (int r = 0; r < RR_ITEMS; ++r) {
            if (!found && g.head[r]) {                          // (uniform)
                found = true;
                const int b = __builtin_ctzll(g.head[r]);
                const uint64_t mine = shfl64(kk[r], b);
                const uint64_t left = b ? shfl64(kk[r], b - 1) : (r ? shfl64(kk[r ? r - 1 : 0], 63) : kk[RR_ITEMS]);
                const int64_t i = wbase + 64 * r + b;
                const uint32_t ph = (i == 0 || ((mine ^ left) >> g_shift) != 0ull) ? 1u : 0u;
                first_code = ((uint32_t)i << 1) | ph;
            }
        }
    }
    if (lane_id() == 0) { wcnt[wave_id()] = cnt; whead[wave_id()] = lasthead; wfirst[wave_id()] = first_code; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0, mx = 0, fc = RR_NO_HEAD;
        for (int w = 0; w < RR_THREADS / WAVE; ++w) { tot += wcnt[w]; mx = mx > whead[w] ? mx : whead[w]; fc = fc < wfirst[w] ? fc : wfirst[w]; }
        tile_cnt[blockIdx.x] = tot;
        tile_head[blockIdx.x] = mx;
        if (PARENTS) tile_first[blockIdx.x] = fc;
    }
}

// one workgroup: tile_first (head code of every tile's first group start) -> for every tile the smallest code of the tiles
// BEHIND it (RR_NO_HEAD: no group starts behind this tile); also clears the round's changed-rank counter
__device__ __forceinline__ void rr_scan_next_body(uint32_t *__restrict__ tile_first, int64_t tiles, uint32_t *__restrict__ changed_cnt)
{
    __shared__ uint32_t s_min[SPINE_THREADS];
    const int t = threadIdx.x;
    const int64_t per = (tiles + SPINE_THREADS - 1) / SPINE_THREADS;
    int64_t b = (int64_t)t * per, e = b + per;
    if (b > tiles) b = tiles;
    if (e > tiles) e = tiles;
    uint32_t mn = RR_NO_HEAD;
    for (int64_t i = b; i < e; ++i) { const uint32_t c = tile_first[i]; mn = mn < c ? mn : c; }
    s_min[t] = mn;
    __syncthreads();
    for (int o = 1; o < SPINE_THREADS; o <<= 1) {            // inclusive suffix minimum over the threads
        const uint32_t other = t + o < SPINE_THREADS ? s_min[t + o] : RR_NO_HEAD;
        __syncthreads();
        if (other < s_min[t]) s_min[t] = other;
        __syncthreads();
    }
    uint32_t run = t + 1 < SPINE_THREADS ? s_min[t + 1] : RR_NO_HEAD;     // everything behind this thread's tiles
    for (int64_t i = e - 1; i >= b; --i) {
        const uint32_t c = tile_first[i];
        tile_first[i] = run;
        run = run < c ? run : c;
    }
    if (t < RR_CHG_COUNTERS && changed_cnt) changed_cnt[t * 32] = 0u;
}

__global__ __launch_bounds__(SPINE_THREADS) void k_rr_scan_next(uint32_t *__restrict__ tile_first, int64_t tiles, uint32_t *__restrict__ changed_cnt)
{
    rr_scan_next_body(tile_first, tiles, changed_cnt);
}

// one workgroup: tile_cnt -> exclusive sums (+ total), tile_head -> exclusive running max.
// Every thread owns a contiguous run of entries (a multiple of 4, read and written as 16-byte vectors:
// the run is a chain of dependent L2 accesses, so fewer, wider ones).
// (tile_head may be nullptr: sums only)
__device__ __forceinline__ void rr_scan_body(uint32_t *__restrict__ tile_cnt, uint32_t *__restrict__ tile_head, int64_t tiles,
                                             uint32_t *__restrict__ out_total)
{
    __shared__ uint32_t lds[SPINE_THREADS / WAVE + 1];
    const int64_t per = ((tiles + SPINE_THREADS - 1) / SPINE_THREADS + 3) & ~(int64_t)3;
    int64_t b = (int64_t)threadIdx.x * per, e = b + per;
    if (b > tiles) b = tiles;
    if (e > tiles) e = tiles;
    const int64_t ev = b + ((e - b) & ~(int64_t)3);           // end of the whole vectors
    const bool heads = tile_head != nullptr;                  // (uniform)
    uint32_t s = 0, mx = 0;
    for (int64_t i = b; i < ev; i += 4) {
        const uint4 c = *(const uint4 *)(tile_cnt + i);
        s += c.x + c.y + c.z + c.w;
        if (heads) { const uint4 h = *(const uint4 *)(tile_head + i); mx = max(max(mx, max(h.x, h.y)), max(h.z, h.w)); }
    }
    for (int64_t i = ev; i < e; ++i) { s += tile_cnt[i]; if (heads) mx = max(mx, tile_head[i]); }
    uint32_t tot, mtot;
    uint32_t off = block_excl_sum<SPINE_THREADS>(s, lds, &tot);
    uint32_t incm = block_incl_max<SPINE_THREADS>(mx, lds, &mtot);
    // exclusive max for this thread = inclusive max of the previous thread
    uint32_t prevm = __shfl_up(incm, 1, WAVE);
    __shared__ uint32_t wlast[SPINE_THREADS / WAVE];
    if (lane_id() == WAVE - 1) wlast[wave_id()] = incm;
    __syncthreads();
    if (lane_id() == 0) prevm = wave_id() ? wlast[wave_id() - 1] : 0u;
    uint32_t run = prevm;
    for (int64_t i = b; i < ev; i += 4) {
        const uint4 c = *(const uint4 *)(tile_cnt + i);
        uint4 oc;
        oc.x = off; off += c.x; oc.y = off; off += c.y; oc.z = off; off += c.z; oc.w = off; off += c.w;
        *(uint4 *)(tile_cnt + i) = oc;
        if (heads) {
            const uint4 h = *(const uint4 *)(tile_head + i);
            uint4 oh;
            oh.x = run; run = max(run, h.x); oh.y = run; run = max(run, h.y); oh.z = run; run = max(run, h.z); oh.w = run; run = max(run, h.w);
            *(uint4 *)(tile_head + i) = oh;
        }
    }
    for (int64_t i = ev; i < e; ++i) {
        const uint32_t c = tile_cnt[i];
        tile_cnt[i] = off; off += c;
        if (heads) { const uint32_t h = tile_head[i]; tile_head[i] = run; run = run > h ? run : h; }
    }
    if (threadIdx.x == 0) *out_total = tot;
}

__global__ __launch_bounds__(SPINE_THREADS) void k_rr_scan(uint32_t *__restrict__ tile_cnt,
                                                            uint32_t *__restrict__ tile_head, int64_t tiles,
                                                            uint32_t *__restrict__ out_total)
{
    rr_scan_body(tile_cnt, tile_head, tiles, out_total);
}

// two independent scans in one launch (grid 2): a launch of a refinement round on a short list is ~5 us whatever it does
__global__ __launch_bounds__(SPINE_THREADS) void k_rr_scan_pair(uint32_t *__restrict__ cntA, uint32_t *__restrict__ headA, uint32_t *__restrict__ outA,
                                                                 uint32_t *__restrict__ cntB, uint32_t *__restrict__ headB, uint32_t *__restrict__ outB,
                                                                 int64_t tiles)
{
    if (blockIdx.x == 0) rr_scan_body(cntA, headA, tiles, outA);
    else rr_scan_body(cntB, headB, tiles, outB);
}

// the two single-workgroup steps between k_rr_count and k_rr_apply of a round in one launch (grid 2): block 0 = k_rr_scan,
// block 1 = k_rr_scan_next (tile_first == nullptr: none) + the two counters the NEXT round's local pass starts from (clear2[0..1],
// copied to clear2[2..3] first)
__global__ __launch_bounds__(SPINE_THREADS) void k_rr_scan_round(uint32_t *__restrict__ tile_cnt, uint32_t *__restrict__ tile_head, int64_t tiles,
                                                                  uint32_t *__restrict__ out_total, uint32_t *__restrict__ tile_first,
                                                                  uint32_t *__restrict__ changed_cnt, uint32_t *__restrict__ clear2,
                                                                  const uint32_t *__restrict__ gate)
{
    if (gate && *gate) return;
    if (blockIdx.x == 0) { rr_scan_body(tile_cnt, tile_head, tiles, out_total); return; }
    if (tile_first) rr_scan_next_body(tile_first, tiles, changed_cnt);
    if (threadIdx.x == 0 && clear2) { clear2[2] = clear2[0]; clear2[3] = clear2[1]; clear2[0] = 0u; clear2[1] = 0u; }      // (saved for the round's one read-back, then zeroed)
}

// ISA_MODE: 0 = scatter ISA[suffix] = rank directly, 1 = the same plus the has_isa bitmap (sparse
// refinement), 3 = no rank output at all (text-keyed rounds), 4 = like 3, but the still-tied suffixes are recorded by slot
// (ISA = group heads, has_isa = bitmap, pair_v = counts per tile) instead of being listed, 2 = write (suffix, rank) pairs in slot order; the host bins them by suffix position
// with one radix pass and k_scatter_pairs then writes the ISA window by window (a random 4-byte
// store costs a whole 64-byte memory transaction, a binned one is merged in the caches).
template <bool FIRST, bool WRITE_SA, int ISA_MODE, typename KeyT = uint64_t, bool FTAIL = false>
__global__ __launch_bounds__(RR_THREADS) void k_rr_apply(
    const KeyT *__restrict__ keys, const uint32_t *__restrict__ V, const uint32_t *__restrict__ U, int64_t m,
    const uint32_t *__restrict__ tile_cnt, const uint32_t *__restrict__ tile_head, uint32_t *__restrict__ SA,
    uint32_t *__restrict__ ISA, uint32_t *__restrict__ Uo, uint32_t *__restrict__ Go, uint32_t *__restrict__ Vo,
    uint32_t n_text, uint32_t *__restrict__ has_isa, int g_shift, uint64_t *__restrict__ pair_k,
    uint32_t *__restrict__ pair_v, const uint32_t *__restrict__ tile_total, int key_shift,
    const uint32_t *__restrict__ tile_next, int parent_tail, uint32_t *__restrict__ changed_cnt,
    const uint32_t *__restrict__ gate = nullptr)      // != nullptr: the launch does nothing when *gate != 0
{
    constexpr bool SPARSE = ISA_MODE == 1;
    if (gate && *gate) return;
    // Dense doubling rounds (TAIL): the rank of a group is its LAST slot + 1 (Larsson-Sadakane's group number).  When a
    // parent group splits, its last subgroup ends where the parent ended, so its members' ranks in the ISA are already
    // right and are not rewritten.  A run, a periodic text or a long repeat that runs into the end of the text sheds its
    // shortest -- smallest -- suffixes at the FRONT of every group round after round: with ranks counted from the first
    // slot every one of the hundreds of millions of remaining members changed rank every round, with ranks counted from
    // the last slot only the members that leave do.  parent_tail = 0: the parents' ranks are not of this form (first
    // doubling round: ranks from the initial order), everything is written.  changed_cnt: ranks written (steers the
    // host's choice between binned and direct ISA stores in the next round).
    // FTAIL: the FIRST ranks (set-up of the dense route, straight from the initial order) are of this form already, so the first
    // doubling round leaves alone every group that does not split and every parent's last subgroup -- on a corpus with copied
    // passages that is most of what stays tied.  (tile_next then comes from k_rr_count<true, ., true> + k_rr_scan_next.)
    constexpr bool TAIL = (FIRST ? FTAIL : true) && (ISA_MODE == 0 || ISA_MODE == 2);
    constexpr bool COUNT_CHANGED = TAIL && !FIRST;      // (the first ranks are all new: nothing to count)
    constexpr int NW = RR_THREADS / WAVE;
    __shared__ uint32_t wcnt[NW], whead[NW], wfirst[NW];
    __shared__ uint32_t s_chg;
    if (COUNT_CHANGED && threadIdx.x == 0) s_chg = 0;          // (the barrier behind the wave totals orders it before the adds)
    if (FIRST && !WRITE_SA && SPARSE) {
        // compaction-only pass (no SA, no ISA write): a tile without tied suffixes has nothing to do
        const uint32_t here = tile_cnt[blockIdx.x];
        const uint32_t next = (blockIdx.x + 1 < gridDim.x) ? tile_cnt[blockIdx.x + 1] : *tile_total;
        if (next == here) return;
    }
    const int l = lane_id(), w = wave_id();
    const int64_t wbase = (int64_t)blockIdx.x * RR_TILE + (int64_t)w * RR_WAVE_ELEMS;
    const WaveGroups g = rr_wave_classify<KeyT, TAIL>(keys, m, wbase, key_shift, g_shift);
    uint32_t slot[RR_ITEMS], v[RR_ITEMS];
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        const bool in = i < m;
        slot[r] = FIRST ? (uint32_t)i : (in ? U[i] : 0u);
        v[r] = in ? V[i] : 0u;
    }
    // this wave's totals -> offsets of the waves of the tile
    {
        uint32_t cnt = 0, lasthead = 0;
#pragma unroll
        for (int r = 0; r < RR_ITEMS; ++r) {
            cnt += (uint32_t)__popcll(g.tied[r]);
            if (g.head[r]) lasthead = (uint32_t)__shfl((int)slot[r], 63 - __builtin_clzll(g.head[r]), WAVE) + 1u;
        }
        if (l == 0) { wcnt[w] = cnt; whead[w] = lasthead; wfirst[w] = TAIL ? rr_first_head_code(g, wbase, 0) : RR_NO_HEAD; }
    }
    __syncthreads();
    uint32_t run_cnt = tile_cnt[blockIdx.x], run_head = tile_head[blockIdx.x];     // carried in from the tiles before
    for (int ww = 0; ww < w; ++ww) { run_cnt += wcnt[ww]; run_head = run_head > whead[ww] ? run_head : whead[ww]; }
    // TAIL: next_code[r] = head code of the first group start behind item r of this wave (later items, later waves, later tiles;
    // none: the list ends there, which also ends the parent)
    uint32_t next_code[RR_ITEMS];
    uint32_t n_changed = 0;
    if (TAIL) {
        uint32_t behind = tile_next[blockIdx.x];
        for (int ww = NW - 1; ww > w; --ww) behind = wfirst[ww] != RR_NO_HEAD ? wfirst[ww] : behind;
        if (behind == RR_NO_HEAD) behind = ((uint32_t)m << 1) | 1u;
#pragma unroll
        for (int r = RR_ITEMS - 1; r >= 0; --r) {
            next_code[r] = behind;
            if (g.head[r]) {
                const int b = __builtin_ctzll(g.head[r]);
                behind = ((uint32_t)(wbase + 64 * r + b) << 1) | (uint32_t)((g.phead[r] >> b) & 1ull);
            }
        }
    }
    const uint64_t le_mask = (l == 63) ? ~0ull : ((2ull << l) - 1ull);              // lanes <= l
    const uint64_t lt_mask = le_mask >> 1;                                          // lanes <  l
    // head code of the first group start behind element (r, l)
    auto code_behind = [&](int r) -> uint32_t {
        const uint64_t hgt = (l == 63) ? 0ull : (g.head[r] & (~0ull << (l + 1)));   // group starts behind this lane in the item
        if (!hgt) return next_code[r];
        const int b = __builtin_ctzll(hgt);
        return ((uint32_t)(wbase + 64 * r + b) << 1) | (uint32_t)((g.phead[r] >> b) & 1ull);
    };
    // PAIRS (dense rounds that bin their ISA writes): only the ranks that change become (suffix, rank) pairs, packed: the
    // workgroup counts them, takes a block of the pair arrays with ONE atomic on changed_cnt[0] (which so also ends up
    // as the number of pairs) and its waves fill it in order
    constexpr bool PAIRS = COUNT_CHANGED && ISA_MODE == 2;
    uint64_t cmask[PAIRS ? RR_ITEMS : 1];
    uint32_t pair_off = 0;
    if (PAIRS) {
        __shared__ uint32_t wpairs[NW];
        __shared__ uint32_t s_base;
        uint32_t mine = 0;
#pragma unroll
        for (int r = 0; r < RR_ITEMS; ++r) {
            const int64_t i = wbase + 64 * r + l;
            cmask[r] = __ballot(i < m && !(parent_tail && (code_behind(r) & 1u)));
            mine += (uint32_t)__popcll(cmask[r]);
        }
        if (l == 0) wpairs[w] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int ww = 0; ww < NW; ++ww) tot += wpairs[ww];
            s_base = tot ? atomicAdd(&changed_cnt[0], tot) : 0u;
        }
        __syncthreads();
        pair_off = s_base;
        for (int ww = 0; ww < w; ++ww) pair_off += wpairs[ww];
    }
#pragma unroll
    for (int r = 0; r < RR_ITEMS; ++r) {
        const int64_t i = wbase + 64 * r + l;
        // rank = (slot of the group's first element) + 1: the last head at or before this lane, else the carry
        const uint64_t hle = g.head[r] & le_mask;
        const int src = hle ? 63 - __builtin_clzll(hle) : 0;
        const uint32_t hslot = (uint32_t)__shfl((int)slot[r], src, WAVE);
        const uint32_t run = hle ? hslot + 1u : run_head;
        const uint32_t off = run_cnt + (uint32_t)__popcll(g.tied[r] & lt_mask);
        uint32_t rank = run;                      // the value that goes into the ISA
        bool changed = true;
        if (TAIL) {
            const uint32_t code = code_behind(r);
            // the group's slots are consecutive: its last one = my slot + (members behind me)
            rank = slot[r] + ((code >> 1) - 1u - (uint32_t)i) + 1u;
            changed = !(parent_tail && (code & 1u));
            if (COUNT_CHANGED && !PAIRS) n_changed += (uint32_t)__popcll(__ballot(i < m && changed));
        }
        if (i < m) {
            if (WRITE_SA && slot[r] < n_text) SA[slot[r]] = v[r];
            if (PAIRS) {
                if (changed) {
                    const uint32_t o = pair_off + (uint32_t)__popcll(cmask[r] & lt_mask);
                    ((uint32_t *)pair_k)[o] = v[r];      // (suffix, rank) as 32-bit pairs: binned by 32-bit radix passes
                    pair_v[o] = rank;
                }
            } else if (ISA_MODE == 2) {
                ((uint32_t *)pair_k)[i] = v[r];
                pair_v[i] = rank;
            } else if (ISA_MODE != 3 && ISA_MODE != 4 && v[r] < n_text) {
                if (changed) {
                    ISA[v[r]] = rank;
                    if (SPARSE) atomicOr(&has_isa[v[r] >> 5], 1u << (v[r] & 31u));   // this rank overrides the initial one
                }
            }
            if ((g.tied[r] >> l) & 1ull) {
                if (ISA_MODE == 4) {
                    // survivors join those of k_finish_sorted: bitmap + group head by slot (k_surv_compact lists them in slot order)
                    ISA[slot[r]] = run - 1u;
                    atomicOr(&has_isa[slot[r] >> 5], 1u << (slot[r] & 31u));
                    atomicAdd(&pair_v[slot[r] / RR_TILE], 1u);
                } else {
                    Uo[off] = slot[r]; Go[off] = run - 1u; Vo[off] = v[r];
                }
            }
        }
        run_cnt += (uint32_t)__popcll(g.tied[r]);
        if (g.head[r]) run_head = (uint32_t)__shfl((int)slot[r], 63 - __builtin_clzll(g.head[r]), WAVE) + 1u;
        if (PAIRS) pair_off += (uint32_t)__popcll(cmask[r]);
    }
    if (COUNT_CHANGED && !PAIRS) {
        // one global atomic per workgroup, spread over RR_CHG_COUNTERS words in different 128-byte lines (a single
        // counter bumped by every wave serialises in one L2 channel: 465 K atomics cost 4 ms at 256 MiB)
        if (l == 0 && n_changed) atomicAdd(&s_chg, n_changed);
        __syncthreads();
        if (threadIdx.x == 0 && s_chg) atomicAdd(&changed_cnt[(blockIdx.x % RR_CHG_COUNTERS) * 32], s_chg);
    }
}

}  // namespace sa
