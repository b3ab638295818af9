// kernels/induce_proto.hpp -- DIAGNOSTIC ONLY (-DSA_AMD_DIAG): micro-prototype of ONE level-0 L-type induce sweep of SA-IS
// as BASELINE.json's north_star sketches it: a wavefront-serialised pass over SA[] that coalesces the reads of SA, gathers
// text[SA[i] - 1], and keeps the 256 bucket heads in LDS.  Exact (its output is compared with the true suffix array by
// tools/induce_proto.py); exists to MEASURE what DESIGN.md section 2 argued: how long a sequential-recurrence sweep
// takes on an MI355X.  Not part of the product library.
//
// The sweep: for i = 0 .. n: v = SA[i]; if v is a suffix (not EMPTY) and v > 0 and suffix v - 1 is L-type:
// SA[head[T[v-1]]++] = v - 1.  Step i may write a slot that a later step reads -- the recurrence.  One wave takes 64
// consecutive slots at a time: lanes that induce a suffix find their rank among the lanes of the same bucket with
// ballots (lane order = slot order), the bucket head lives in LDS.  A suffix induced into the block that is being
// processed (or into the prefetched next one) is picked up by re-reading those slots before moving on.
#pragma once
#include "common.hpp"

namespace sa {

constexpr uint32_t IND_EMPTY = 0xffffffffu;

__device__ __forceinline__ uint32_t ind_load(const uint32_t *p)        // bypasses the wave's L1: it re-reads what it stored
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// SA: n + 1 slots (slot 0 = the empty suffix n); LMS suffixes sit at their places, everything else is IND_EMPTY.
// typeL: bit j = suffix j is L-type.  head: 256 bucket starts (slot indices).  counters[0] = induced suffixes,
// counters[1] = blocks that had to be re-read.
__global__ __launch_bounds__(64) void k_induce_l_sweep(const uint8_t *__restrict__ T, const uint8_t *__restrict__ typeL,
                                                        uint32_t *SA, int64_t n, const uint32_t *__restrict__ head0,
                                                        unsigned long long *__restrict__ counters)
{
    __shared__ uint32_t head[256];
    const int l = threadIdx.x;
    for (int c = l; c < 256; c += 64) head[c] = head0[c];
    __syncthreads();
    unsigned long long induced = 0, reread = 0;
    const int64_t slots = n + 1;
    uint32_t nextv = l < slots ? ind_load(SA + l) : IND_EMPTY;        // prefetched block
    for (int64_t i0 = 0; i0 < slots; i0 += 64) {
        uint32_t v = nextv;
        const int64_t in = i0 + 64 + l;
        nextv = in < slots ? ind_load(SA + in) : IND_EMPTY;             // issue the next block's load now
        int cur = 0;                                                     // lanes below cur are finished
        for (;;) {
            const bool have = v != IND_EMPTY && l >= cur;
            bool ind = false;
            uint32_t j = 0, c1 = 0;
            if (have && v > 0) {
                j = v - 1;
                ind = (typeL[j >> 3] >> (j & 7)) & 1;
                if (ind) c1 = T[j];
            }
            const uint64_t im = __ballot(ind);
            if (!im) break;
            // rank among the inducing lanes of the same bucket (lower lanes first = slot order)
            uint32_t xlo = ~(uint32_t)im, xhi = ~(uint32_t)(im >> 32);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint32_t sel = (uint32_t)((int32_t)(c1 << (31 - b)) >> 31);
                const uint64_t bal = __ballot(ind && sel != 0);
                xlo |= (uint32_t)bal ^ sel;
                xhi |= (uint32_t)(bal >> 32) ^ sel;
            }
            const uint32_t mlo = ~xlo, mhi = ~xhi;
            uint32_t pos = 0, below = 0, base = 0;
            if (ind) {
                below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                base = head[c1];
                pos = base + below;
            }
            // the first slot of THIS block that one of these inductions fills: only the lanes before it may commit now, the
            // slot itself (and everything behind it) is taken up again after a re-read -- strict slot order is the semantics
            int tmin = (ind && (int64_t)pos < i0 + 64) ? (int)((int64_t)pos - i0) : 64;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tmin = min(tmin, __shfl_xor(tmin, o, WAVE));
            const bool commit = ind && l < tmin;
            const uint64_t cm = __ballot(commit);
            if (commit) SA[pos] = j;
            if (ind && below == 0 && l < tmin) head[c1] = base + (uint32_t)(__popc(mlo & (uint32_t)cm) + __popc(mhi & (uint32_t)(cm >> 32)));
            induced += (unsigned long long)__popcll(cm);
            const bool into_next = commit && (int64_t)pos >= i0 + 64 && (int64_t)pos < i0 + 128;
            const bool stale_next = __ballot(into_next) != 0;
            if (tmin == 64 && !stale_next) break;
            ++reread;
            __builtin_amdgcn_s_waitcnt(0);                               // the wave's own stores are out
            __threadfence();
            if (stale_next) nextv = in < slots ? ind_load(SA + in) : IND_EMPTY;
            if (tmin == 64) break;
            cur = tmin;
            const int64_t ii = i0 + l;
            v = ii < slots ? ind_load(SA + ii) : IND_EMPTY;
        }
    }
    if (l == 0) { counters[0] = induced; counters[1] = reread; }
}

}  // namespace sa
