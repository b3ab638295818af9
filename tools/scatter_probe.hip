// scatter_probe.hip -- what does a radix tile scatter cost on MI355X by WHO writes the neighbouring runs, and what does
// a decoupled look-back (single-pass "onesweep" scatter) add?  Informs the sort engine behind sa_amd_saca_u8 (the arithmetic
// that replaces cdivsufsort::sort_in_place, reference src/saca.rs:14).  Synthetic: the digit of every element is drawn
// from a seeded hash, tiles are "already ranked" (an element's place inside its digit run is its index), so only the
// memory side is measured: one coalesced read of the keys, scattered stores of (key, value) runs.
//   mode 0  chunked      workgroup g owns tiles [g k, (g+1) k): a line shared by two tiles is completed by the SAME CU one
//                        tile later (the first-generation scatter of this engine, no carries)
//   mode 1  ticket       tiles in global ticket order: neighbouring tiles run at the same time on different XCDs
//   mode 2  xcd ticket   tiles in ticket order inside 8 * SEG segments, a workgroup prefers the segments of its own XCD
//                        (HW_REG_XCC_ID), so neighbouring tiles meet in ONE L2; steals from other segments when out of work
//   mode 3  sequential   output position = input position (the ceiling of this loop structure)
//   mode 4  = 1 with the offsets coming from a real decoupled look-back (8-byte {tag, value} granules, agent scope)
//   mode 5  = 2 with the look-back inside each segment (segment bases known beforehand)
//   build:  hipcc -O3 --offload-arch=gfx950 -o tools/bin/scatter_probe tools/scatter_probe.hip
//   run:    tools/bin/scatter_probe [log2 n = 28] [key bytes = 8] [items per thread = 8] [skew = 0] [no values = 0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int THREADS = 1024;
constexpr int RADIX = 256;
constexpr int NSEG_MAX = 64;

__host__ __device__ inline uint32_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}
__host__ __device__ inline uint32_t digit_of_elem(uint64_t i, int skew)
{
    const uint32_t r = mix(i * 2 + 1);
    if (!skew) return r & 255u;
    // skewed: product of two uniform bytes >> 8 (small digits frequent, like a low-entropy text)
    return ((r & 255u) * ((r >> 8) & 255u)) >> 8;
}

// counts[t][d] of every tile
__global__ __launch_bounds__(THREADS) void k_counts(uint32_t *counts, int64_t n, int tile, int skew)
{
    __shared__ uint32_t h[RADIX];
    if (threadIdx.x < RADIX) h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * tile;
    for (int e = threadIdx.x; e < tile && base + e < n; e += THREADS) atomicAdd(&h[digit_of_elem(base + e, skew)], 1u);
    __syncthreads();
    if (threadIdx.x < RADIX) counts[(int64_t)blockIdx.x * RADIX + threadIdx.x] = h[threadIdx.x];
}

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct Params {
    const uint32_t *counts;    // [tiles][256]
    const uint32_t *offs;      // [tiles][256] exclusive global offsets (precomputed on the host)
    const uint32_t *segbase;   // [nseg][256]  global offset of the segment's first element of digit d (mode 5)
    unsigned long long *status;  // [tiles][256] granules
    uint32_t *tickets;         // [0] global ticket, [16 * (1 + s)] ticket of segment s
    uint32_t *err;
    int64_t n;
    int tile, tiles, mode, nseg, tiles_per_seg, tiles_per_wg, epoch, novals;
};

template <typename KeyT, int ITEMS>
__global__ __launch_bounds__(THREADS) void k_scatter(const KeyT *__restrict__ kin, KeyT *__restrict__ kout, uint32_t *__restrict__ vout, Params P)
{
    constexpr int TILE = THREADS * ITEMS;
    __shared__ uint32_t lstart[RADIX + 1];
    __shared__ uint32_t goff[RADIX];
    __shared__ uint32_t scan_lds[THREADS / 64 + 1];
    __shared__ int s_tile;
    const int tid = threadIdx.x;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID, bits [3:0]
    int local_t = 0;                      // mode 0: next tile of my chunk
    int cand = 0;                         // modes 2 / 5: index into my list of candidate segments
    for (;;) {
        // ---- which tile ----
        if (tid == 0) {
            int t = -1;
            if (P.mode == 0 || P.mode == 3) {
                if (local_t < P.tiles_per_wg) t = blockIdx.x * P.tiles_per_wg + local_t;
                if (t >= P.tiles) t = -1;
            } else if (P.mode == 1 || P.mode == 4) {
                t = (int)atomicAdd(&P.tickets[0], 1u);
                if (t >= P.tiles) t = -1;
            } else {
                // my XCD's segments first (xcc, xcc + 8, ...), then every segment in order (work stealing)
                const int own = ((int)xcc < P.nseg) ? (P.nseg - 1 - (int)xcc) / 8 + 1 : 0;
                while (cand < own + P.nseg) {
                    const int s = cand < own ? (int)xcc + 8 * cand : cand - own;
                    const int first = s * P.tiles_per_seg;
                    int cnt = P.tiles - first; if (cnt > P.tiles_per_seg) cnt = P.tiles_per_seg;
                    if (cnt > 0) {
                        const int k = (int)atomicAdd(&P.tickets[16 * (1 + s)], 1u);
                        if (k < cnt) { t = first + k; break; }
                    }
                    ++cand;
                }
            }
            s_tile = t;
        }
        __syncthreads();
        const int t = s_tile;
        if (t < 0) break;
        ++local_t;
        const int64_t base = (int64_t)t * TILE;
        // ---- keys (coalesced read) ----
        KeyT key[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int e = j * THREADS + tid;
            key[j] = base + e < P.n ? kin[base + e] : (KeyT)0;
        }
        // ---- tile-local digit starts ----
        uint32_t cnt = tid < RADIX ? P.counts[(int64_t)t * RADIX + tid] : 0u;
        {
            // exclusive scan over 256 threads' counts (all 1024 threads take part, the others add 0)
            uint32_t v = cnt;
            const int l = tid & 63, w = tid >> 6;
            uint32_t inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o, 64); if (l >= o) inc += u; }
            if (l == 63) scan_lds[w] = inc;
            __syncthreads();
            uint32_t woff = 0;
            for (int i = 0; i < w; ++i) woff += scan_lds[i];
            if (tid < RADIX) lstart[tid] = woff + inc - v;
            if (tid == RADIX) lstart[RADIX] = woff;           // == tile total
        }
        // ---- global offsets: table or look-back ----
        if (tid < RADIX) {
            uint32_t g;
            if (P.mode == 3) g = 0;
            else if (P.mode < 4) g = P.offs[(int64_t)t * RADIX + tid];
            else {
                const unsigned long long tagA = ((unsigned long long)(P.epoch * 4 + 1)) << 32, tagI = ((unsigned long long)(P.epoch * 4 + 2)) << 32;
                gu64 *st = (gu64 *)P.status;
                const int first = (t / P.tiles_per_seg) * P.tiles_per_seg;
                uint32_t prefix = P.segbase[(t / P.tiles_per_seg) * RADIX + tid];
                if (t > first) {
                    __hip_atomic_store(st + (int64_t)t * RADIX + tid, tagA | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    uint32_t acc = 0;
                    unsigned spins = 0;
                    for (int k = t - 1; k >= first; ) {
                        const unsigned long long x = __hip_atomic_load(st + (int64_t)k * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long tg = x & 0xffffffff00000000ull;
                        if (tg == tagI) { acc += (uint32_t)x; prefix = 0; break; }
                        if (tg == tagA) { acc += (uint32_t)x; --k; continue; }
                        if (++spins > (1u << 24)) { atomicAdd(P.err, 1u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    // (a walk that reaches `first` ends on that tile's INCLUSIVE tag: it carries the segment base / zero already)
                    g = acc + prefix;
                } else g = prefix;
                __hip_atomic_store(st + (int64_t)t * RADIX + tid, tagI | (unsigned long long)(g + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            goff[tid] = g - lstart[tid];
        }
        __syncthreads();
        // ---- scattered stores: element e of the (ranked) tile belongs to digit d(e): binary search in lstart ----
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int e = j * THREADS + tid;
            if (base + e < P.n) {
                uint32_t gp;
                if (P.mode == 3) gp = (uint32_t)(base + e);
                else {
                    int lo = 0, hi = RADIX;            // largest d with lstart[d] <= e
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (lstart[mid] <= (uint32_t)e) lo = mid; else hi = mid; }
                    gp = goff[lo] + (uint32_t)e;
                }
                kout[gp] = key[j];
                if (!P.novals) vout[gp] = (uint32_t)(base + e);
            }
        }
        __syncthreads();
    }
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    const int kbytes = argc > 2 ? atoi(argv[2]) : 8;
    const int items = argc > 3 ? atoi(argv[3]) : 8;
    const int skew = argc > 4 ? atoi(argv[4]) : 0;
    const int novals = argc > 5 ? atoi(argv[5]) : 0;      // 1: one stream only (packed (key, value) pairs as one 8-byte word)
    const int64_t n = (int64_t)1 << lg;
    const int tile = THREADS * items;
    const int tiles = (int)((n + tile - 1) / tile);
    printf("scatter probe: n = 2^%d, %d-byte keys%s, tile %d (%d tiles), %s digits\n", lg, kbytes, novals ? ", no values" : " + 4-byte values", tile, tiles, skew ? "skewed" : "uniform");
    void *kin, *kout; uint32_t *vout, *counts, *offs, *segbase, *tickets, *err; unsigned long long *status;
    CK(hipMalloc(&kin, (size_t)n * kbytes)); CK(hipMalloc(&kout, (size_t)n * kbytes)); CK(hipMalloc(&vout, (size_t)n * 4));
    CK(hipMalloc(&counts, (size_t)tiles * RADIX * 4)); CK(hipMalloc(&offs, (size_t)tiles * RADIX * 4));
    CK(hipMalloc(&segbase, (size_t)NSEG_MAX * RADIX * 4)); CK(hipMalloc(&tickets, 16 * (1 + NSEG_MAX) * 4)); CK(hipMalloc(&err, 4));
    CK(hipMalloc(&status, (size_t)tiles * RADIX * 8));
    CK(hipMemset(kin, 0x5a, (size_t)n * kbytes)); CK(hipMemset(err, 0, 4)); CK(hipMemset(status, 0, (size_t)tiles * RADIX * 8));
    hipLaunchKernelGGL(k_counts, dim3(tiles), dim3(THREADS), 0, 0, counts, n, tile, skew);
    std::vector<uint32_t> hc((size_t)tiles * RADIX), ho((size_t)tiles * RADIX);
    CK(hipMemcpy(hc.data(), counts, hc.size() * 4, hipMemcpyDeviceToHost));
    {
        std::vector<uint64_t> tot(RADIX, 0);
        for (int t = 0; t < tiles; ++t) for (int d = 0; d < RADIX; ++d) tot[d] += hc[(size_t)t * RADIX + d];
        std::vector<uint64_t> run(RADIX, 0);
        uint64_t s = 0;
        for (int d = 0; d < RADIX; ++d) { run[d] = s; s += tot[d]; }
        for (int t = 0; t < tiles; ++t) for (int d = 0; d < RADIX; ++d) { ho[(size_t)t * RADIX + d] = (uint32_t)run[d]; run[d] += hc[(size_t)t * RADIX + d]; }
    }
    CK(hipMemcpy(offs, ho.data(), ho.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int epoch = 0;
    const double bytes = (double)n * (2.0 * kbytes + (novals ? 0.0 : 4.0));
    struct Cfg { int mode, wgs, nseg; const char *name; };
    const Cfg cfgs[] = {
        { 3, 256, 0, "sequential stores (ceiling)            " },
        { 0, 256, 0, "chunked, 256 WGs, no carries           " },
        { 1, 256, 0, "global tickets, table offsets          " },
        { 2, 256, 8, "xcd tickets (8 segments), table offsets " },
        { 2, 256, 32, "xcd tickets (32 segments), table offsets" },
        { 4, 256, 0, "global tickets + look-back             " },
        { 5, 256, 8, "xcd tickets (8 segments) + look-back   " },
        { 5, 256, 32, "xcd tickets (32 segments) + look-back  " },
    };
    for (const Cfg &c : cfgs) {
        Params P;
        P.counts = counts; P.offs = offs; P.segbase = segbase; P.status = status; P.tickets = tickets; P.err = err;
        P.n = n; P.tile = tile; P.tiles = tiles; P.mode = c.mode; P.nseg = c.nseg ? c.nseg : 1; P.novals = novals;
        P.tiles_per_seg = (tiles + P.nseg - 1) / P.nseg;
        P.tiles_per_wg = (tiles + c.wgs - 1) / c.wgs;
        if (c.mode >= 4) {
            std::vector<uint32_t> sb((size_t)P.nseg * RADIX, 0);
            for (int s = 0; s < P.nseg; ++s) {
                const int first = s * P.tiles_per_seg;
                if (first < tiles) for (int d = 0; d < RADIX; ++d) sb[(size_t)s * RADIX + d] = ho[(size_t)first * RADIX + d];
            }
            CK(hipMemcpy(segbase, sb.data(), sb.size() * 4, hipMemcpyHostToDevice));
        }
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemsetAsync(tickets, 0, 16 * (1 + NSEG_MAX) * 4, 0));
            P.epoch = ++epoch;
            CK(hipEventRecord(e0, 0));
            if (kbytes == 8 && items == 8) hipLaunchKernelGGL((k_scatter<uint64_t, 8>), dim3(c.wgs), dim3(THREADS), 0, 0, (const uint64_t *)kin, (uint64_t *)kout, vout, P);
            else if (kbytes == 4 && items == 8) hipLaunchKernelGGL((k_scatter<uint32_t, 8>), dim3(c.wgs), dim3(THREADS), 0, 0, (const uint32_t *)kin, (uint32_t *)kout, vout, P);
            else if (kbytes == 4 && items == 12) hipLaunchKernelGGL((k_scatter<uint32_t, 12>), dim3(c.wgs), dim3(THREADS), 0, 0, (const uint32_t *)kin, (uint32_t *)kout, vout, P);
            else if (kbytes == 4 && items == 16) hipLaunchKernelGGL((k_scatter<uint32_t, 16>), dim3(c.wgs), dim3(THREADS), 0, 0, (const uint32_t *)kin, (uint32_t *)kout, vout, P);
            else if (kbytes == 8 && items == 12) hipLaunchKernelGGL((k_scatter<uint64_t, 12>), dim3(c.wgs), dim3(THREADS), 0, 0, (const uint64_t *)kin, (uint64_t *)kout, vout, P);
            else if (kbytes == 8 && items == 4) hipLaunchKernelGGL((k_scatter<uint64_t, 4>), dim3(c.wgs), dim3(THREADS), 0, 0, (const uint64_t *)kin, (uint64_t *)kout, vout, P);
            else { fprintf(stderr, "unsupported key bytes / items\n"); return 2; }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        uint32_t herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        // spot check: the values of the first and last digit runs must be a permutation slice (sum of vout over the run of digit 0)
        printf("  %s  %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)%s\n", c.name, best, bytes / best / 1e6, bytes / best / 1e6 / 8000.0, herr ? "  LOOK-BACK TIMEOUT" : "");
        fflush(stdout);
    }
    // correctness of the look-back offsets: the last mode's output must equal the table-offset output
    {
        std::vector<uint32_t> a((size_t)1 << 20), b((size_t)1 << 20);
        CK(hipMemcpy(a.data(), vout, a.size() * 4, hipMemcpyDeviceToHost));
        Params P; P.counts = counts; P.offs = offs; P.segbase = segbase; P.status = status; P.tickets = tickets; P.err = err;
        P.n = n; P.tile = tile; P.tiles = tiles; P.mode = 1; P.nseg = 1; P.novals = 0; P.tiles_per_seg = tiles; P.tiles_per_wg = (tiles + 255) / 256; P.epoch = ++epoch;
        CK(hipMemsetAsync(tickets, 0, 16 * (1 + NSEG_MAX) * 4, 0));
        CK(hipMemsetAsync(vout, 0, (size_t)n * 4, 0));
        if (kbytes == 8 && items == 8) hipLaunchKernelGGL((k_scatter<uint64_t, 8>), dim3(256), dim3(THREADS), 0, 0, (const uint64_t *)kin, (uint64_t *)kout, vout, P);
        else if (kbytes == 4 && items == 8) hipLaunchKernelGGL((k_scatter<uint32_t, 8>), dim3(256), dim3(THREADS), 0, 0, (const uint32_t *)kin, (uint32_t *)kout, vout, P);
        else if (kbytes == 4 && items == 12) hipLaunchKernelGGL((k_scatter<uint32_t, 12>), dim3(256), dim3(THREADS), 0, 0, (const uint32_t *)kin, (uint32_t *)kout, vout, P);
        else if (kbytes == 4 && items == 16) hipLaunchKernelGGL((k_scatter<uint32_t, 16>), dim3(256), dim3(THREADS), 0, 0, (const uint32_t *)kin, (uint32_t *)kout, vout, P);
        else if (kbytes == 8 && items == 12) hipLaunchKernelGGL((k_scatter<uint64_t, 12>), dim3(256), dim3(THREADS), 0, 0, (const uint64_t *)kin, (uint64_t *)kout, vout, P);
        else hipLaunchKernelGGL((k_scatter<uint64_t, 4>), dim3(256), dim3(THREADS), 0, 0, (const uint64_t *)kin, (uint64_t *)kout, vout, P);
        CK(hipMemcpy(b.data(), vout, b.size() * 4, hipMemcpyDeviceToHost));
        printf("look-back offsets %s the table offsets (first 2^20 outputs)\n", a == b ? "reproduce" : "DIFFER FROM");
    }
    return 0;
}
