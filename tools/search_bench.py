"""Row (f) of SURVEY.md section 8, measured: batched contains / search_all / search_lcp (reference src/sa.rs:164-253) on a
device-resident index of the headline text.  python tools/search_bench.py [patterns] -> queries per second of one
sa_amd_index_search call (host patterns in, host results out), with and without the bucket table; every answer of a
sample is checked against a plain binary search over the suffix array on the CPU."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus

cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
t = corpus.workload("c3_english_256m")
n = t.size
rng = np.random.default_rng(11)
t0 = time.perf_counter(); ix = sa.DeviceIndex(t); dt = time.perf_counter() - t0
print(f"index of {n} bytes built and kept in HBM in {dt*1e3:.0f} ms (host text in)", flush=True)
L = sa.lib()
for plen in (8, 32):
    pos = rng.integers(0, n - plen, cnt)
    data = np.empty(cnt * plen, dtype=np.uint8)
    for j in range(plen):
        data[j::plen] = t[pos + j]
    miss = rng.random(cnt) < 0.5                                   # half of the patterns get one byte changed: mostly misses
    data[np.flatnonzero(miss) * plen + plen // 2] ^= 0x5a
    off = (np.arange(cnt + 1, dtype=np.int64) * plen)
    c = np.zeros(cnt, dtype=np.uint8); lo, hi, ls, ll = (np.zeros(cnt, dtype=np.uint32) for _ in range(4))
    for label in ("no bucket table", "bucket table"):
        if label == "bucket table":
            ix.buckets()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            rc = L.sa_amd_index_search(ix._h, data.ctypes.data, off.ctypes.data, cnt, c.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                                       ls.ctypes.data, ll.ctypes.data)
            best = min(best, time.perf_counter() - t0)
            assert rc == 0
        print(f"patterns of {plen:2d} bytes, {label:15s}: {cnt} queries in {best*1e3:7.2f} ms = {cnt/best/1e6:6.2f} M queries/s "
              f"(contains: {int(c.sum())} hits)", flush=True)
    # check a sample against the suffix array on the CPU
    arr = ix.suffix_array()
    tb = t.tobytes()
    for q in rng.integers(0, cnt, 200):
        p = data[q * plen:(q + 1) * plen].tobytes()
        a, b = 1, n + 1                                            # lower bound of p among the suffixes sa[1..n]
        while a < b:
            m = (a + b) // 2
            if tb[arr[m]:arr[m] + plen] < p: a = m + 1
            else: b = m
        first = a
        b = n + 1
        while a < b:
            m = (a + b) // 2
            if tb[arr[m]:arr[m] + plen] <= p: a = m + 1
            else: b = m
        assert bool(c[q]) == (a > first) and (not c[q] or (lo[q] == first and hi[q] == a)), (q, first, a, lo[q], hi[q], c[q])
    print(f"  200 sampled answers agree with a binary search over the array on the CPU", flush=True)
