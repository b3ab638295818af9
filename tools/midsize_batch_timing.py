"""Many MID-SIZE texts through one sa_amd_saca_batch call: host threads per device (SA_AMD_BATCH_THREADS) and the per-device
phase lanes (SA_AMD_NO_LANES) varied per call.  A mid-size build is bound by launches and read-backs, not by the GPU, so
several of them in flight on one device overlap.  python tools/midsize_batch_timing.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus

def run(texts, outs):
    cnt = len(texts)
    T = (ctypes.c_void_p * cnt)(*[t.ctypes.data for t in texts])
    S = (ctypes.c_void_p * cnt)(*[o.ctypes.data for o in outs])
    N = (ctypes.c_int32 * cnt)(*[t.size for t in texts])
    st = (ctypes.c_int32 * cnt)()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); rc = sa.lib().sa_amd_saca_batch(T, S, N, None, cnt, st); best = min(best, time.perf_counter() - t0)
        assert rc == 0
    return best

for gen, mb, count in (("uniform", 1, 128), ("english_corpus", 1, 128), ("uniform", 4, 64), ("english_corpus", 4, 64), ("uniform", 16, 32), ("english_corpus", 16, 16), ("uniform", 64, 8)):
    n = mb << 20
    texts = [getattr(corpus, gen)(n, 100 + i) for i in range(count)]
    outs = [np.zeros(n + 1, dtype=np.uint32) for _ in texts]
    ref = None
    for env in ({}, {"SA_AMD_BATCH_THREADS": "2", "SA_AMD_LANES_MIN_N": "0"}, {"SA_AMD_BATCH_THREADS": "2"}, {"SA_AMD_BATCH_THREADS": "4"}, {"SA_AMD_BATCH_THREADS": "8"}, {"SA_AMD_BATCH_THREADS": "12"}, {}):
        for k in ("SA_AMD_BATCH_THREADS", "SA_AMD_NO_LANES", "SA_AMD_LANES_MIN_N"): os.environ.pop(k, None)
        os.environ.update(env)
        dt = run(texts, outs)
        if ref is None: ref = [o.copy() for o in outs[:2]]
        same = all(np.array_equal(a, b) for a, b in zip(ref, outs[:2]))
        print(f"{gen:15s} {mb:3d} MiB x {count:4d} {str(env):62s}: {dt*1e3:8.1f} ms = {dt/count*1e3:6.2f} ms per text, {count*n/dt/1e6:8.0f} MB/s  same {same}", flush=True)
