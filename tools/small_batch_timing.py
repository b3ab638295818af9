"""Many small texts through ONE sa_amd_saca_batch call (k_small_sa_batch: one workgroup per text, one launch per chunk)
against one SuffixArray::new-equivalent call per text (reference src/sa.rs:23-27; src/tests.rs:13-17 is the size domain).
python tools/small_batch_timing.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus

rng = np.random.default_rng(1)
sa.saca_batch([rng.integers(0, 256, 100, dtype=np.uint8) for _ in range(4)])      # context, code objects, pool
for kind in ("uniform", "english"):
    for n, count in ((64, 16384), (256, 16384), (1024, 8192), (4096, 4096), (8192, 2048)):
        texts = [rng.integers(0, 256, n, dtype=np.uint8) if kind == "uniform" else corpus.english(n, int(rng.integers(0, 1 << 30))) for _ in range(count)]
        sa.saca_batch(texts[:8])
        outs = sa.saca_batch(texts)
        # the C call alone (the Python wrapper spends ~1.5 us per text on ctypes pointers and output arrays)
        cnt = len(texts)
        T = (ctypes.c_void_p * cnt)(*[t.ctypes.data for t in texts])
        S = (ctypes.c_void_p * cnt)(*[o.ctypes.data for o in outs])
        N = (ctypes.c_int32 * cnt)(*[t.size for t in texts])
        stt = (ctypes.c_int32 * cnt)()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); rc = sa.lib().sa_amd_saca_batch(T, S, N, None, cnt, stt); best = min(best, time.perf_counter() - t0)
            assert rc == 0
        py = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); outs2 = sa.saca_batch(texts); py = min(py, time.perf_counter() - t0)
        assert all(np.array_equal(a, b) for a, b in zip(outs[::97], outs2[::97]))
        out = np.zeros(n + 1, dtype=np.uint32)
        k = min(count, 512)
        t0 = time.perf_counter()
        for t in texts[:k]: sa.saca(t, out)
        single = (time.perf_counter() - t0) / k
        ok = np.array_equal(outs[k - 1], out)
        print(f"{kind:8s} n={n:5d} x {count:6d}: batch {best*1e3:8.2f} ms = {best/count*1e6:6.2f} us per text, {count*n/best/1e6:8.1f} MB/s of text | "
              f"one call per text {single*1e6:6.1f} us -> {single/(best/count):5.1f}x  (same array: {ok}) | python saca_batch {py*1e3:7.2f} ms", flush=True)
