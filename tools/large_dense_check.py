"""One-off validation of the rank-doubling route at the top of the size range (not in the test suite: minutes of host time):
a 1.5 GiB English-like corpus, and repetitive texts of MAX_LENGTH = 2^31 - 1 bytes (one byte repeated; `abc` repeated) --
32-bit rank keys, 2^15-entry ISA windows, index arithmetic next to 2^32.  Each array is checked with the device integrity
check (reference src/sa.rs:72-84 in linear time).  python tools/large_dense_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus

def run(name, t):
    out = np.empty(t.size + 1, dtype=np.uint32)
    t0 = time.perf_counter(); sa.saca(t, out); dt = time.perf_counter() - t0
    ok = sa.check_integrity(t, out)
    st = sa.last_stats()
    print(f"{name}: n={t.size} {dt*1e3:.0f} ms end to end, rounds {st['rounds']}, passes {st['sort_passes']}, verified {ok}", flush=True)
    sa.lib().sa_amd_release_cache()
    return ok

ok = True
ok &= run("english corpus 1.5 GiB", corpus.english_corpus(3 << 29, 21))
n = sa.MAX_LENGTH
ok &= run("one byte x MAX_LENGTH", np.full(n, 120, dtype=np.uint8))
ok &= run("abc x MAX_LENGTH", np.resize(np.frombuffer(b"abc", dtype=np.uint8), n).copy())
sys.exit(0 if ok else 1)
