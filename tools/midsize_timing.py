import sys, time, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT","."))
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
for gen in ("english_corpus", "uniform", "dna"):
    for mb in (1, 4, 16, 64):
        n = mb << 20
        t = getattr(corpus, gen)(n, 3)
        out = np.zeros(n + 1, dtype=np.uint32)
        sa.saca(t, out)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); sa.saca(t, out); best = min(best, time.perf_counter() - t0)
        ht = sa.last_host_timing(); st = sa.last_stats()
        print(f"{gen:15s} {mb:3d} MiB  e2e {best*1e3:7.2f} ms  build {ht['build']:6.2f}  h2d {ht['h2d']:.2f} d2h {ht['d2h']:.2f}  rounds {st['rounds']} passes {st['sort_passes']}", flush=True)
