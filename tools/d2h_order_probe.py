"""Does the D2H phase of a 64 MiB text depend on what the process built before?  (pool state: pinned stage blocks, helpers)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
def run(gen, mb, reps=4):
    t = getattr(corpus, gen)(mb << 20, 3); out = np.zeros(t.size + 1, dtype=np.uint32)
    sa.saca(t, out)
    best = None
    for _ in range(reps):
        sa.saca(t, out); ht = sa.last_host_timing()
        if best is None or ht["total"] < best["total"]: best = ht
    import time
    src = np.ones(out.size, dtype=np.uint32)
    np.copyto(out, src); t0 = time.perf_counter(); np.copyto(out, src); dt = time.perf_counter() - t0
    numa = ""
    try:
        pid = os.getpid(); addr = out.ctypes.data
        for line in open(f"/proc/{pid}/numa_maps"):
            a = int(line.split()[0], 16)
            if a <= addr < a + (1 << 34) and "anon" in line and abs(a - addr) < (1 << 21): numa = " ".join(x for x in line.split()[1:] if x.startswith(("N", "anon", "kernelpagesize", "huge", "bind", "default", "prefer")))
    except Exception as e:
        numa = str(e)
    print(f"{gen:15s} {mb:3d} MiB", {k: round(v, 2) for k, v in best.items() if k in ("h2d", "build", "d2h", "total", "staged_threads")}, f"| 1-thread copy into out {out.nbytes / dt / 1e9:.1f} GB/s | {numa}", flush=True)
order = sys.argv[1:] or ["uniform:64", "uniform:1", "uniform:64", "uniform:4", "uniform:64", "uniform:16", "uniform:64", "english_corpus:64", "uniform:64"]
for o in order:
    g, mb = o.split(":"); run(g, int(mb))
