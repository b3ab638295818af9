"""A few builds of one generated text through the host-pointer entry point, with the per-round trace of the last one on
stderr: python tools/mid_build.py <generator> <bytes> [builds]   (the thing to put under rocprofv3 --kernel-trace for a
dispatch sequence of a mid-size text: tools/trace_sequence.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
gen, n = sys.argv[1], int(sys.argv[2])
k = int(sys.argv[3]) if len(sys.argv) > 3 else 4
t = getattr(corpus, gen)(n, 3)
out = np.zeros(n + 1, dtype=np.uint32)
best = 1e9
for i in range(k):
    if i == k - 1:
        os.environ["SA_AMD_VERBOSE"] = "3"
    t0 = time.perf_counter(); sa.saca(t, out); dt = time.perf_counter() - t0
    best = min(best, dt)
print(f"{gen} {n}: best {best*1e3:.3f} ms, last {dt*1e3:.3f} ms, phases {sa.last_host_timing()}, stats {sa.last_stats()}", file=sys.stderr)
