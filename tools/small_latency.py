"""Per-call latency of the host-pointer entry point on small texts (the reference's test domain)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
for n in (16, 1024, 4096, 65536, 1 << 20):
    t = corpus.uniform(n, 1); out = np.zeros(n + 1, dtype=np.uint32)
    sa.saca(t, out)
    t0 = time.perf_counter(); reps = 50
    for _ in range(reps): sa.saca(t, out)
    print(f"n={n:8d}: {(time.perf_counter()-t0)/reps*1e3:8.3f} ms per SuffixArray::new-equivalent call")
