"""Per-call latency of the host-pointer entry point on small texts (the reference's test domain, src/tests.rs:14: n < 4096):
the one-launch small kernel (kernels/small.hpp, n <= 8192) against the general pipeline (SA_AMD_SMALL_MAX=0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
for env in (None, "0"):
    if env is None:
        os.environ.pop("SA_AMD_SMALL_MAX", None)
    else:
        os.environ["SA_AMD_SMALL_MAX"] = env
    print("general pipeline (SA_AMD_SMALL_MAX=0)" if env == "0" else "default (one-launch kernel up to 8192 bytes)")
    for name, gen in (("uniform bytes", lambda n: corpus.uniform(n, 1)), ("english", lambda n: corpus.english(n, 1)),
                      ("one byte value", lambda n: np.full(n, 65, dtype=np.uint8))):
        for n in (16, 256, 1024, 4095, 4096, 8192, 65536):
            if name != "uniform bytes" and n > 8192:
                continue
            t = gen(n); out = np.zeros(n + 1, dtype=np.uint32)
            for _ in range(3): sa.saca(t, out)
            reps = 200 if n <= 8192 else 50
            t0 = time.perf_counter()
            for _ in range(reps): sa.saca(t, out)
            print(f"  {name:15s} n={n:6d}: {(time.perf_counter()-t0)/reps*1e6:9.1f} us per SuffixArray::new-equivalent call")
