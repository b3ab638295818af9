"""One build per workload with SA_AMD_VERBOSE=3: the library prints one line per refinement round (tied suffixes in and
out) on stderr.  python tools/round_trace.py [workload ...]"""
import os, sys, time
os.environ["SA_AMD_VERBOSE"] = "3"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
for name in sys.argv[1:] or ["c3_english_256m"]:
    t = corpus.workload(name)
    out = np.zeros(t.size + 1, dtype=np.uint32)
    os.environ["SA_AMD_VERBOSE"] = "0"
    sa.saca(t, out)
    os.environ["SA_AMD_VERBOSE"] = "3"
    print(f"== {name}", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); sa.saca(t, out); dt = time.perf_counter() - t0
    print(f"== {name}: {dt*1e3:.1f} ms end to end, stats {sa.last_stats()}", file=sys.stderr, flush=True)
