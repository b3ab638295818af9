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
    if name.startswith("adv:"):                 # adversarial families of tools/adversarial_timing.py: adv:<family>:<n>
        _, fam, nn = name.split(":"); nn = int(nn)
        t = {"one": lambda: np.full(nn, 97, dtype=np.uint8),
             "ab": lambda: np.resize(np.frombuffer(b"ab", dtype=np.uint8), nn).copy(),
             "fib": lambda: (lambda f: np.frombuffer(f(nn)[:nn], dtype=np.uint8).copy())(lambda k: (lambda ab: ab)(__import__("functools").reduce(lambda p, _: (p[1], p[1] + p[0]) if len(p[1]) < k else p, range(64), (b"a", b"ab"))[1])),
             "twice": lambda: np.concatenate([corpus.english(nn // 2, 5)] * 2),
             "p1000": lambda: np.resize(np.random.default_rng(7).integers(0, 256, 1000, dtype=np.uint8), nn).copy()}[fam]()
    else:
        t = corpus.workload(name)
    out = np.zeros(t.size + 1, dtype=np.uint32)
    os.environ["SA_AMD_VERBOSE"] = "0"
    sa.saca(t, out)
    os.environ["SA_AMD_VERBOSE"] = "3"
    print(f"== {name}", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); sa.saca(t, out); dt = time.perf_counter() - t0
    print(f"== {name}: {dt*1e3:.1f} ms end to end, stats {sa.last_stats()}", file=sys.stderr, flush=True)
