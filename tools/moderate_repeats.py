"""Corpora with few copied passages (dup_fraction 0.2 % .. 3 %): which route the repeat probe picks and what a build costs,
with the per-round trace (SA_AMD_VERBOSE=3).  python tools/moderate_repeats.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SA_AMD_VERBOSE"] = "3"
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
n = 256 << 20
for frac in (0.002, 0.01, 0.03):
    t = corpus.english_corpus(n, 11, 50000, frac)
    # second printings are part of the generator: cut them out by regenerating the two regions? keep: they are ~1 % of the text
    out = np.zeros(n + 1, dtype=np.uint32)
    os.environ["SA_AMD_VERBOSE"] = "0"; sa.saca(t, out); os.environ["SA_AMD_VERBOSE"] = "3"
    print(f"== dup_fraction {frac}", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); sa.saca(t, out); dt = time.perf_counter() - t0
    print(f"== {frac}: {dt*1e3:.1f} ms e2e, ok {sa.check_integrity(t, out)} {sa.last_stats()}", file=sys.stderr, flush=True)
