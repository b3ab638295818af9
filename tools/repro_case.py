"""Repeat one build several times and compare every array with the oracle (debug aid).
python tools/repro_case.py gen n seed [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
from conftest import Oracle
gen, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
t = getattr(corpus, gen)(n, seed)
exp = Oracle().sais(t)
warm = corpus.uniform(n - 1, 7)
for r in range(reps):
    if r % 2 == 1:
        sa.SuffixArray(warm)                      # another size in between: the pooled block holds other data
    got = sa.SuffixArray(t).into_parts()[1]
    bad = np.nonzero(got != exp)[0]
    print(r, "ok" if bad.size == 0 else f"MISMATCH at {bad.size} slots, first {bad[:8]} got {got[bad[:4]]} exp {exp[bad[:4]]}", sa.last_stats(), flush=True)
