"""Build times on adversarial inputs (SURVEY.md section 7.3 families) at a given size; every result verified.
python tools/adversarial_timing.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
from conftest import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else (64 << 20)
orc = Oracle()
rng = np.random.default_rng(7)

def fib(n):
    a, b = b"a", b"ab"
    while len(b) < n: a, b = b, b + a
    return np.frombuffer(b[:n], dtype=np.uint8).copy()

cases = {
    "all one byte": np.full(n, 97, dtype=np.uint8),
    "period 2 (abab...)": np.resize(np.frombuffer(b"ab", dtype=np.uint8), n).copy(),
    "period 1000 random block": np.resize(rng.integers(0, 256, 1000, dtype=np.uint8), n).copy(),
    "Fibonacci word": fib(n),
    "text ++ text (one long repeat)": np.concatenate([corpus.english(n // 2, 5)] * 2),
    "DNA with planted repeats": corpus.dna_repeats(n, 9, 0.4),
    "english, iid words (round-1 C3 model)": corpus.english(n, 3),
    "english corpus (C3)": corpus.english_corpus(n, 3),
}
warm = corpus.uniform(n, 1)                      # context, code objects and the pooled device block are paid here, not by the first case
sa.saca(warm, np.zeros(n + 1, dtype=np.uint32))
del warm
for name, t in cases.items():
    t = np.ascontiguousarray(t)
    out = np.zeros(t.size + 1, dtype=np.uint32)
    t0 = time.perf_counter(); sa.saca(t, out); dt = time.perf_counter() - t0
    st = sa.last_stats()
    ok = orc.verify(t, out) if hasattr(orc, "verify") else None
    print(f"{name:38s} n={t.size:>10d}  {dt*1e3:9.1f} ms end-to-end (host pointers)  rounds {st['rounds']:2d} text {st['text_rounds']} passes {st['sort_passes']:4d} "
          f"sparse {st['sparse_mode']}  verified {ok}", flush=True)
