"""LCP / entropy statistics of a benchmark workload, computed with the oracle (SA-IS + Kasai) on the CPU:
python tools/corpus_stats.py c3_english_256m [n]  -> one JSON object (committed under profiles/r02_corpus_stats.json
and quoted by bench.py in `config`).  VERDICT r1 item 3: the headline text must state its mean / max LCP and H0."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from suffix_array_amd import corpus
from conftest import Oracle


def stats(name, n=None):
    text = corpus.workload(name, n_override=n)
    orc = Oracle()
    t0 = time.time()
    arr = orc.sais(text)
    t1 = time.time()
    out = (ctypes.c_uint64 * 32)()
    orc.L.oracle_lcp_stats.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    assert orc.L.oracle_lcp_stats(text.ctypes.data, text.size, arr.ctypes.data, out) == 0
    c = np.bincount(text, minlength=256)
    p = c[c > 0] / text.size
    return {"workload": name, "n_bytes": int(text.size), "seed": corpus.WORKLOADS[name][2], "sigma": int((c > 0).sum()),
            "H0_bits_per_byte": round(float(-(p * np.log2(p)).sum()), 4),
            "mean_lcp": round(out[0] / text.size, 2), "max_lcp": int(out[1]),
            "frac_lcp_ge": {str(1 << k): round(out[2 + k] / text.size, 6) for k in range(0, 24) if out[2 + k]},
            "oracle_sais_seconds": round(t1 - t0, 1), "kasai_seconds": round(time.time() - t1, 1)}


if __name__ == "__main__":
    print(json.dumps(stats(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)))
