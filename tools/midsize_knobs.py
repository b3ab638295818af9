"""Build time of mid-size texts under a few environment settings: python tools/midsize_knobs.py "K=V,K=V" ...  ('-' = default)"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
sizes = [int(x) for x in os.environ["MIDSIZE_SIZES"].split(",")] if os.environ.get("MIDSIZE_SIZES") else [1 << 17, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 24]
for gen in ("uniform", "dna", "english_corpus", "english"):
    for n in sizes:
        t = getattr(corpus, gen)(n, 3)
        out = np.zeros(n + 1, dtype=np.uint32)
        row = []
        for spec in sys.argv[1:]:
            saved = dict(os.environ)
            if spec != "-":
                for kv in spec.split(","):
                    k, v = kv.split("="); os.environ[k] = v
            sa.saca(t, out)
            best = 1e9
            for _ in range(6):
                sa.saca(t, out); best = min(best, sa.last_host_timing()["build"])
            row.append(f"{best:7.3f} (p{sa.last_stats()['sort_passes']})")
            os.environ.clear(); os.environ.update(saved)
        print(f"{gen:15s} {n:9d}  build ms: " + "  ".join(row), flush=True)
