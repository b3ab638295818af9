"""Host-pointer calls into a FRESH output buffer (np.zeros per call: pages mapped on first write) under a few settings."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
t = corpus.uniform(mb << 20, 3)
out = np.zeros(t.size + 1, dtype=np.uint32); sa.saca(t, out)
for spec in sys.argv[2:]:
    saved = dict(os.environ)
    if spec != "-":
        for kv in spec.split(","):
            k, v = kv.split("="); os.environ[k] = v
    rows = []
    for _ in range(7):
        o = np.zeros(t.size + 1, dtype=np.uint32)
        sa.saca(t, o); rows.append(sa.last_host_timing()); del o
    rows.sort(key=lambda r: r["total"]); m = rows[len(rows) // 2]
    sa.saca(t, out); r = sa.last_host_timing()
    print(f"{spec:45s} fresh (median of 7): build {m['build']:.2f} d2h {m['d2h']:.2f} total {m['total']:.2f} | reused: build {r['build']:.2f} d2h {r['d2h']:.2f} total {r['total']:.2f}", flush=True)
    os.environ.clear(); os.environ.update(saved)
