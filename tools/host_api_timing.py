"""End-to-end timing of the host-pointer entry point (sa_amd_saca_u8: hipMalloc + H2D + build + D2H)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
for name, n in (("c2_uniform_64m", None), ("c3_english_256m", None)):
    t = corpus.workload(name)
    out = np.zeros(t.size + 1, dtype=np.uint32)
    sa.saca(t, out)          # warm-up (context, code objects)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); sa.saca(t, out); best = min(best, time.perf_counter() - t0)
    print(f"{name}: host-pointer end-to-end {best*1e3:.1f} ms = {t.size/1e6/best:.0f} MB/s (PCIe + allocation inclusive)")
