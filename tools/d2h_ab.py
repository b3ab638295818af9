"""End-to-end A/B of the download path in the two states of the copy engine: python tools/d2h_ab.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
def run(tag, t, reps=5):
    out = np.zeros(t.size + 1, dtype=np.uint32); sa.saca(t, out)
    best = None
    for _ in range(reps):
        sa.saca(t, out); ht = sa.last_host_timing()
        if best is None or ht["total"] < best["total"]: best = ht
    print(f"{tag:60s}", {k: round(v, 2) for k, v in best.items() if k in ("h2d", "build", "d2h", "total", "early_fraction")}, flush=True)
u = corpus.uniform(64 << 20, 3); c3 = corpus.workload("c3_english_256m"); small = corpus.uniform(1 << 20, 5)
for env in ({}, {"SA_AMD_NO_KERNEL_D2H": "1"}):
    os.environ.pop("SA_AMD_NO_KERNEL_D2H", None); os.environ.pop("SA_AMD_KERNEL_D2H_ALWAYS", None); os.environ.update(env)
    run(f"64 MiB random, first blocks of the process {env}", u)
    run(f"C3 {env}", c3)
import time
os.environ["SA_AMD_CACHE_IDLE_MS"] = "1"   # the pool lets its idle blocks go at the next call: what follows is allocated after a free
time.sleep(0.05); run("1 MiB random", small, 1); time.sleep(0.05); run("1 MiB random", small, 1)
os.environ.pop("SA_AMD_CACHE_IDLE_MS")
os.environ["SA_AMD_VERBOSE"] = "3"
for env in ({}, {"SA_AMD_NO_KERNEL_D2H": "1"}, {"SA_AMD_KERNEL_D2H_ALWAYS": "1"}, {}):
    os.environ.pop("SA_AMD_NO_KERNEL_D2H", None); os.environ.pop("SA_AMD_KERNEL_D2H_ALWAYS", None); os.environ.update(env)
    run(f"64 MiB random, after the pool let its blocks go {env}", u)
    run(f"C3, after the pool let its blocks go {env}", c3)
