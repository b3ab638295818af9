"""Gram keys below the default size threshold: device-resident build times of an English-like corpus at several sizes with
SA_AMD_GRAM_MIN_N at its default and at 1.  python tools/gram_sizes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import suffix_array_amd as sa
from suffix_array_amd import corpus
dev = torch.device("cuda", 0)
for n in (1 << 20, 4 << 20, 16 << 20, 64 << 20):
    t = corpus.english_corpus(n, 3)
    text = torch.from_numpy(t).to(dev)
    out = torch.empty(n + 1, dtype=torch.int32, device=dev)
    wb = sa.workspace_bytes(n)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    st = sa.Stats()
    for env in ({}, {"SA_AMD_GRAM_MIN_N": "1"}, {"SA_AMD_NO_GRAM_KEYS": "1"}):
        os.environ.update(env)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sa.saca_device_ptr(text.data_ptr(), out.data_ptr(), n, work.data_ptr(), wb, 0, st)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"n={n:>10d} {str(env):36s} {best*1e3:8.3f} ms  symbols/key {st.symbols_per_key} rounds {st.rounds} passes {st.sort_passes}", flush=True)
        for k in env: os.environ.pop(k)
