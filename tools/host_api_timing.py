"""End-to-end timing of the host-pointer entry point (sa_amd_saca_u8 = `saca()`, reference src/saca.rs:9-15) with the
phase breakdown of sa_amd_last_host_timing.  Two callers' habits are timed: a REUSED output buffer (SuffixArray::set,
src/sa.rs:30-33) and a FRESH zeroed one per call (SuffixArray::new, src/sa.rs:23-27: vec![0; n + 1], first touched by
the download).  python tools/host_api_timing.py [workload ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus

names = sys.argv[1:] or ["c2_uniform_64m", "c3_english_256m"]
for name in names:
    t = corpus.workload(name)
    out = np.zeros(t.size + 1, dtype=np.uint32)
    sa.saca(t, out)          # warm-up (context, code objects, pool)
    for threads in (os.environ.get("SA_HOST_TIMING_THREADS", "0,2,4,8").split(",")):
        os.environ["SA_AMD_COPY_THREADS"] = threads
        for mode in ("reused", "fresh"):
            best, bt = 1e9, None
            for _ in range(4):
                if mode == "fresh":
                    out = np.zeros(t.size + 1, dtype=np.uint32)      # calloc: untouched pages, like vec![0; n + 1]
                t0 = time.perf_counter(); sa.saca(t, out); dt = time.perf_counter() - t0
                if dt < best:
                    best, bt = dt, sa.last_host_timing()
            print(f"{name} copy_threads={threads} {mode:6s}: {best*1e3:7.1f} ms = {t.size/1e6/best:6.0f} MB/s | "
                  + " ".join(f"{k} {v:.1f}" for k, v in bt.items()), flush=True)
