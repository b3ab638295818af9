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

# batch entry point on one GPU: items alternate between host threads per device (SA_AMD_BATCH_THREADS);
# output arrays allocated and touched beforehand (first-touch page faults are the caller's, not the library's)
import ctypes
texts = [corpus.workload("c2_uniform_64m", rank=r) for r in range(6)]
outs = [np.zeros(t.size + 1, dtype=np.uint32) for t in texts]
cnt = len(texts)
T = (ctypes.c_void_p * cnt)(*[t.ctypes.data for t in texts])
S = (ctypes.c_void_p * cnt)(*[o.ctypes.data for o in outs])
N = (ctypes.c_int32 * cnt)(*[t.size for t in texts])
stt = (ctypes.c_int32 * cnt)()
L = sa.lib()
for threads in ("1", "2", "3", "1", "2"):
    os.environ["SA_AMD_BATCH_THREADS"] = threads
    t0 = time.perf_counter(); rc = L.sa_amd_saca_batch(T, S, N, None, cnt, stt); dt = time.perf_counter() - t0
    assert rc == 0
    print(f"batch of {cnt} x 64 MiB, {threads} host thread(s) per device: {dt*1e3:.1f} ms = {sum(t.size for t in texts)/1e6/dt:.0f} MB/s end to end")
