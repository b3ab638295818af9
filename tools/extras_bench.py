"""Rows (f) of SURVEY.md section 8 besides search, measured on the headline text: bucket table (reference src/sa.rs:89-119),
integrity check (src/sa.rs:72-84), packed format (src/packed_sa.rs) -- through the host-pointer C ABI and, where there is
one, on the device-resident index.  python tools/extras_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus

def best(f, reps=3):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); b = min(b, time.perf_counter() - t0)
    return b, r

t = corpus.workload("c3_english_256m")
n = t.size
arr = np.zeros(n + 1, dtype=np.uint32)
sa.saca(t, arr)
ix = sa.DeviceIndex(t, arr)
dt, ok = best(lambda: ix.check_integrity())
print(f"check_integrity, device-resident index      : {dt*1e3:8.2f} ms  -> {ok}")
dt, ok = best(lambda: sa.check_integrity(t, arr))
print(f"check_integrity, host text + array in       : {dt*1e3:8.2f} ms  -> {ok}")
dt, b1 = best(lambda: ix.buckets())
print(f"bucket table (65 537 entries), device index : {dt*1e3:8.2f} ms")
dt, b2 = best(lambda: sa.bucket_table(t, arr))
print(f"bucket table, host text + array in          : {dt*1e3:8.2f} ms  (equal: {bool(np.array_equal(b1, b2))})")
dt, blob = best(lambda: sa.pack(arr), 2)
print(f"pack   {4*(n+1)/1e6:8.1f} MB -> {len(blob)/1e6:8.1f} MB         : {dt*1e3:8.2f} ms = {4*(n+1)/dt/1e9:5.2f} GB/s of array")
dt, back = best(lambda: sa.unpack(blob), 2)
print(f"unpack                                      : {dt*1e3:8.2f} ms = {4*(n+1)/dt/1e9:5.2f} GB/s of array (round trip equal: {bool(np.array_equal(back, arr))})")
