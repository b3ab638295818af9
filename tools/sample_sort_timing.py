"""The initial sort of the 64-bit stage, sample sort against the LSD engine, on the C3 corpus (or another generator):
    python tools/sample_sort_timing.py [workload] [builds]
prints ms per build and the per-class kernel table of one profiled build for SA_AMD_SAMPLE_SORT=0 and =1 (diagnostic library)"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import suffix_array_amd as sa
from suffix_array_amd import corpus
name = sys.argv[1] if len(sys.argv) > 1 else "c3_english_256m"
builds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t = corpus.workload(name)
n = int(t.size)
dev = torch.device("cuda", 0)
text = torch.from_numpy(t).to(dev)
out = torch.empty(n + 1, dtype=torch.int32, device=dev)
wb = sa.workspace_bytes(n)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
L = sa.diag_lib()          # (the sample sort lives in the diagnostic library)
for fn, at, rt in (("sa_amd_profile_kernel_name", [ctypes.c_int32], ctypes.c_char_p), ("sa_amd_profile_begin_classes", [ctypes.c_uint64], None),
                   ("sa_amd_profile_end", [ctypes.c_void_p] * 3 + [ctypes.c_int32], ctypes.c_int32),
                   ("sa_amd_check_integrity_work_bytes", [ctypes.c_int32], ctypes.c_int64),
                   ("sa_amd_check_integrity_device", [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p], ctypes.c_int32)):
    getattr(L, fn).argtypes = at; getattr(L, fn).restype = rt
stats = sa.Stats()
names = []
while True:
    nm = L.sa_amd_profile_kernel_name(len(names)).decode()
    if not nm:
        break
    names.append(nm)
for flag in ("0", "1"):
    os.environ["SA_AMD_SAMPLE_SORT"] = flag
    st = torch.cuda.current_stream().cuda_stream
    def step():
        assert L.sa_amd_saca_device(text.data_ptr(), out.data_ptr(), n, work.data_ptr(), wb, st, ctypes.byref(stats)) == 0
    step()
    L.sa_amd_profile_begin_classes(ctypes.c_uint64(~0 & 0xFFFFFFFFFFFFFFFF))
    step(); torch.cuda.synchronize()
    cap = 32
    ms, launches, units = (ctypes.c_double * cap)(), (ctypes.c_int64 * cap)(), (ctypes.c_int64 * cap)()
    k = L.sa_amd_profile_end(ms, launches, units, cap)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(builds):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / builds
    ci = int(L.sa_amd_check_integrity_work_bytes(n))
    cw = torch.empty(ci, dtype=torch.uint8, device=dev)
    ok = L.sa_amd_check_integrity_device(text.data_ptr(), n, out.data_ptr(), cw.data_ptr(), ci, st) == 1
    del cw
    print(f"== {name} SA_AMD_SAMPLE_SORT={flag}: {dt*1e3:.2f} ms per build, verified {ok}, rounds {stats.rounds}, passes {stats.sort_passes}")
    print("   " + "  ".join(f"{names[i]} {ms[i]:.2f}/{launches[i]}" for i in range(k) if launches[i]))
