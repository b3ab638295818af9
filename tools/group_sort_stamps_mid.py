"""Diagnostic: where a 2048-suffix tile of k_group_sort spends its time (s_memtime stamps of thread 0, C3)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
t = corpus.english_corpus(int(sys.argv[1]) if len(sys.argv) > 1 else 1048576, 3)
out = np.zeros(t.size + 1, dtype=np.uint32)
L = sa.diag_lib()        # the stamps are compiled into the diagnostic library only
saca = lambda text, arr: L.sa_amd_saca_u8(text.ctypes.data, arr.ctypes.data, text.size)
buf = (ctypes.c_uint64 * 16)()
assert saca(t, out) == 0
L.sa_amd_debug_group_sort_stamps(1)
L.sa_amd_debug_phase_cycles(buf, 16)          # zero
assert saca(t, out) == 0
L.sa_amd_debug_phase_cycles(buf, 16)
L.sa_amd_debug_group_sort_stamps(0)
names = ["list loads issued", "secondary keys gathered (+ wait for the list loads)", "keys + bitmap in LDS, next-start table (3 barriers)",
         "group extents + rank loops", "permute through LDS (2 barriers)", "stores issued"]
tot = sum(buf[8 + i] for i in range(6))
for i, nm in enumerate(names):
    print(f"{nm:56s} {100.0 * buf[8 + i] / max(tot, 1):5.1f} %   {buf[8 + i]}")
