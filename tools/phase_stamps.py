"""Diagnostic: where a tile of the dominant kernel spends its cycles (in-kernel s_memtime stamps).
Runs through libsuffix_array_amd_diag.so: the stamped kernel and SA_AMD_TIMING_ONLY_INITIAL_SORT exist only there
(variant 9 of the diagnostic table = phase stamps; sa_amd_debug_sort_variant_name lists them)."""
import sys, os, ctypes
os.environ["SA_AMD_SORT_VARIANT"] = sys.argv[1] if len(sys.argv) > 1 else "9"
os.environ["SA_AMD_TIMING_ONLY_INITIAL_SORT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
t = corpus.uniform(256 << 20, 5)
out = np.zeros(t.size + 1, dtype=np.uint32)
L = sa.diag_lib()
saca = lambda text, arr: L.sa_amd_saca_u8(text.ctypes.data, arr.ctypes.data, text.size)
buf = (ctypes.c_uint64 * 16)()
assert saca(t, out) == 0
L.sa_amd_debug_phase_cycles(buf, 16)          # discard warm-up
assert saca(t, out) == 0
L.sa_amd_debug_phase_cycles(buf, 16)
names = ["load issue + zero + barrier", "ranking (+ key wait)", "value loads + barrier", "digit prefix + bookkeeping",
         "keys -> LDS + carry out", "keys LDS -> global", "values -> LDS + carry out", "values LDS -> global"]
tot = sum(buf[i] for i in range(8))
tiles = 8 * (t.size // 8192)
for i, nm in enumerate(names):
    print(f"{nm:32s} {buf[i] / tiles:10.0f} cycles/tile  {100.0 * buf[i] / tot:5.1f} %")
print(f"{'total':32s} {tot / tiles:10.0f} cycles/tile (100 MHz s_memtime ticks x ... see MI355X_MICROARCH)")
