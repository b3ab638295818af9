"""Diagnostic: where a tile of the single-pass tile scatter (kernels/onesweep.hpp) spends its cycles -- in-kernel s_memtime
stamps of wave 0, through libsuffix_array_amd_diag.so (SA_AMD_ONESWEEP_FLAGS bit 7 is read by the diagnostic build only).
    python tools/onesweep_stamps.py [32|64] [log2 count = 26] [shape = 0]"""
import ctypes, os, sys
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 26
os.environ["SA_AMD_ONESWEEP_FLAGS"] = "128"
shape = int(sys.argv[3]) if len(sys.argv) > 3 else 0
os.environ["SA_AMD_ONESWEEP%d_SHAPE" % kb] = str(shape)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import suffix_array_amd as sa
L = sa.diag_lib()
n = 1 << lg
rng = np.random.default_rng(1)
vals = np.arange(n, dtype=np.uint32)
buf = (ctypes.c_uint64 * 16)()
if kb == 32:
    keys = rng.integers(0, 2**32, n, dtype=np.uint32)
    def run():
        k, v = keys.copy(), vals.copy()          # (the hook sorts in place; keep the copies alive during the call)
        return L.sa_amd_test_sort_pairs32(k.ctypes.data, v.ctypes.data, n, 0, 32)
    tile, passes = [12288, 8192, 6144, 8192][shape], 4
else:
    keys = rng.integers(0, 2**64, n, dtype=np.uint64)
    def run():
        k, v = keys.copy(), vals.copy()
        return L.sa_amd_test_sort_pairs(k.ctypes.data, v.ctypes.data, n, 0, 64)
    tile, passes = [8192, 8192, 4096][shape], 8
assert run() == 0
L.sa_amd_debug_phase_cycles(buf, 16)          # discard warm-up
assert run() == 0
L.sa_amd_debug_phase_cycles(buf, 16)
names = ["ticket, key loads, zero, barrier (key wait)", "ranking", "value loads issued + barrier", "totals, publish, look-back issue, prefix",
         "keys (+ values) -> LDS", "look-back finish, barrier", "LDS -> global (+ next digit count)", "values through the stage (SEQ shapes)"]
tot = sum(buf[i] for i in range(8))
tiles = passes * ((n + tile - 1) // tile)
print(f"{kb}-bit keys, 2^{lg} pairs, {passes} passes, shape {shape}, tile {tile}; s_memtime ticks of wave 0 per tile")
for i, nm in enumerate(names):
    print(f"{nm:44s} {buf[i] / tiles:9.1f} ticks/tile  {100.0 * buf[i] / tot:5.1f} %")
print(f"{'total':44s} {tot / tiles:9.1f} ticks/tile")
if buf[8]:
    print(f"look-back of one digit's thread: {buf[8]} walks, {buf[9] / buf[8]:.2f} tiles walked (rounded up to rounds of 4), {buf[10] / buf[8]:.2f} polls of a granule that was not there yet per walk")
