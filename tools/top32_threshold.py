"""Where does the 32-bit first stage (sort the top 32 key bits, finish the ties by the low bits) stop paying?  DNA with a
growing share of planted repeats and skewed alphabets, default route against SA_AMD_FORCE_TOP32 / SA_AMD_NO_TOP32, with the
probe's estimate (expected partners per suffix) printed by SA_AMD_VERBOSE=3.   python tools/top32_threshold.py [log2 n = 30]"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import suffix_array_amd as sa
from suffix_array_amd import corpus
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << lg
dev = torch.device("cuda", 0)
L = sa.lib()
cases = [("dna iid", lambda: corpus.dna(n, 4))] + [(f"dna, {int(f*100)} % in repeats", (lambda f: (lambda: corpus.dna_repeats(n, 4, f)))(f)) for f in (0.1, 0.2, 0.4, 0.6, 0.8)]
cases += [("sigma 16 iid", lambda: corpus.sigma(n, 5, 16, 65)), ("sigma 16 skewed", lambda: (np.minimum(corpus.sigma(n, 6, 16, 0), corpus.sigma(n, 7, 16, 0)) + 65).astype(np.uint8))]
for name, gen in cases:
    t = gen()
    text = torch.from_numpy(t).to(dev)
    out = torch.empty(n + 1, dtype=torch.int32, device=dev)
    wb = sa.workspace_bytes(n)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    st = sa.Stats()
    res = []
    for env in ({}, {"SA_AMD_FORCE_TOP32": "1"}, {"SA_AMD_NO_TOP32": "1"}):
        for k in ("SA_AMD_FORCE_TOP32", "SA_AMD_NO_TOP32"):
            os.environ.pop(k, None)
        os.environ.update(env)
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sa.saca_device_ptr(text.data_ptr(), out.data_ptr(), n, work.data_ptr(), wb, 0, st)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res.append((best * 1e3, st.top32_first, st.unresolved_after_initial))
    os.environ["SA_AMD_VERBOSE"] = "3"
    for k in ("SA_AMD_FORCE_TOP32", "SA_AMD_NO_TOP32"):
        os.environ.pop(k, None)
    sys.stderr.flush()
    sa.saca_device_ptr(text.data_ptr(), out.data_ptr(), n, work.data_ptr(), wb, 0, st)
    os.environ["SA_AMD_VERBOSE"] = "0"
    print(f"{name:26s} default {res[0][0]:8.2f} ms (top32 {res[0][1]})   forced 32-bit stage {res[1][0]:8.2f} ms   full keys {res[2][0]:8.2f} ms   tied after the initial sort {res[2][2]}", flush=True)
    del text, out, work
    torch.cuda.empty_cache()
