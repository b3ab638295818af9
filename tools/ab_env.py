"""A/B of environment knobs on device-resident builds: python tools/ab_env.py workload "K=V,K=V" "K=V" ...
('-' = default environment).  Prints ms per build (best of 3), rounds and radix passes; every array is checked with the
device integrity check."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import suffix_array_amd as sa
from suffix_array_amd import corpus
name = sys.argv[1]
if name.startswith("adv:"):                     # adversarial families: adv:<one|ab|fib|p1000|twice>:<n>
    _, fam, nn = name.split(":"); nn = int(nn)
    def _fib(k):
        a, b = b"a", b"ab"
        while len(b) < k: a, b = b, b + a
        return np.frombuffer(b[:k], dtype=np.uint8).copy()
    t = {"one": lambda: np.full(nn, 97, dtype=np.uint8), "ab": lambda: np.resize(np.frombuffer(b"ab", dtype=np.uint8), nn).copy(),
         "fib": lambda: _fib(nn), "twice": lambda: np.concatenate([corpus.english(nn // 2, 5)] * 2),
         "p1000": lambda: np.resize(np.random.default_rng(7).integers(0, 256, 1000, dtype=np.uint8), nn).copy()}[fam]()
elif "@" in name:                               # <workload>@<n>: the workload's generator at another size
    t = corpus.workload(name.split("@")[0], 0, int(name.split("@")[1]))
else:
    t = corpus.workload(name)
n = t.size
dev = torch.device("cuda", 0)
text = torch.from_numpy(t).to(dev)
out = torch.empty(n + 1, dtype=torch.int32, device=dev)
wb = sa.workspace_bytes(n)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
L = sa.lib()
ci_bytes = int(L.sa_amd_check_integrity_work_bytes(n))
ci = torch.empty(ci_bytes, dtype=torch.uint8, device=dev)
st = sa.Stats()
L = sa.lib()
for spec in sys.argv[2:]:
    saved = dict(os.environ)
    if spec != "-":
        for kv in spec.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sa.saca_device_ptr(text.data_ptr(), out.data_ptr(), n, work.data_ptr(), wb, 0, st)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    ok = L.sa_amd_check_integrity_device(text.data_ptr(), n, out.data_ptr(), ci.data_ptr(), ci_bytes, None)
    print(f"{name} [{spec}]: {best*1e3:8.2f} ms  rounds {st.rounds} (text {st.text_rounds}) passes {st.sort_passes} verified {ok == 1}", flush=True)
    os.environ.clear(); os.environ.update(saved)
