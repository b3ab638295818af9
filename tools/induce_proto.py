"""measure ONE level-0 L-type induce sweep of SA-IS done the way BASELINE.json's north_star sketches it
(wavefront-serialised, bucket heads in LDS, coalesced SA reads, gathered text[SA[i]-1]) on the benchmark corpus, exactly:
the LMS suffixes are placed (taken from the finished suffix array), the sweep runs as one wavefront
(csrc/kernels/induce_proto.hpp, diagnostic library only), and every L-type suffix must come out at its true place.
    python tools/induce_proto.py [workload] [n ...]      -> lines for profiles/r02_induce_lsweep_proto.txt"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus

name = sys.argv[1] if len(sys.argv) > 1 else "c3_english_256m"
sizes = [int(x) for x in sys.argv[2:]] or [4 << 20, 16 << 20, 64 << 20]
D = sa.diag_lib()
D.sa_amd_proto_induce_l.argtypes = [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
D.sa_amd_proto_induce_l.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                    ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
D.sa_amd_proto_induce_l.restype = ctypes.c_int32
for n in sizes:
    T = corpus.workload(name, n_override=n)
    t0 = time.perf_counter()
    final = sa.SuffixArray(T).into_parts()[1]                     # the true suffix array (this engine, verified below)
    t_build = time.perf_counter() - t0
    assert sa.check_integrity(T, final)
    # S / L types: L iff T[i] > T[k+1] at the first k >= i with T[k] != T[k+1]; a run reaching the end is L
    neq = T[:-1] != T[1:]
    idx = np.where(neq, np.arange(n - 1, dtype=np.int64), np.int64(n))
    nxt = np.minimum.accumulate(idx[::-1])[::-1]
    isL = np.ones(n, dtype=bool)
    has = nxt < n
    k = nxt[has[: n - 1].nonzero()[0]] if False else None
    sel = np.nonzero(has)[0]
    isL[sel] = T[nxt[sel]] > T[nxt[sel] + 1]
    lms = np.zeros(n, dtype=bool)
    lms[1:] = (~isL[1:]) & isL[:-1]
    typeL = np.packbits(isL, bitorder="little")
    typeL = np.concatenate([typeL, np.zeros(16, dtype=np.uint8)])
    cnt = np.bincount(T, minlength=256)
    head = (1 + np.concatenate([[0], np.cumsum(cnt)[:-1]])).astype(np.uint32)
    SA0 = np.full(n + 1, 0xFFFFFFFF, dtype=np.uint32)
    SA0[0] = n
    body = final[1:]
    keep = lms[body]
    SA0[1:][keep] = body[keep]
    ms = ctypes.c_double()
    counters = np.zeros(2, dtype=np.uint64)
    rc = D.sa_amd_proto_induce_l(T.ctypes.data, typeL.ctypes.data, SA0.ctypes.data, n, head.ctypes.data, ctypes.byref(ms), counters.ctypes.data)
    assert rc == 0, rc
    wantL = isL[body]
    okL = np.array_equal(SA0[1:][wantL], body[wantL])
    rest = ~wantL & ~keep
    ok_rest = bool((SA0[1:][rest] == 0xFFFFFFFF).all())
    nL = int(wantL.sum())
    print(f"{name} n={n}: L-sweep by one wavefront {ms.value:10.1f} ms for {nL} L-type suffixes ({ms.value*1e6/max(nL,1):6.1f} ns per induced suffix, "
          f"{ms.value*1e6/(n+1):6.1f} ns per slot), {int(counters[1])} block re-reads; exact: {okL and ok_rest} "
          f"(induced {int(counters[0])}); the whole build of this engine on the same text: {t_build*1e3:.1f} ms end to end", flush=True)
