"""Randomised stress test on the GPU: random sizes / alphabets / structures, every regime switch
toggled at random, each result compared bit for bit with the oracle.  python tools/stress.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
from conftest import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
orc = Oracle()
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
BIG = int(sys.argv[3]) if len(sys.argv) > 3 else 400000
ENV = ["SA_AMD_FORCE_TOP32", "SA_AMD_NO_LOCAL_SORT", "SA_AMD_NO_TEXT_ROUNDS", "SA_AMD_FORCE_DENSE", "SA_AMD_BINNED_ISA_ALWAYS",
       "SA_AMD_NO_TOP32", "SA_AMD_FUSED64", "SA_AMD_NO_FUSED_FINISH", "SA_AMD_NO_PACKED_TEXT", "SA_AMD_NO_REPEAT_PROBE",
       "SA_AMD_NO_RUN_SKIP", "SA_AMD_NO_GRAM_KEYS", "SA_AMD_NO_SPLIT", "SA_AMD_NO_ONESWEEP", "SA_AMD_NO_BIG_GROUP_SORT",
       "SA_AMD_NO_FIRST_TAIL", "SA_AMD_NO_BUCKET_SORT", "SA_AMD_NO_BUCKET_FINISH", "SA_AMD_BUCKET_FINISH_ALWAYS", "SA_AMD_NO_TEXT_KEYS",
       "SA_AMD_NO_DEFER", "SA_AMD_NO_UNARY_SHORTCUT", "SA_AMD_NO_VALUE_BITS", "SA_AMD_NO_POSTED_READBACK", "SA_AMD_NO_FLAT_RULE", "SA_AMD_NO_KERNEL_D2H", "SA_AMD_KERNEL_D2H_ALWAYS", "SA_AMD_NO_UPFRONT_COUNTS"]
NUM = {"SA_AMD_SPARSE_DIV": [1, 4, 64, 10**9], "SA_AMD_GROUP_CAP": [2, 3, 7, 40, 300], "SA_AMD_CHASE": [1, 2, 3, 7, 15],
       "SA_AMD_SCATTER_LEVELS": [1, 2], "SA_AMD_DENSE_REKEY_MIN": [1, 1000], "SA_AMD_MAX_TEXT_ROUNDS": [0, 1, 2, 6],
       "SA_AMD_BINNED_MIN": [1, 5000], "SA_AMD_KEY_BITS": [16, 24, 40, 56], "SA_AMD_RUN_SKIP_MIN": [1, 100000],
       "SA_AMD_GRAM_MIN_N": [1, 1, 1000], "SA_AMD_GRAM_G": [0, 2, 3, 4, 8], "SA_AMD_GRAM_TAIL": [0, 1, 2, 8], "SA_AMD_CHASE_BIG": [1, 2, 5], "SA_AMD_CHASE_BIG_MIN": [1, 5000],
       "SA_AMD_SPLIT_MIN": [1, 1, 3000], "SA_AMD_SPLIT_GROUP_MIN": [1, 2, 50, 1000],
       "SA_AMD_SMALL_MAX": [0, 0, 0, 100, 8192], "SA_AMD_ONESWEEP64_SHAPE": [0, 1, 2], "SA_AMD_ONESWEEP32_SHAPE": [0, 1, 2, 3],
       "SA_AMD_BUCKET_MIN_N": [1, 1, 1, 100000], "SA_AMD_BUCKET_BITS": [0, 16, 18], "SA_AMD_BUCKET_SHAPE": [-1, 0, 1, 2, 3, 4],
       "SA_AMD_ONESWEEP_FLAGS": [0, 1], "SA_AMD_TOP32_PARTNERS_X100": [0, 50, 100000], "SA_AMD_TOP32_COLLISIONS_X100": [0, 400, 100000],
       # the early download at small sizes: staged download for every array, 64 KiB pulls, start at n / d tied suffixes, the build waits for w chunks
       "SA_AMD_STAGED_MIN_BYTES": [0, 0, 1 << 20], "SA_AMD_EARLY_MIN_BYTES": [0, 0, 1 << 40], "SA_AMD_EARLY_CHUNK_BYTES": [65536, 65536, 262144],
       "SA_AMD_EARLY_DIV": [0, 1, 1, 2, 8], "SA_AMD_EARLY_WAIT_CHUNKS": [0, 1, 3, 8], "SA_AMD_COPY_THREADS": [0, 1, 4, 12], "SA_AMD_HELPER_THREADS": [0, 2, 16],
       "SA_AMD_PREFAULT_KEEP": [0, 1, 3, 12], "SA_AMD_NETWORK_MIN": [0, 1, 8, 300, 2000], "SA_AMD_COUNT_NEXT_MIN_N": [0, 1 << 40], "SA_AMD_COUNT_NEXT_BELOW_N": [0, 1 << 40, 5000]}
t0 = time.time(); cases = 0; fails = 0
while time.time() - t0 < budget:
    n = int(rng.choice([rng.integers(0, 300), rng.integers(300, 20000), rng.integers(20000, BIG)]))
    kind = int(rng.integers(0, 9))
    if kind == 0:
        s = rng.integers(0, 256, n, dtype=np.uint8)
    elif kind == 1:
        sig = int(rng.integers(1, 6)); s = (rng.integers(0, sig, n) + int(rng.integers(0, 250))).astype(np.uint8)
    elif kind == 2:
        s = corpus.english(n, int(rng.integers(0, 1 << 30))) if n else np.zeros(0, np.uint8)
    elif kind == 3:
        per = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8); s = np.resize(per, n).astype(np.uint8) if n else np.zeros(0, np.uint8)
    elif kind == 4:
        s = rng.integers(0, 256, n, dtype=np.uint8)
        for _ in range(int(rng.integers(1, 6))):
            if n > 50:
                ln = int(rng.integers(2, max(3, n // 3))); a = int(rng.integers(0, n - ln)); b = int(rng.integers(0, n - ln)); s[b:b + ln] = s[a:a + ln]
    elif kind == 5:
        s = np.concatenate([rng.integers(0, 3, n // 2, dtype=np.uint8), np.zeros(n - n // 2, dtype=np.uint8)])
    elif kind == 6:
        s = corpus.dna_repeats(n, int(rng.integers(0, 1 << 30)), 0.4) if n > 5000 else rng.integers(65, 69, n, dtype=np.uint8)
    elif kind == 7:
        s = corpus.english_corpus(n, int(rng.integers(0, 1 << 30)), 2000, 0.3) if n else np.zeros(0, np.uint8)
    else:
        # a block repeated a few thousand times: every suffix tied with thousands of others (k_group_sort_big's groups), plus a tail
        per = rng.integers(0, 256, int(rng.integers(30, 400)), dtype=np.uint8)
        s = np.concatenate([np.resize(per, n - n // 8), rng.integers(0, 256, n // 8, dtype=np.uint8)]) if n else np.zeros(0, np.uint8)
    s = np.ascontiguousarray(s, dtype=np.uint8)
    for k in ENV + list(NUM):
        os.environ.pop(k, None)
    chosen = [k for k in ENV if rng.random() < 0.2]
    for k in chosen:
        os.environ[k] = "1"
    for k, vals in NUM.items():
        if rng.random() < 0.25:
            os.environ[k] = str(int(rng.choice(vals))); chosen.append(k[7:] + "=" + os.environ[k])
    try:
        got = sa.SuffixArray(s).into_parts()[1]
        ok = np.array_equal(got, orc.sais(s))
    except Exception as e:
        ok = False; print("EXC", e)
    cases += 1
    if cases % 2000 == 0:
        print(f"  ... {cases} cases, {fails} failures, {time.time()-t0:.0f} s", flush=True)   # (a silent run is taken to be hung)
    if not ok:
        fails += 1
        print("FAIL n", n, "kind", kind, "env", chosen, sa.last_stats())
        np.save(f"gpurun_out/stress_fail_{fails}.npy", s)
        if fails >= 5: break
print(f"stress: {cases} cases, {fails} failures, {time.time()-t0:.0f} s")
sys.exit(1 if fails else 0)
