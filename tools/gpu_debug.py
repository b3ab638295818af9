"""Ad-hoc GPU diagnostics: prints where a primitive or a build first disagrees with the checker."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus
from conftest import Oracle, adversarial_cases

def first_diff(a, b):
    d = np.nonzero(a != b)[0]
    return None if d.size == 0 else (int(d[0]), int(d.size))

def check_sort():
    ok = True
    for count in (1, 2, 63, 64, 65, 255, 1024, 1025, 4095, 4096, 4097, 8193, 100000, 1000003):
        for lo, hi in ((0, 8), (8, 16), (0, 64), (5, 62)):
            rng = np.random.default_rng(count)
            keys = rng.integers(0, 2**64, count, dtype=np.uint64)
            if count > 10:
                keys[count // 3: count // 3 + count // 10] &= np.uint64(0xFF)
            vals = np.arange(count, dtype=np.uint32)
            mask = np.uint64((1 << hi) - 1) if hi < 64 else np.uint64(2**64 - 1)
            order = np.argsort((keys & mask) >> np.uint64(lo), kind="stable")
            k2, v2 = keys.copy(), vals.copy()
            rc = sa.diag_lib().sa_amd_test_sort_pairs(k2.ctypes.data, v2.ctypes.data, count, lo, hi)
            fd = first_diff(v2, vals[order])
            if rc or fd:
                ok = False
                print(f"SORT FAIL count={count} bits=({lo},{hi}) rc={rc} first_diff={fd}")
                if fd:
                    i = fd[0]
                    print("   got", v2[max(0, i - 3):i + 5], "exp", vals[order][max(0, i - 3):i + 5])
    print("sort", "ok" if ok else "FAILED")
    return ok

def check_build():
    orc = Oracle()
    ok = True
    cases = dict(adversarial_cases())
    cases["uniform_100k"] = corpus.uniform(100_000, 2).tobytes()
    cases["english_100k"] = corpus.english(100_000, 3).tobytes()
    cases["dna_100k"] = corpus.dna(100_000, 4).tobytes()
    for name, s in cases.items():
        try:
            got = sa.SuffixArray(s).into_parts()[1]
        except Exception as e:
            print("BUILD EXC", name, e); ok = False; continue
        exp = orc.sais(s)
        fd = first_diff(got, exp)
        if fd:
            ok = False
            i = fd[0]
            print(f"BUILD FAIL {name} n={len(s)} first_diff={fd} got={got[max(0,i-2):i+4]} exp={exp[max(0,i-2):i+4]}")
    print("build", "ok" if ok else "FAILED")
    return ok

if __name__ == "__main__":
    a = check_sort()
    b = check_build()
    sys.exit(0 if (a and b) else 1)
