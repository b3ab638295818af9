"""Regenerates the golden (text, SA) fixtures in this directory.

The reference cannot be run in this image (no cargo; its engine `cdivsufsort` is not vendored),
so the expected arrays come from oracle_naive_sa -- a comparison sort with the exact ordering of
reference src/sa.rs:76-82 -- and are cross-checked with oracle_sais and oracle_check_integrity
(the literal restatement of reference src/sa.rs:72-84).  The suffix array of a text is unique, so
these are the arrays the reference's divsufsort path produces.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import Oracle, fibonacci_word, thue_morse   # noqa: E402
from suffix_array_amd import corpus                        # noqa: E402


def main():
    orc = Oracle()
    cases = {
        "doctest_splendid": b"splendid splendor",               # reference src/lib.rs:16-41
        "mississippi": b"mississippi",
        "c1_uniform_1k_seed1": corpus.uniform(1024, 1).tobytes(),   # BASELINE.json configs[0]
        "english_4k_seed3": corpus.english(4096, 3).tobytes(),
        "dna_4k_seed4": corpus.dna(4096, 4).tobytes(),
        "fibonacci_15": fibonacci_word(15),
        "thue_morse_2k": thue_morse(2048),
        "a_run_1025": b"a" * 1025,
        "zeros_then_ffs": b"\x00" * 200 + b"\xff" * 200,
        "all_bytes_twice": bytes(range(256)) * 2,
    }
    manifest = {}
    for name, text in cases.items():
        sa = orc.naive(text)
        assert np.array_equal(sa, orc.sais(text)), name
        assert orc.check_integrity(text, sa) == 1, name
        with open(os.path.join(HERE, name + ".text"), "wb") as f:
            f.write(text)
        sa.astype("<u4").tofile(os.path.join(HERE, name + ".sa.u32le"))
        manifest[name] = {"n": len(text), "sa_len": int(sa.size)}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(manifest), "fixtures")


if __name__ == "__main__":
    main()
