"""CPU suite (-m "not gpu"): pins the oracle against the reference's known answers, its own test
domain (reference src/tests.rs:6-17) and the committed golden fixtures."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import KNOWN_ANSWERS, ROOT, adversarial_cases

import pd_model

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("text,expected", KNOWN_ANSWERS)
def test_known_answers(oracle, text, expected):
    assert oracle.naive(text).tolist() == expected
    assert oracle.sais(text).tolist() == expected
    assert oracle.check_integrity(text, np.array(expected, dtype=np.uint32)) == 1
    assert oracle.verify(text, np.array(expected, dtype=np.uint32)) == 1


def test_doctest_search_all_vector(oracle):
    """reference src/lib.rs:28-29: search_all(b"splend") == [0, 9] on b"splendid splendor"."""
    s = b"splendid splendor"
    sa = oracle.sais(s)
    hits = sorted(int(p) for p in sa if s[int(p):].startswith(b"splend"))
    assert hits == [0, 9]
    pos = [i for i, p in enumerate(sa) if int(p) in (0, 9)]
    assert pos[1] - pos[0] == 1      # one contiguous SA range


@settings(max_examples=200, deadline=None)
@given(st.binary(min_size=0, max_size=4095))
def test_conversion_correctness_domain(oracle, s):
    """reference src/tests.rs:13-17, with the oracle standing in for SuffixArray::new."""
    sa = oracle.sais(s)
    assert np.array_equal(sa, oracle.naive(s))
    assert oracle.check_integrity(s, sa) == 1
    assert oracle.verify(s, sa) == 1


@pytest.mark.parametrize("name", sorted(adversarial_cases()))
def test_adversarial(oracle, name):
    s = adversarial_cases()[name]
    sa = oracle.sais(s)
    assert np.array_equal(sa, oracle.naive(s))
    assert oracle.check_integrity(s, sa) == 1
    assert np.array_equal(pd_model.build(s), sa)


@settings(max_examples=120, deadline=None)
@given(st.integers(0, 2500), st.sampled_from([1, 2, 3, 4, 5, 16, 64, 200, 256]), st.integers(0, 2**32 - 1))
def test_device_algorithm_model(oracle, n, sigma, seed):
    """the numpy restatement of the GPU pipeline (packed keys, end-of-text rule, doubling)"""
    rng = np.random.default_rng(seed)
    s = rng.integers(0, sigma, n, dtype=np.uint8).tobytes()
    assert np.array_equal(pd_model.build(s), oracle.sais(s))


def test_checkers_reject_wrong_arrays(oracle):
    s = b"mississippi"
    sa = oracle.sais(s)
    bad = sa.copy(); bad[3], bad[4] = bad[4], bad[3]
    assert oracle.check_integrity(s, bad) == 0 and oracle.verify(s, bad) == 0
    assert oracle.check_integrity(s, sa[:-1]) == 0 and oracle.verify(s, sa[:-1]) == 0
    dup = sa.copy(); dup[5] = dup[6]
    assert oracle.verify(s, dup) == 0
    oob = sa.copy(); oob[2] = 99
    assert oracle.check_integrity(s, oob) == -1      # the reference panics on the slice index


def test_golden_fixtures(oracle):
    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    assert len(manifest) >= 10
    for name, meta in manifest.items():
        text = open(os.path.join(GOLDEN, name + ".text"), "rb").read()
        sa = np.fromfile(os.path.join(GOLDEN, name + ".sa.u32le"), dtype="<u4")
        assert len(text) == meta["n"] and sa.size == meta["sa_len"] == len(text) + 1
        assert np.array_equal(oracle.sais(text), sa), name
        assert oracle.check_integrity(text, sa) == 1, name


def test_oracle_medium_sizes(oracle):
    from suffix_array_amd import corpus
    for text in (corpus.english(300_000, 3), corpus.dna(300_000, 4), corpus.uniform(300_000, 2),
                 corpus.dna_repeats(300_000, 5, 0.3)):
        sa = oracle.sais(text)
        assert oracle.verify(text, sa) == 1
        assert np.array_equal(pd_model.build(text.tobytes()), sa)


def test_bucket_table_restatement(oracle):
    """reference src/sa.rs:89-119: bkt[i] is the exclusive right edge of bucket i in the SA."""
    s = b"splendid splendor"
    bkt = oracle.bucket_table(s)
    sa = oracle.sais(s)
    assert bkt[0] == 1 and bkt[-1] == len(s) + 1
    c0, c1 = ord("s"), ord("p")
    idx = c0 * 257 + (c1 + 1) + 1
    rng = sa[bkt[idx - 1]:bkt[idx]]
    assert sorted(int(p) for p in rng) == [0, 9]


def test_pack_model_round_trip_and_layout():
    """numpy restatement of the packed format (reference src/packed_sa.rs); byte-level parity with the
    external bitpacking crate is unpinned, the round trip is what the reference's own test pins"""
    import pack_model
    rng = np.random.default_rng(4)
    for length in (1, 2, 3, 127, 128, 129, 255, 256, 1000, 4097):
        sa = rng.permutation(length).astype(np.uint32)
        blob = pack_model.pack(sa)
        assert np.array_equal(pack_model.unpack(blob), sa)
        bits = pack_model.sa_bits(length)
        assert len(blob) <= 16 + ((length + 127) // 128) * bits * 16
    # hand-checked block: bits = 7 for length 128; value 4 i + c of the identity sits at row i of lane c
    blob = pack_model.pack(np.arange(128, dtype=np.uint32))
    assert blob[:4] == b"SA4x" and blob[4:8] == (128).to_bytes(4, "little") and len(blob) == 16 + 7 * 16
    lane0_word0 = int.from_bytes(blob[16:20], "little")
    assert lane0_word0 & 0x7F == 0 and (lane0_word0 >> 7) & 0x7F == 4 and (lane0_word0 >> 14) & 0x7F == 8
    lane1_word0 = int.from_bytes(blob[20:24], "little")
    assert lane1_word0 & 0x7F == 1 and (lane1_word0 >> 7) & 0x7F == 5


def test_sanitized_selftest():
    """the oracle and the corpus generators under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle sanitize`):
    known answers, naive sort == SA-IS == both integrity checks, the threaded verifier, every generator at ragged sizes.
    (GPU sanitizers are unavailable on this pool: the CPU side of the test infrastructure is what gets sanitized.)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    proc = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=600)
    assert proc.returncode == 0 and "selftest ok" in proc.stdout, proc.stdout[-3000:]
