"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
bytes.  Bit-exact everywhere (integer / index work)."""
import ctypes
import json
import os

import numpy as np
import pytest

import suffix_array_amd as sa
from suffix_array_amd import corpus
from conftest import KNOWN_ANSWERS, adversarial_cases, fibonacci_word

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(text):
    return sa.SuffixArray(text).into_parts()[1]


def build_diag(text):
    """the same build through libsuffix_array_amd_diag.so: the engines that exist only there (the three-kernel radix pass of
    rounds 1-2 behind SA_AMD_NO_ONESWEEP / SA_AMD_SORT_VARIANT, the sample sort) are reached through its copy of the entry point"""
    t = np.ascontiguousarray(np.frombuffer(bytes(text), dtype=np.uint8) if not isinstance(text, np.ndarray) else text)
    out = np.zeros(t.size + 1, dtype=np.uint32)
    assert sa.diag_lib().sa_amd_saca_u8(t.ctypes.data if t.size else None, out.ctypes.data, t.size) == 0
    return out


def test_device_visible():
    assert sa.lib().sa_amd_device_count() >= 1


# ---- primitives -------------------------------------------------------------------------------

@pytest.mark.parametrize("count,bits", [(1, (0, 64)), (2, (0, 64)), (63, (0, 64)), (4095, (0, 64)), (4096, (0, 64)),
                                        (4097, (0, 64)), (100_000, (0, 64)), (1_000_003, (0, 64)),
                                        (300_000, (0, 24)), (300_000, (8, 40)), (300_000, (5, 62)),
                                        (5_000_000, (0, 64)),
                                        # tile / segment geometry of the single-pass scatter (8192-pair tiles, up to 8 segments)
                                        (8191, (0, 16)), (8192, (0, 16)), (8193, (0, 16)), (8192 * 8, (0, 24)), (8192 * 8 + 1, (0, 24)),
                                        (8192 * 9 + 5, (3, 21)), (8192 * 17 - 1, (0, 9)), (20_000_003, (0, 40))])
@pytest.mark.parametrize("engine", ["single-pass", "shape1", "shape2", "three-kernel"])
def test_radix_sort_pairs(count, bits, engine, monkeypatch):
    if engine == "three-kernel":
        monkeypatch.setenv("SA_AMD_NO_ONESWEEP", "1")
    if engine.startswith("shape"):
        monkeypatch.setenv("SA_AMD_ONESWEEP64_SHAPE", engine[5:])
    rng = np.random.default_rng(count + bits[1])
    keys = rng.integers(0, 2**64, count, dtype=np.uint64)
    if count > 10:
        keys[count // 3: count // 3 + count // 10] &= np.uint64(0xFF)       # many duplicates
    vals = np.arange(count, dtype=np.uint32)
    lo, hi = bits
    mask = np.uint64((1 << hi) - 1) if hi < 64 else np.uint64(2**64 - 1)
    field = (keys & mask) >> np.uint64(lo)
    order = np.argsort(field, kind="stable")
    k2, v2 = keys.copy(), vals.copy()
    assert sa.diag_lib().sa_amd_test_sort_pairs(k2.ctypes.data, v2.ctypes.data, count, lo, hi) == 0
    assert np.array_equal(v2, vals[order])          # stable: equal fields keep input order
    assert np.array_equal(k2, keys[order])


@pytest.mark.parametrize("count,bits", [(1, (0, 32)), (3, (0, 32)), (16383, (0, 32)), (16384, (0, 32)), (16385, (0, 32)),
                                        (100_001, (0, 32)), (2_000_003, (0, 32)), (300_000, (8, 24)), (5_000_000, (0, 32)),
                                        (12287, (0, 16)), (12288, (0, 16)), (12289, (0, 16)), (12288 * 8 + 1, (0, 24)),
                                        (12288 * 9 + 5, (3, 21)), (8192 * 9 + 5, (0, 32)), (30_000_001, (0, 32))])
@pytest.mark.parametrize("engine", ["single-pass", "shape1", "shape2", "shape3", "three-kernel"])
def test_radix_sort_pairs_32bit_keys(count, bits, engine, monkeypatch):
    if engine == "three-kernel":
        monkeypatch.setenv("SA_AMD_NO_ONESWEEP", "1")
    if engine.startswith("shape"):
        monkeypatch.setenv("SA_AMD_ONESWEEP32_SHAPE", engine[5:])
    rng = np.random.default_rng(count + bits[1])
    keys = rng.integers(0, 2**32, count, dtype=np.uint32)
    if count > 10:
        keys[count // 3: count // 3 + count // 10] &= np.uint32(0xFF)
        keys[-(count // 7):] = 0x01020304
    vals = np.arange(count, dtype=np.uint32)
    lo, hi = bits
    field = (keys.astype(np.uint64) & np.uint64((1 << hi) - 1)) >> np.uint64(lo)
    order = np.argsort(field, kind="stable")
    k2, v2 = keys.copy(), vals.copy()
    assert sa.diag_lib().sa_amd_test_sort_pairs32(k2.ctypes.data, v2.ctypes.data, count, lo, hi) == 0
    assert np.array_equal(v2, vals[order]) and np.array_equal(k2, keys[order])


@pytest.mark.parametrize("count,tops,shape", [(1, 1, "0"), (2, 1, "3"), (255, 3, "4"), (2_560, 1, "0"), (2_561, 1, "0"), (5_000, 1, "0"), (5_120, 1, "0"),
                                              (5_121, 1, "0"), (100_003, 0, "-1"), (3_000_001, 0, "1"), (200_000, 64, "3"), (200_000, 64, "1"),
                                              (200_000, 64, "2"), (200_000, 64, "4"), (600_000, 64, "0"), (600_000, 64, "3"), (1_200_000, 64, "0"),
                                              (20_480, 1, "0"), (20_481, 1, "0"), (2_000_000, 64, "2"), (400_000, 70_000, "2"), (8_000_000, 0, "-1")])
@pytest.mark.parametrize("top_bits", [16, 18])
def test_bucket_sort_32bit_keys(count, tops, shape, top_bits, monkeypatch):
    """kernels/bucket_sort.hpp: two global passes over the top 16 key bits (18: nine-bit digits), then every bucket (= value of
    those bits) ordered by its low bits in LDS -- the same stable order as four global passes.  tops = distinct values of the
    top 16 bits (0: all 65 536), i.e. 16-bit buckets of count / tops pairs; a bucket of more than 20 480 pairs is reported,
    not sorted."""
    monkeypatch.setenv("SA_AMD_BUCKET_SHAPE", shape)
    rng = np.random.default_rng(count + tops)
    keys = rng.integers(0, 2**32, count, dtype=np.uint32)
    if tops:
        top = (rng.integers(0, 65536, tops, dtype=np.uint32)[rng.integers(0, tops, count)]) << np.uint32(16)
        keys = (keys & np.uint32(0xFFFF)) | top.astype(np.uint32)
    if count > 10:
        keys[count // 3: count // 3 + count // 10] &= np.uint32(0xFFFF00FF)      # a constant digit inside the buckets
        keys[-(count // 7):] = keys[-1]                                          # equal keys: the order of the values decides
    vals = np.arange(count, dtype=np.uint32)
    rng.shuffle(vals)
    order = np.argsort(keys, kind="stable")
    k2, v2 = keys.copy(), vals.copy()
    pair_of = np.empty(count, dtype=np.uint32)
    pair_of[vals] = keys                               # every value still travels with its own key
    largest = ctypes.c_uint32(0)
    rc = sa.diag_lib().sa_amd_test_bucket_sort32(k2.ctypes.data, v2.ctypes.data, count, top_bits, ctypes.byref(largest))
    true_largest = int(np.bincount(keys >> np.uint32(32 - top_bits), minlength=1 << top_bits).max())
    assert largest.value == true_largest
    if true_largest > 20_480:
        assert rc == 1 and np.array_equal(k2, keys) and np.array_equal(v2, vals)
    else:
        assert rc == 0
        # equal keys may come in any order (only the second in-LDS pass is stable): the keys are sorted, the pairs are the same
        assert np.array_equal(k2, keys[order])
        assert np.array_equal(v2[np.lexsort((v2, k2))], vals[np.lexsort((vals, keys))])
        assert np.array_equal(pair_of[v2], k2)


def _sample_sort_cases(rng):
    """key sets for the sample sort of the 64-bit stage: flat, clustered, with keys that thousands to millions of elements share
    (equality buckets), constant, two-valued, already in order"""
    n = 1_500_000
    flat = rng.integers(0, 1 << 61, size=n, dtype=np.uint64)
    cases = {"flat": flat}
    heavy = flat.copy()
    heavy[rng.random(n) < 0.45] = np.uint64(0x0123456789ABCDE)              # one key on 45 % of the elements
    for k, reps in enumerate((200_000, 20_000, 9_000, 5_000, 3_000, 1_000)):   # and medium-heavy ones around a workgroup's capacity
        heavy[rng.integers(0, n, size=reps)] = np.uint64(0x1000000000000000 + 977 * k)
    cases["heavy keys"] = heavy
    cases["clustered"] = (rng.integers(0, 3000, size=n, dtype=np.uint64) << np.uint64(40)) | rng.integers(0, 1 << 12, size=n, dtype=np.uint64)
    cases["constant"] = np.full(400_000, 7, dtype=np.uint64)
    cases["two values"] = np.where(rng.random(700_000) < 0.3, np.uint64(5), np.uint64(1 << 60)).astype(np.uint64)
    cases["in order"] = np.sort(flat[:900_000])
    cases["few distinct"] = rng.integers(0, 97, size=1_000_003, dtype=np.uint64) * np.uint64((1 << 54) + 12345)      # (< 2^61)
    cases["smallest"] = rng.integers(0, 1 << 50, size=262_144, dtype=np.uint64)
    return cases


@pytest.mark.parametrize("level3", ["lsd", "merge"])
def test_sample_sort_of_the_64_bit_stage(monkeypatch, level3):
    """kernels/sample_sort.hpp through the diagnostic library's hook: two distribution levels over sampled splitters (equality
    buckets for keys that are splitters) + every bucket ordered in LDS (LSD passes, or the merge sort of variant 3) == numpy's sort,
    the values say where each key came from"""
    if level3 == "merge":
        monkeypatch.setenv("SA_AMD_SAMPLE_MERGE", "1")
    D = sa.diag_lib()
    rng = np.random.default_rng(20261005)
    for name, keys in _sample_sort_cases(rng).items():
        for bits in (61, 64):
            k = keys.copy()
            if bits == 64 and name == "flat":
                k |= rng.integers(0, 8, size=k.size, dtype=np.uint64) << np.uint64(61)
            vals = np.zeros(k.size, dtype=np.uint32)
            done = ctypes.c_int32(-1)
            src = k.copy()
            assert D.sa_amd_test_sample_sort64(k.ctypes.data, vals.ctypes.data, k.size, bits, ctypes.byref(done)) == 0, name
            assert done.value == 1, name
            assert np.array_equal(k, np.sort(src)), name
            assert np.array_equal(src[vals], k), name
            assert np.array_equal(np.sort(vals), np.arange(k.size, dtype=np.uint32)), name


def test_sample_sort_as_the_initial_sort_of_a_build(oracle, monkeypatch):
    """diagnostic library, SA_AMD_SAMPLE_SORT=1: whole builds whose 64-bit initial sort is the sample sort (a word-structured text,
    one with a stock sentence that tens of thousands of suffixes share -- equality buckets --, a periodic one) give the oracle's
    arrays; the trace of the dead end's numbers is profiles/r04_sample_sort_64.txt"""
    D = sa.diag_lib()
    monkeypatch.setenv("SA_AMD_SAMPLE_SORT", "1")
    monkeypatch.setenv("SA_AMD_SAMPLE_SORT_MIN_N", str(1 << 17))
    monkeypatch.setenv("SA_AMD_NO_TOP32", "1")
    stock = np.frombuffer(b" the quick brown fox jumps over the lazy dog and runs away;", dtype=np.uint8)
    base = corpus.english(3_000_000, 17)
    with_stock = base.copy()
    for at in np.random.default_rng(5).integers(0, base.size - stock.size, size=40_000):
        with_stock[at:at + stock.size] = stock
    texts = [corpus.english_corpus(2_000_003, 3), with_stock, np.resize(np.frombuffer(b"abcdefghij" * 37 + b"z", dtype=np.uint8), 1_500_000).copy(),
             corpus.english(400_000, 9)]
    for t in texts:
        out = np.zeros(t.size + 1, dtype=np.uint32)
        assert D.sa_amd_saca_u8(t.ctypes.data, out.ctypes.data, t.size) == 0
        assert np.array_equal(out, oracle.sais(t)), t.size


@pytest.mark.parametrize("engine", ["single-pass", "three-kernel"])
def test_radix_sort_constant_and_skewed_digits(engine, monkeypatch):
    if engine == "three-kernel":
        monkeypatch.setenv("SA_AMD_NO_ONESWEEP", "1")
    count = 777_777
    keys = np.full(count, 0x0102030405060708, dtype=np.uint64)
    keys[::1000] = 7
    vals = np.arange(count, dtype=np.uint32)
    order = np.argsort(keys, kind="stable")
    k2, v2 = keys.copy(), vals.copy()
    assert sa.diag_lib().sa_amd_test_sort_pairs(k2.ctypes.data, v2.ctypes.data, count, 0, 64) == 0
    assert np.array_equal(v2, vals[order]) and np.array_equal(k2, keys[order])


@pytest.mark.parametrize("gen,n", [("uniform", 100_003), ("dna", 70_001), ("english", 50_000), ("unary", 5000)])
def test_build_keys(gen, n):
    import pd_model
    text = {"uniform": lambda: corpus.uniform(n, 1), "dna": lambda: corpus.dna(n, 1),
            "english": lambda: corpus.english(n, 1), "unary": lambda: np.full(n, 65, dtype=np.uint8)}[gen]()
    keys = np.zeros(n, dtype=np.uint64)
    bits, k = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)
    assert sa.diag_lib().sa_amd_test_build_keys(text.ctypes.data, n, keys.ctypes.data, bits.ctypes.data, k.ctypes.data) == 0
    exp, ebits, ek = pd_model.pack_keys(text)
    assert (int(bits[0]), int(k[0])) == (ebits, ek)
    assert np.array_equal(keys, exp)


# ---- the path itself --------------------------------------------------------------------------

@pytest.fixture(params=["one-launch small kernel", "general pipeline"])
def small_path(request, monkeypatch):
    """texts of up to 8192 bytes are built by one launch of one workgroup (kernels/small.hpp); SA_AMD_SMALL_MAX=0 sends them
    through the general pipeline, which the small-input tests must keep covering"""
    if request.param == "general pipeline":
        monkeypatch.setenv("SA_AMD_SMALL_MAX", "0")
    return request.param


@pytest.mark.parametrize("text,expected", KNOWN_ANSWERS)
def test_known_answers(text, expected, small_path):
    assert build(text).tolist() == expected


def test_small_kernel_every_length_and_the_adversarial_families(oracle):
    """the one-launch kernel on its whole domain: every n up to 300 and n around the powers of two up to 8192 for several
    alphabets, plus runs, periods, Fibonacci words and text ++ text at its largest sizes (up to 13 doubling rounds in LDS);
    host-pointer and device-pointer entry points"""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rng = np.random.default_rng(99)
    sizes = list(range(0, 301)) + [511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192]
    for n in sizes:
        sigma = int(rng.choice([1, 2, 4, 7, 256]))
        s = (rng.integers(0, sigma, n) + int(rng.integers(0, 257 - sigma))).astype(np.uint8)
        assert np.array_equal(build(s), oracle.sais(s)), (n, sigma)
    fams = {"run": np.full(8192, 255, dtype=np.uint8), "zeros": np.zeros(8191, dtype=np.uint8),
            "abab": np.resize(np.frombuffer(b"ab", dtype=np.uint8), 8192).copy(),
            "period 1000": np.resize(rng.integers(0, 256, 1000, dtype=np.uint8), 8000).copy(),
            "fibonacci": np.frombuffer(fibonacci_word(19)[:8192], dtype=np.uint8).copy(),
            "twice": np.tile(rng.integers(0, 256, 4096, dtype=np.uint8), 2),
            "0xff 0x00": np.resize(np.array([255, 0, 255], dtype=np.uint8), 7001).copy()}
    for name, s in fams.items():
        exp = oracle.sais(s)
        assert np.array_equal(build(s), exp), name
        # the device-pointer entry point takes the same kernel
        n = int(s.size)
        wb = sa.workspace_bytes(n)
        dt, do, dw = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(dt), n + 64) == 0 and hip.hipMalloc(ctypes.byref(do), 4 * (n + 1) + 64) == 0
        assert hip.hipMalloc(ctypes.byref(dw), wb + 512) == 0
        assert hip.hipMemcpy(dt.value, s.ctypes.data, n, 1) == 0
        st = sa.Stats()
        assert sa.lib().sa_amd_saca_device(dt.value, do.value, n, (dw.value + 255) & ~255, wb, None, ctypes.byref(st)) == 0
        out = np.zeros(n + 1, dtype=np.uint32)
        assert hip.hipMemcpy(out.ctypes.data, do.value, 4 * (n + 1), 2) == 0
        for ptr in (dt, do, dw):
            hip.hipFree(ptr)
        assert np.array_equal(out, exp), name
        assert st.sort_passes == 0                                  # (no radix pass ran: it was the small kernel)


def test_golden_fixtures(small_path):
    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    for name in manifest:
        text = open(os.path.join(GOLDEN, name + ".text"), "rb").read()
        exp = np.fromfile(os.path.join(GOLDEN, name + ".sa.u32le"), dtype="<u4")
        assert np.array_equal(build(text), exp), name


@pytest.mark.parametrize("name", sorted(adversarial_cases()))
def test_adversarial(oracle, name, small_path):
    s = adversarial_cases()[name]
    assert np.array_equal(build(s), oracle.sais(s))


def test_conversion_correctness(oracle, small_path):
    """reference src/tests.rs:13-17: random bytes, n in [0, 4096): new().into_parts() must pass
    from_parts (check_integrity) -- and here also equal the oracle bit for bit."""
    rng = np.random.default_rng(20240)
    for it in range(150):
        n = int(rng.integers(0, 4096))
        s = rng.integers(0, 256, n, dtype=np.uint8)
        text, arr = sa.SuffixArray(s).into_parts()
        assert sa.SuffixArray.from_parts(text, arr) is not None
        assert oracle.check_integrity(s, arr) == 1
        assert np.array_equal(arr, oracle.sais(s))


def test_small_alphabets_and_ragged_lengths(oracle, small_path):
    rng = np.random.default_rng(7)
    for it in range(120):
        n = int(rng.integers(0, 20000))
        sigma = int(rng.choice([1, 2, 3, 4, 5, 9, 17, 33, 65, 129, 256]))
        base = int(rng.integers(0, 257 - sigma))
        s = (rng.integers(0, sigma, n) + base).astype(np.uint8)
        assert np.array_equal(build(s), oracle.sais(s)), (n, sigma, base)


def test_set_reuses_buffer_with_stale_contents(oracle, small_path):
    """reference src/sa.rs:30-33: set() re-runs construction into a previously used buffer"""
    obj = sa.SuffixArray(b"mississippi")
    obj.set(b"banana")
    assert obj.into_parts()[1].tolist() == [6, 5, 3, 1, 0, 4, 2]
    obj.set(corpus.english(5000, 9))
    assert np.array_equal(obj.into_parts()[1], oracle.sais(corpus.english(5000, 9)))


def test_divsufsort_entry_point(oracle):
    """the C engine's signature (reference src/saca.rs:14): n int32 entries, no sentinel slot"""
    s = corpus.uniform(100_000, 5)
    out = np.full(s.size, -1, dtype=np.int32)
    sa.divsufsort(s, out)
    assert np.array_equal(out.astype(np.uint32), oracle.sais(s)[1:])


@pytest.mark.parametrize("gen,n,seed", [("uniform", 1 << 20, 2), ("english", 1 << 20, 3), ("dna", 1 << 20, 4),
                                        ("dna_repeats", 1 << 20, 5), ("english", (4 << 20) + 1, 6),
                                        ("uniform", (16 << 20) - 1, 7), ("dna", 16 << 20, 8)])
def test_medium_sizes_bit_exact(oracle, gen, n, seed):
    text = getattr(corpus, gen)(n, seed)
    assert np.array_equal(build(text), oracle.sais(text))


@pytest.mark.parametrize("n", [(16 << 20) - 1, 16 << 20, (16 << 20) + 1, (20 << 20) + 4099])
def test_staged_download_sizes(oracle, monkeypatch, n):
    """the suffix array of a large text travels back through pinned staging chunks split over helper threads: sizes around
    the chunk boundaries (a last chunk of 4 bytes, a ragged one), few and many helpers, and the plain-copy route"""
    text = corpus.uniform(n, 12)
    exp = oracle.sais(text)
    for threads in ("8", "3", "1", "0"):
        monkeypatch.setenv("SA_AMD_COPY_THREADS", threads)
        monkeypatch.setenv("SA_AMD_STAGED_MIN_BYTES", "0" if threads == "3" else str(64 << 20))
        out = np.full(n + 1, 0xFFFFFFFF, dtype=np.uint32)
        sa.saca(text, out)
        assert np.array_equal(out, exp), (n, threads)
        assert sa.last_host_timing()["staged_threads"] == (int(threads) if (n + 1) * 4 >= (64 << 20) or threads == "3" else 0)
    div = np.full(n, -1, dtype=np.int32)
    sa.divsufsort(text, div)                      # n entries, no sentinel slot: nothing may be written past them
    assert np.array_equal(div.astype(np.uint32), exp[1:])
    # the chunks by the copy KERNEL instead of the copy engine (what a device block that the engine reads at half rate gets):
    # with the sentinel (16-byte aligned source) and without (the source starts one element in), and the engine forced
    for knob in ("SA_AMD_KERNEL_D2H_ALWAYS", "SA_AMD_NO_KERNEL_D2H"):
        monkeypatch.setenv(knob, "1")
        monkeypatch.setenv("SA_AMD_COPY_THREADS", "5")
        out = np.full(n + 1, 0xFFFFFFFF, dtype=np.uint32)
        sa.saca(text, out)
        assert np.array_equal(out, exp), (n, knob)
        div = np.full(n, -1, dtype=np.int32)
        sa.divsufsort(text, div)
        assert np.array_equal(div.astype(np.uint32), exp[1:]), (n, knob)
        monkeypatch.delenv(knob)


@pytest.mark.parametrize("gen,n,seed,div", [("english_corpus", 3_000_001, 3, "1"), ("english_corpus", (4 << 20) + 77, 4, "3"), ("dna_repeats", 2_500_000, 5, "1"),
                                            ("periodic", 1_200_007, 1, "1"), ("twice", 2_000_000, 9, "1")])
@pytest.mark.parametrize("sentinel", [True, False])
def test_early_download_patches_the_slots_that_were_still_tied(oracle, monkeypatch, gen, n, seed, div, sentinel):
    """the front of the array starts its way to the host while refinement rounds still run (host/host_path.hpp, EarlyPull): the
    entries that were tied when the copy began arrive stale and are patched from the values sent behind.  Forced here at small
    sizes: 64 KiB chunks, the copy starts as soon as at most n / div suffixes are tied, and the build WAITS until eight chunks
    have been copied, so every run has early chunks with stale entries in them -- through both entry points (with and without
    the sentinel entry in front)"""
    monkeypatch.setenv("SA_AMD_STAGED_MIN_BYTES", "0")
    monkeypatch.setenv("SA_AMD_EARLY_MIN_BYTES", "0")
    monkeypatch.setenv("SA_AMD_EARLY_CHUNK_BYTES", "65536")
    monkeypatch.setenv("SA_AMD_EARLY_DIV", div)
    monkeypatch.setenv("SA_AMD_EARLY_WAIT_CHUNKS", "8")
    if gen == "periodic":
        t = np.resize(np.frombuffer(b"abcabcabd" * 7 + b"x", dtype=np.uint8), n).copy()
    elif gen == "twice":
        t = np.concatenate([corpus.english(n // 2, seed)] * 2)
    else:
        t = getattr(corpus, gen)(n, seed)
    exp = oracle.sais(t)
    for it in range(3):
        if sentinel:
            out = np.full(t.size + 1, 0xDEADBEEF, dtype=np.uint32)
            sa.saca(t, out)
            assert np.array_equal(out, exp), (gen, it)
        else:
            out = np.full(t.size + 1, 0xDEADBEEF, dtype=np.uint32)          # (n entries + one that must stay untouched)
            assert sa.lib().sa_amd_divsufsort(t.ctypes.data, out.ctypes.data, t.size) == 0
            assert np.array_equal(out[:-1], exp[1:]) and out[-1] == 0xDEADBEEF, (gen, it)
        ht = sa.last_host_timing()
        assert 0 < ht["early_fraction"] <= 0.76, ht              # (some chunks travelled early, never more than three quarters)
        assert sa.last_stats()["rounds"] >= 1


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    return hip


def _free_hbm(hip):
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return int(f.value)


def test_reduced_memory_route_when_the_device_is_nearly_full(oracle, monkeypatch):
    """another tenant holds nearly all of the HBM: the full workspace (53.6 n bytes) cannot be had, and instead of SA_AMD_ENOMEM
    the build keeps the text, the array and the most-used slabs on the device and puts the rest of the workspace into pinned
    host memory (host/host_path.hpp) -- the oracle's array either way; SA_AMD_NO_REDUCED=1 restores the error"""
    hip = _hip()
    t = corpus.english_corpus(40 << 20, 6)
    exp = oracle.sais(t)
    sa.lib().sa_amd_release_cache()
    need = int(t.size) * 5 + sa.workspace_bytes(int(t.size))
    free = _free_hbm(hip)
    hog = ctypes.c_void_p()
    leave = int(need * 0.7)                                      # the device part gets ~70 % of what a build asks for, less the margin
    assert hip.hipMalloc(ctypes.byref(hog), free - leave) == 0
    try:
        out = np.zeros(t.size + 1, dtype=np.uint32)
        monkeypatch.setenv("SA_AMD_NO_REDUCED", "1")
        with pytest.raises(sa.SuffixArrayError) as e:
            sa.saca(t, out)
        assert e.value.code == -2
        monkeypatch.delenv("SA_AMD_NO_REDUCED")
        sa.saca(t, out)
        ht = sa.last_host_timing()
        assert ht["workspace_bytes_in_host_memory"] > 0, ht
        assert np.array_equal(out, exp)
    finally:
        hip.hipFree(hog)
        sa.lib().sa_amd_release_cache()
    out2 = np.zeros(t.size + 1, dtype=np.uint32)
    sa.saca(t, out2)                                             # and with the memory back: the ordinary route
    assert sa.last_host_timing()["workspace_bytes_in_host_memory"] == 0 and np.array_equal(out2, exp)


def test_pool_gives_back_what_recent_builds_do_not_need(oracle, monkeypatch):
    """the device-block pool keeps at most twice what the largest of a device's builds of the last SA_AMD_CACHE_IDLE_MS asked for,
    and frees blocks nobody has asked for in that time at the next call: a process that built one large array and goes on with
    small ones does not sit on the large block (3.7 GiB here) for its lifetime -- but a dozen small builds between two large ones
    do not cost the large block either (the block allocated in its place would download at half the rate: host_path.hpp)"""
    hip = _hip()
    sa.lib().sa_amd_release_cache()
    big, small = corpus.uniform(64 << 20, 3), corpus.uniform(1 << 20, 4)
    out_b, out_s = np.zeros(big.size + 1, dtype=np.uint32), np.zeros(small.size + 1, dtype=np.uint32)
    sa.saca(big, out_b)
    held = _free_hbm(hip)
    for _ in range(12):
        sa.saca(small, out_s)
    assert _free_hbm(hip) < held + (1 << 30)                     # the 3.7 GiB block is still the pool's
    assert np.array_equal(out_s, oracle.sais(small))
    monkeypatch.setenv("SA_AMD_CACHE_IDLE_MS", "20")
    import time
    time.sleep(0.05)
    sa.saca(small, out_s)                                        # (touches the pool: the idle big block is freed)
    assert _free_hbm(hip) > held + (3 << 30)
    sa.saca(big, out_b)
    held = _free_hbm(hip)
    time.sleep(0.05)
    for _ in range(3):
        sa.saca(small, out_s)                                    # the large request has left the window: nothing large is kept
    assert _free_hbm(hip) > held + (3 << 30)
    sa.lib().sa_amd_release_cache()


def test_batch_entry_point(oracle):
    texts = [corpus.uniform(50_000, 50 + i) for i in range(3)] + [np.zeros(0, dtype=np.uint8), corpus.dna(30_000, 1)]
    outs = sa.saca_batch(texts)
    for t, o in zip(texts, outs):
        assert np.array_equal(o, oracle.sais(t))


def _small_texts(rng, count, max_n):
    out = []
    for i in range(count):
        n = int(rng.choice([rng.integers(1, 40), rng.integers(1, 700), rng.integers(1, max_n + 1)]))
        kind = i % 5
        if kind == 0: t = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1: t = corpus.english(n, int(rng.integers(0, 1 << 30)))
        elif kind == 2: t = np.full(n, int(rng.integers(0, 256)), dtype=np.uint8)
        elif kind == 3: t = np.resize(rng.integers(0, 4, int(rng.integers(1, 9)), dtype=np.uint8), n).astype(np.uint8)
        else: t = rng.integers(0, 3, n, dtype=np.uint8)
        out.append(np.ascontiguousarray(t, dtype=np.uint8))
    return out


def test_batch_of_small_texts_is_one_launch_per_chunk(oracle, monkeypatch):
    """sa_amd_saca_batch builds the texts of up to SA_AMD_SMALL_MAX bytes of a device together (k_small_sa_batch, one workgroup
    per text; the reference's own test domain is n < 4096, src/tests.rs:13-17): every length class, runs / periods / small
    alphabets, the sizes around the kernel's limits, empty and larger texts in the same call (those take the single-text path),
    a batch that needs several chunks, one small text alone, and the same arrays with the batched path off"""
    rng = np.random.default_rng(20260)
    texts = _small_texts(rng, 700, 8192)
    texts += [rng.integers(0, 256, n, dtype=np.uint8) for n in (1, 2, 4095, 4096, 4097, 8191, 8192)]
    texts += [np.zeros(0, dtype=np.uint8), corpus.english(20_000, 3), corpus.uniform(8193, 4), np.zeros(0, dtype=np.uint8)]
    order = rng.permutation(len(texts))
    texts = [texts[i] for i in order]
    exp = [oracle.sais(t) for t in texts]
    outs = sa.saca_batch(texts)
    for k, (o, e) in enumerate(zip(outs, exp)):
        assert np.array_equal(o, e), (k, texts[k].size)
    # several chunks: 2 600 texts of 8 192 bytes are 104 MiB of texts + arrays (a chunk holds 96 MiB)
    big = [rng.integers(0, 256, 8192, dtype=np.uint8) for _ in range(2600)]
    outs = sa.saca_batch(big)
    for k in range(0, len(big), 37):
        assert np.array_equal(outs[k], oracle.sais(big[k])), k
    assert all(o[0] == 8192 and np.array_equal(np.sort(o[1:]), np.arange(8192, dtype=np.uint32)) for o in outs[:: 11])
    assert np.array_equal(outs[-1], oracle.sais(big[-1]))
    # one small text among large ones, and a lower / disabled small-text limit: same arrays
    outs = sa.saca_batch([corpus.uniform(30_000, 1), texts[0], corpus.dna(40_000, 2)])
    assert np.array_equal(outs[1], exp[0])
    for lim in ("0", "100"):
        monkeypatch.setenv("SA_AMD_SMALL_MAX", lim)
        outs = sa.saca_batch(texts[:60])
        for k, (o, e) in enumerate(zip(outs, exp[:60])):
            assert np.array_equal(o, e), (lim, k, texts[k].size)


def test_degenerate_large_runs(oracle, monkeypatch):
    """runs and periodic texts: one or a few huge groups in every round -- whole-list global sorts keyed by group index, radix
    passes whose digit is constant skipped (both forced on at this size), the local pass given up and probed again"""
    texts = (np.full(1 << 20, 0, dtype=np.uint8), np.tile(np.array([1, 2], dtype=np.uint8), 1 << 19),
             np.full((1 << 20) + 3, 255, dtype=np.uint8), np.resize(np.arange(1, 98, dtype=np.uint8), 700_001).copy())
    exp = [oracle.sais(s) for s in texts]
    for env in ({}, {"SA_AMD_RUN_SKIP_MIN": "1", "SA_AMD_DENSE_REKEY_MIN": "1"}, {"SA_AMD_RUN_SKIP_MIN": "1", "SA_AMD_NO_LOCAL_SORT": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for s, e in zip(texts, exp):
            assert np.array_equal(build(s), e), env
        for k in env:
            monkeypatch.delenv(k)


def test_text_of_one_byte_value_takes_the_closed_form(oracle, monkeypatch):
    """a zero-filled file (any single byte value repeated) has a closed form -- the shorter suffix is a prefix of the longer, SA[i] =
    n - 1 - i -- and the pipeline takes it right behind the byte histogram (k_fill_descending: no sort, no rounds; 214 ms -> 0.3 ms
    at 256 MiB).  Every other test of this suite runs with SA_AMD_NO_UNARY_SHORTCUT=1 (conftest.py), so that one-byte texts keep
    exercising the general path's giant-group machinery; here the knob is off, through both entry points and the device one"""
    monkeypatch.delenv("SA_AMD_NO_UNARY_SHORTCUT", raising=False)
    for n in (8193, 100_000, (1 << 22) + 5):
        for b in (0, 65, 255):
            t = np.full(n, b, dtype=np.uint8)
            got = build(t)
            assert np.array_equal(got, np.arange(n, -1, -1, dtype=np.uint32)) and got[0] == n
            st = sa.last_stats()
            assert st["rounds"] == 0 and st["sort_passes"] == 0 and st["sigma"] == 1
    t = np.full(300_001, 9, dtype=np.uint8)
    assert np.array_equal(build(t), oracle.sais(t))
    div = np.full(t.size, 0xFFFFFFFF, dtype=np.uint32)
    assert sa.lib().sa_amd_divsufsort(t.ctypes.data, div.ctypes.data, t.size) == 0 and np.array_equal(div, oracle.sais(t)[1:])
    t[-1] = 8                                                    # (two byte values: not this route)
    assert np.array_equal(build(t), oracle.sais(t)) and sa.last_stats()["sort_passes"] > 0


def test_dense_rounds_keep_the_ranks_of_the_last_subgroup(oracle, monkeypatch):
    """dense doubling rounds label a group by its last slot and do not rewrite the ranks of a parent's last subgroup
    (k_rr_apply, TAIL): groups that shed members at the front (run to the end of the text), at the back (run before a larger
    symbol), on both sides, groups longer than a re-rank tile and a wave; direct and binned rank stores, local pass on / off"""
    rng = np.random.default_rng(5)
    a = np.full(300_000, 7, dtype=np.uint8)
    texts = [
        np.concatenate([a, [9]]).astype(np.uint8),                              # a^k b: the longest run is the smallest suffix
        np.concatenate([a, [3]]).astype(np.uint8),                              # a^k followed by a smaller symbol
        np.concatenate([a[:100_000], [9], a[:150_000], [3], a[:70_000]]).astype(np.uint8),
        np.concatenate([np.tile(np.array([5, 6, 7], dtype=np.uint8), 90_000), [8], np.tile(np.array([5, 6, 7], dtype=np.uint8), 60_000), [1]]).astype(np.uint8),
        np.concatenate([rng.integers(0, 3, 5000, dtype=np.uint8)] * 40 + [rng.integers(0, 3, 777, dtype=np.uint8)]),   # one block 40 times
    ]
    # (+ texts whose FIRST ranks matter: word-structured with copied passages, groups that never split before the last rounds)
    texts += [corpus.english_corpus(1 << 20, 4, 2000, 0.3), np.concatenate([corpus.english(150_000, 8)] * 2), _planted(400_000, 11, 3),
              corpus.dna_repeats(250_000, 5)]
    exp = [oracle.sais(s) for s in texts]
    # the rank set-up of the dense route writes tail ranks too (k_rr_apply FTAIL; SA_AMD_NO_FIRST_TAIL=1: head ranks, as before)
    for first in ({}, {"SA_AMD_NO_FIRST_TAIL": "1"}):
        for env in ({"SA_AMD_FORCE_DENSE": "1"}, {"SA_AMD_FORCE_DENSE": "1", "SA_AMD_BINNED_ISA_ALWAYS": "1"},
                    {"SA_AMD_FORCE_DENSE": "1", "SA_AMD_BINNED_ISA_ALWAYS": "1", "SA_AMD_NO_LOCAL_SORT": "1", "SA_AMD_DENSE_REKEY_MIN": "1"},
                    {"SA_AMD_FORCE_DENSE": "1", "SA_AMD_NO_TOP32": "1", "SA_AMD_BINNED_MIN": "1"},
                    {"SA_AMD_FORCE_DENSE": "1", "SA_AMD_NO_TOP32": "1", "SA_AMD_NO_BINNED_ISA": "1", "SA_AMD_CHASE": "1"},
                    {"SA_AMD_FORCE_DENSE": "1", "SA_AMD_DENSE_REKEY_MIN": "1", "SA_AMD_GROUP_CAP": "16"}, {}):
            env = dict(env); env.update(first)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            for s, e in zip(texts, exp):
                assert np.array_equal(build(s), e), (env, s.size)
            for k in env:
                monkeypatch.delenv(k)


@pytest.mark.parametrize("g", ["0", "2", "3", "5"])
def test_gram_keys(oracle, monkeypatch, g):
    """initial keys made of dense ranks of the g-grams that occur in the text (step 2c; by default only for texts of at
    least 4 MiB -- the threshold is lowered here): word-structured texts, small alphabets where the gram form wins or
    loses, texts shorter than a key, grams that run past the end of the text; then every refinement regime on top of them"""
    monkeypatch.setenv("SA_AMD_GRAM_MIN_N", "1")
    monkeypatch.setenv("SA_AMD_GRAM_G", g)
    texts = [corpus.english(300_001, 3), corpus.english_corpus(1 << 20, 4, 2000, 0.3), corpus.sigma(200_000, 7, 5, 97),
             corpus.dna_repeats(150_000, 5), np.resize(np.frombuffer(b"abcab", dtype=np.uint8), 50_001).copy(),
             np.frombuffer(b"to be or not to be that is the question", dtype=np.uint8).copy(), corpus.english(37, 1),
             _planted(400_000, 11, 3), np.full(5000, 9, dtype=np.uint8)]
    exp = [oracle.sais(t) for t in texts]
    for env in ({}, {"SA_AMD_NO_TOP32": "1"}, {"SA_AMD_NO_TOP32": "1", "SA_AMD_FORCE_DENSE": "1"},
                {"SA_AMD_NO_TOP32": "1", "SA_AMD_NO_REPEAT_PROBE": "1", "SA_AMD_SPARSE_DIV": "4"},
                {"SA_AMD_NO_TOP32": "1", "SA_AMD_FUSED64": "1"}, {"SA_AMD_NO_TOP32": "1", "SA_AMD_NO_LOCAL_SORT": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for t, e in zip(texts, exp):
            assert np.array_equal(build(t), e), (env, t.size)
        for k in env:
            monkeypatch.delenv(k)
    st = None
    monkeypatch.setenv("SA_AMD_NO_TOP32", "1")
    build(texts[0]); st = sa.last_stats()
    if g in ("0", "3"):
        assert st["symbols_per_key"] > 10, st          # english-like sigma ~ 56: 10 or 11 symbols in the plain form


@pytest.mark.parametrize("tail", ["0", "1", "2", "8"])
def test_gram_keys_with_plain_tail(oracle, monkeypatch, tail):
    """the key bits a whole further gram does not fit into carry plain symbols behind the gram ranks (SA_AMD_GRAM_TAIL caps
    how many; 0 = the pure gram form): same arrays whatever the cap, keys that end past the text, the sparse route's
    re-computation of a key from the text (text_key), and more symbols per key than the pure form when a tail is allowed"""
    monkeypatch.setenv("SA_AMD_GRAM_MIN_N", "1")
    monkeypatch.setenv("SA_AMD_NO_TOP32", "1")
    monkeypatch.setenv("SA_AMD_GRAM_TAIL", tail)
    texts = [corpus.english(300_001, 3), corpus.english_corpus(1 << 20, 4, 2000, 0.3), corpus.sigma(200_000, 7, 5, 97),
             np.frombuffer(b"to be or not to be that is the question", dtype=np.uint8).copy(), corpus.english(37, 1),
             _planted(400_000, 11, 3), np.resize(np.frombuffer(b"abcab", dtype=np.uint8), 50_001).copy()]
    exp = [oracle.sais(t) for t in texts]
    syms = {}
    for g in ("0", "2", "3"):
        monkeypatch.setenv("SA_AMD_GRAM_G", g)
        for env in ({}, {"SA_AMD_FORCE_DENSE": "1"}, {"SA_AMD_NO_REPEAT_PROBE": "1", "SA_AMD_SPARSE_DIV": "4"},
                    {"SA_AMD_NO_REPEAT_PROBE": "1", "SA_AMD_NO_TEXT_ROUNDS": "1", "SA_AMD_SPARSE_DIV": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            for t, e in zip(texts, exp):
                assert np.array_equal(build(t), e), (g, env, t.size)
            for k in env:
                monkeypatch.delenv(k)
        build(texts[0]); syms[g] = sa.last_stats()["symbols_per_key"]
    if tail == "0":
        assert syms["3"] % 3 == 0 and syms["2"] % 2 == 0, syms      # whole grams only
    else:
        assert syms["0"] >= 11, syms                                  # english-like sigma ~ 56: 10 or 11 symbols in the plain form


def test_giant_groups_are_split_around_their_majority_key(oracle, monkeypatch):
    """whole-list rounds over a few giant groups (runs, periodic texts, a block repeated): the members whose key differs from
    their group's majority are extracted, sorted on their own and merged back (k_split_*), the rest only shifts -- thresholds
    lowered so that the path runs at this size, with many and few groups, minorities on one side, on both sides, none at
    all, and the fallback to the radix sort when the minority is not small"""
    rng = np.random.default_rng(9)
    a = np.full(400_000, 7, dtype=np.uint8)
    blk = rng.integers(0, 256, 1000, dtype=np.uint8)
    texts = [a, np.tile(np.array([1, 2], dtype=np.uint8), 250_000), np.resize(blk, 700_001).copy(),
             np.concatenate([a[:100_000], [9], a[:150_000], [3], a[:70_000]]).astype(np.uint8),
             np.concatenate([np.resize(blk[:37], 300_000), rng.integers(0, 4, 50_000, dtype=np.uint8), np.resize(blk[:37], 200_000)]).astype(np.uint8),
             np.frombuffer(bytes(fibonacci_word(27)[:600_000]), dtype=np.uint8).copy()]
    exp = [oracle.sais(s) for s in texts]
    base = {"SA_AMD_SPLIT_MIN": "1", "SA_AMD_DENSE_REKEY_MIN": "1", "SA_AMD_FORCE_DENSE": "1"}
    for extra in ({"SA_AMD_SPLIT_GROUP_MIN": "1000"}, {"SA_AMD_SPLIT_GROUP_MIN": "1", "SA_AMD_NO_LOCAL_SORT": "1"},
                  {"SA_AMD_SPLIT_GROUP_MIN": "64", "SA_AMD_NO_LOCAL_SORT": "1", "SA_AMD_BINNED_ISA_ALWAYS": "1"},
                  {"SA_AMD_SPLIT_GROUP_MIN": "1", "SA_AMD_GROUP_CAP": "8"}, {"SA_AMD_NO_SPLIT": "1", "SA_AMD_NO_LOCAL_SORT": "1"}):
        env = dict(base); env.update(extra)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for s, e in zip(texts, exp):
            assert np.array_equal(build(s), e), (env, s.size)
        for k in env:
            monkeypatch.delenv(k)


def _planted(n, seed, copies):
    rng = np.random.default_rng(seed)
    s = corpus.uniform(n, seed).copy()
    for _ in range(copies):
        ln = int(rng.integers(20, max(21, min(3000, n // (160 * copies)))))
        src = int(rng.integers(0, n - ln)); dst = int(rng.integers(0, n - ln))
        s[dst:dst + ln] = s[src:src + ln]
    return s


@pytest.mark.parametrize("n,copies", [(300_000, 1), (1 << 20, 5), (3_000_001, 40)])
def test_sparse_refinement_mode(oracle, monkeypatch, n, copies):
    """few tied suffixes after the initial sort: ranks come from a binary search in the sorted keys
    (no ISA); must agree with the dense mode and the oracle"""
    s = _planted(n, n, copies)
    exp = oracle.sais(s)
    assert np.array_equal(build(s), exp)
    st = sa.last_stats()
    assert st["sparse_mode"] == 1 and st["rounds"] >= 1 and 0 < st["unresolved_after_initial"] <= n // 64
    monkeypatch.setenv("SA_AMD_FORCE_DENSE", "1")
    assert np.array_equal(build(s), exp)
    assert sa.last_stats()["sparse_mode"] == 0


def test_sparse_mode_with_tail_and_runs(oracle):
    # tied suffixes that run into the end of the text, and a long run inside otherwise random bytes
    s = corpus.uniform(400_000, 9).copy()
    s[-50:] = s[1000:1050]
    s[200_000:200_700] = 7
    assert np.array_equal(build(s), oracle.sais(s))
    assert sa.last_stats()["sparse_mode"] == 1


@pytest.mark.parametrize("gen,n,seed", [("english", 300_000, 3), ("dna_repeats", 1 << 20, 5), ("periodic", 200_001, 1)])
def test_binned_isa_update_path(oracle, monkeypatch, gen, n, seed):
    """the (suffix, rank) pairs + one radix pass + windowed scatter form of the ISA update (used for
    large texts) forced on at small sizes; must give the same array as the direct scatter"""
    text = np.resize(np.frombuffer(b"abcab", dtype=np.uint8), n) if gen == "periodic" else getattr(corpus, gen)(n, seed)
    exp = oracle.sais(text)
    monkeypatch.setenv("SA_AMD_BINNED_ISA_ALWAYS", "1")
    monkeypatch.setenv("SA_AMD_FORCE_DENSE", "1")
    for levels in ("1", "2"):                         # one radix pass + plain scatter / two passes + windows assembled in LDS
        monkeypatch.setenv("SA_AMD_SCATTER_LEVELS", levels)
        assert np.array_equal(build(text), exp), levels
        assert sa.last_stats()["rounds"] >= 1
        monkeypatch.setenv("SA_AMD_MAX_TEXT_ROUNDS", "1")      # the late ISA build (after a text-keyed round) too
        monkeypatch.delenv("SA_AMD_FORCE_DENSE")
        monkeypatch.setenv("SA_AMD_SPARSE_DIV", "1000000000")
        assert np.array_equal(build(text), exp), levels
        monkeypatch.delenv("SA_AMD_MAX_TEXT_ROUNDS"); monkeypatch.delenv("SA_AMD_SPARSE_DIV")
        monkeypatch.setenv("SA_AMD_FORCE_DENSE", "1")
    monkeypatch.delenv("SA_AMD_BINNED_ISA_ALWAYS")
    monkeypatch.setenv("SA_AMD_NO_BINNED_ISA", "1")
    assert np.array_equal(build(text), exp)


def test_binned_isa_paths_on_tiny_texts(oracle, monkeypatch):
    """both binned ISA routes forced on texts far smaller than a window (found by tools/stress.py: with no key bits above
    the window the two-pass route has nothing to sort and must hand over to the one-pass form)"""
    rng = np.random.default_rng(5)
    monkeypatch.setenv("SA_AMD_BINNED_ISA_ALWAYS", "1")
    monkeypatch.setenv("SA_AMD_BINNED_MIN", "1")
    for levels in ("1", "2"):
        monkeypatch.setenv("SA_AMD_SCATTER_LEVELS", levels)
        for dense in (True, False):
            if dense:
                monkeypatch.setenv("SA_AMD_FORCE_DENSE", "1")
            else:
                monkeypatch.delenv("SA_AMD_FORCE_DENSE", raising=False)
                monkeypatch.setenv("SA_AMD_SPARSE_DIV", "1000000000")
            for n in (2, 3, 21, 51, 222, 226, 1023, 1024, 1025, 2049, 5000):
                for sig in (1, 3, 10):
                    s = (rng.integers(0, sig, n) + 60).astype(np.uint8)
                    s[n // 2:] = s[: n - n // 2]                       # a long repeat: several doubling rounds
                    assert np.array_equal(build(s), oracle.sais(s)), (levels, dense, n, sig)
        monkeypatch.delenv("SA_AMD_SPARSE_DIV", raising=False)


def test_concurrent_callers(oracle):
    """SuffixArray is Send + Sync in the reference (src/sa.rs:15-19): arrays may be built from many
    threads at once; every thread owns its stream and device block here"""
    import threading
    texts = [corpus.english(200_000 + 1111 * i, 30 + i) for i in range(6)]
    outs = [None] * len(texts)

    def work(i):
        for _ in range(3):
            outs[i] = build(texts[i])

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(texts))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for t_, o in zip(texts, outs):
        assert np.array_equal(o, oracle.sais(t_))


def test_max_length_text():
    """n = MAX_LENGTH = i32::MAX (reference src/saca.rs:6,10): index arithmetic at the top of the range.
    Checked with the HIP integrity check (reference src/sa.rs:72-84) and direct comparisons of
    sampled neighbours."""
    n = sa.MAX_LENGTH
    text = corpus.uniform(n, 77)
    arr = np.empty(n + 1, dtype=np.uint32)
    sa.saca(text, arr)
    assert arr[0] == n
    assert sa.check_integrity(text, arr) is True
    rng = np.random.default_rng(1)
    for i in rng.integers(1, n, 2000):
        a, b = int(arr[i]), int(arr[i + 1])
        assert text[a:a + 64].tobytes() < text[b:b + 64].tobytes() or text[a:a + 64].tobytes() == text[b:b + 64].tobytes()
    sa.lib().sa_amd_release_cache()


@pytest.mark.parametrize("gen,n,seed", [("english", 400_000, 3), ("dna_repeats", 1 << 20, 5), ("periodic", 150_001, 1),
                                        ("uniform", 300_000, 2), ("tail_zeros", 90_000, 4)])
@pytest.mark.parametrize("div", ["1000000000", "1", None])
def test_refinement_regimes(oracle, monkeypatch, gen, n, seed, div):
    """text-keyed rounds followed by the dense ISA rebuild (div = 1e9: never sparse), sparse doubling
    from the start (div = 1), and the default switch-over; all must give the oracle's array"""
    if gen == "periodic":
        text = np.resize(np.frombuffer(b"abcab", dtype=np.uint8), n)
    elif gen == "tail_zeros":
        text = np.concatenate([corpus.english(n - 5000, seed), np.zeros(5000, dtype=np.uint8)])
    else:
        text = getattr(corpus, gen)(n, seed)
    if div is not None:
        monkeypatch.setenv("SA_AMD_SPARSE_DIV", div)
    assert np.array_equal(build(text), oracle.sais(text))
    st = sa.last_stats()
    if div == "1000000000" and st["unresolved_after_initial"] > 0:
        assert st["text_rounds"] >= 1 and st["sparse_mode"] == 0
    if div == "1" and st["unresolved_after_initial"] > 0:
        assert st["text_rounds"] == 0 and st["sparse_mode"] == 1


@pytest.mark.parametrize("gen,n,seed", [("uniform", 500_000, 2), ("dna", 1 << 20, 4), ("english", 300_000, 3),
                                        ("dna_repeats", 400_000, 5), ("periodic", 100_001, 1), ("sigma2", 250_000, 7)])
def test_two_stage_initial_sort(oracle, monkeypatch, gen, n, seed):
    """initial sort on the top 32 key bits only, ties finished on the low bits (normally chosen by the
    entropy probe for large high-entropy texts; forced here), with and without the in-LDS group sort"""
    if gen == "periodic":
        text = np.resize(np.frombuffer(b"abcab", dtype=np.uint8), n)
    elif gen == "sigma2":
        text = corpus.sigma(n, seed, 2, 120)
    else:
        text = getattr(corpus, gen)(n, seed)
    exp = oracle.sais(text)
    monkeypatch.setenv("SA_AMD_FORCE_TOP32", "1")
    assert np.array_equal(build(text), exp)
    assert sa.last_stats()["top32_first"] == 1
    monkeypatch.setenv("SA_AMD_NO_LOCAL_SORT", "1")      # the probe path is off without the local sort
    assert np.array_equal(build(text), exp)
    assert sa.last_stats()["top32_first"] == 0
    monkeypatch.setenv("SA_AMD_DENSE_REKEY_MIN", "1")    # every round: the whole list through the global sort, keyed by group index
    assert np.array_equal(build(text), exp)
    monkeypatch.setenv("SA_AMD_FORCE_DENSE", "1")
    assert np.array_equal(build(text), exp)


@pytest.mark.parametrize("gen,n,seed", [("uniform", 500_000, 2), ("uniform", 3_000_001, 3), ("dna", 1 << 20, 4), ("dna", 1_000_003, 6), ("dna", (1 << 22) + 5, 8), ("english", 300_000, 3),
                                        ("dna_repeats", 400_000, 5), ("periodic", 100_001, 1), ("sigma2", 250_000, 7), ("sigma200", 900_000, 9)])
@pytest.mark.parametrize("shape,bits", [("-1", "0"), ("3", "16"), ("4", "0"), ("-1", "18"), ("2", "18")])
def test_bucket_route_of_the_two_stage_initial_sort(oracle, monkeypatch, gen, n, seed, shape, bits):
    """the 32-bit first stage as two global passes + the in-LDS bucket sort (taken from 32 Mi suffixes on; here from 1), on texts
    whose buckets fit a workgroup and on texts where they do not (a period, two symbols: the keys are rebuilt and the four
    global passes run) -- the same array either way"""
    if gen == "periodic":
        text = np.resize(np.frombuffer(b"abcab", dtype=np.uint8), n)
    elif gen == "sigma2":
        text = corpus.sigma(n, seed, 2, 120)
    elif gen == "sigma200":
        text = corpus.sigma(n, seed, 200, 1)
    else:
        text = getattr(corpus, gen)(n, seed)
    exp = oracle.sais(text)
    monkeypatch.setenv("SA_AMD_FORCE_TOP32", "1")
    monkeypatch.setenv("SA_AMD_BUCKET_MIN_N", "1")
    monkeypatch.setenv("SA_AMD_BUCKET_SHAPE", shape)
    monkeypatch.setenv("SA_AMD_BUCKET_BITS", bits)        # 18: two global passes of nine bits, the low 14 inside the buckets
    assert np.array_equal(build(text), exp)
    st = sa.last_stats()
    assert st["top32_first"] == 1
    if gen in ("uniform", "dna", "sigma200"):
        assert st["sort_passes"] == 3, st                 # two global passes + the bucket pass
    monkeypatch.setenv("SA_AMD_GROUP_CAP", "3")           # larger tied groups are left to the general path
    assert np.array_equal(build(text), exp)
    monkeypatch.delenv("SA_AMD_GROUP_CAP")
    monkeypatch.setenv("SA_AMD_BUCKET_FINISH_ALWAYS", "1")   # the fused round with the 20-pairs-per-thread shapes too
    assert np.array_equal(build(text), exp)
    monkeypatch.setenv("SA_AMD_NO_BUCKET_FINISH", "1")    # the round on the low key bits as a pass of its own (k_finish_sorted)
    assert np.array_equal(build(text), exp)
    monkeypatch.setenv("SA_AMD_NO_VALUE_BITS", "1")       # (18 bits, keys read from the text: two more key bits would travel in the values' top bits)
    assert np.array_equal(build(text), exp)
    monkeypatch.delenv("SA_AMD_NO_VALUE_BITS")
    monkeypatch.setenv("SA_AMD_NO_TEXT_KEYS", "1")        # (all 256 byte values: the first global pass would read its keys from the text)
    assert np.array_equal(build(text), exp)
    monkeypatch.setenv("SA_AMD_NO_ONESWEEP", "1")         # the global passes by the three-kernel engine (diagnostic library)
    assert np.array_equal(build_diag(text), exp)
    monkeypatch.delenv("SA_AMD_NO_ONESWEEP")
    monkeypatch.setenv("SA_AMD_NO_FUSED_FINISH", "1")     # ties on the top 32 bits through the general path
    assert np.array_equal(build(text), exp)
    monkeypatch.setenv("SA_AMD_NO_BUCKET_SORT", "1")
    assert np.array_equal(build(text), exp)
    if gen in ("uniform", "dna", "sigma200"):
        assert sa.last_stats()["sort_passes"] >= 4


@pytest.mark.parametrize("period,copies", [(300, 1025), (300, 1500), (64, 4096), (64, 4097), (64, 5000), (50, 8192), (50, 8193), (40, 20000), (700, 3000)])
def test_groups_of_thousands_are_ordered_in_lds(oracle, monkeypatch, period, copies):
    """k_group_sort_big: groups of 1 025 .. 8 192 members, one workgroup each, instead of the global radix sort.  A random block
    repeated `copies` times ties every suffix with `copies` - 1 others (minus the ones cut off by the end of the text) round
    after round; a second text mixes such groups with small ones and with a run.  With the kernel, without it, with a small cap
    on the groups k_group_sort owns, with every rank-doubling route."""
    rng = np.random.default_rng(period * 31 + copies)
    block = rng.integers(0, 256, period, dtype=np.uint8)
    text = np.tile(block, copies)
    mixed = np.concatenate([np.tile(block, copies // 2), rng.integers(0, 4, 50_000, dtype=np.uint8), np.tile(block[: period // 2], copies),
                            np.full(3000, 7, dtype=np.uint8), corpus.english(60_000, 5)])
    for t in (text, mixed):
        exp = oracle.sais(t)
        for env in ({}, {"SA_AMD_NO_BIG_GROUP_SORT": "1"}, {"SA_AMD_GROUP_CAP": "64"}, {"SA_AMD_FORCE_DENSE": "1"},
                    {"SA_AMD_FORCE_DENSE": "1", "SA_AMD_CHASE": "1"}, {"SA_AMD_NO_SPLIT": "1", "SA_AMD_DENSE_REKEY_MIN": "1"},
                    {"SA_AMD_SPARSE_DIV": "1"}, {"SA_AMD_NO_TEXT_ROUNDS": "1", "SA_AMD_NO_TOP32": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            assert np.array_equal(build(t), exp), (period, copies, env)
            for k in env:
                monkeypatch.delenv(k)


@pytest.mark.parametrize("period,copies", [(700, 97), (700, 130), (500, 300), (333, 700), (211, 1000), (211, 1024)])
def test_groups_across_tile_boundaries(oracle, monkeypatch, period, copies):
    """k_group_sort_straddle: a random block repeated `copies` times ties every suffix with `copies` - 1 others, and groups of
    that many members cut by a 2 048-element tile boundary are ordered by one workgroup each -- by counting up to 96 members,
    by a bitonic network in LDS beyond; text-keyed rounds, rank rounds and the chase, plus a mixed text"""
    rng = np.random.default_rng(period * 7 + copies)
    block = rng.integers(0, 256, period, dtype=np.uint8)
    text = np.tile(block, copies)
    mixed = np.concatenate([np.tile(block, copies // 2), corpus.english(50_000, 3), np.tile(block[: period // 3], copies), rng.integers(0, 3, 20_000, dtype=np.uint8)])
    for t in (text, mixed):
        exp = oracle.sais(t)
        for env in ({}, {"SA_AMD_NO_REPEAT_PROBE": "1"}, {"SA_AMD_NO_REPEAT_PROBE": "1", "SA_AMD_NO_GRAM_KEYS": "1", "SA_AMD_KEY_BITS": "24"},
                    {"SA_AMD_FORCE_DENSE": "1", "SA_AMD_CHASE": "1"}, {"SA_AMD_FORCE_DENSE": "1", "SA_AMD_CHASE": "7"}, {"SA_AMD_NO_BIG_GROUP_SORT": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            assert np.array_equal(build(t), exp), (period, copies, env)
            for k in env:
                monkeypatch.delenv(k)


def test_bucket_route_is_not_tried_on_a_text_with_one_huge_bucket(oracle, monkeypatch):
    """a text the entropy probe sends to the 32-bit first stage (its 4-byte prefixes are nearly unique) although one value of
    the top 16 key bits holds 1/32 of the suffixes: the probe's samples, counted per bucket, say so and the two global passes
    of the bucket route are not spent -- four global passes, the same array.  Without the probe (route forced) the exact check
    after the two passes finds the bucket and the keys are built again: the same array."""
    n = 1 << 24
    rng = np.random.default_rng(77)
    text = rng.integers(0, 256, n, dtype=np.uint8)
    text[0::32] = 97
    text[1::32] = 97
    exp = oracle.sais(text)
    monkeypatch.setenv("SA_AMD_BUCKET_MIN_N", "1")
    assert np.array_equal(build(text), exp)
    st = sa.last_stats()
    assert st["top32_first"] == 1 and st["sort_passes"] == 4, st
    monkeypatch.setenv("SA_AMD_FORCE_TOP32", "1")
    assert np.array_equal(build(text), exp)
    assert sa.last_stats()["sort_passes"] == 6, sa.last_stats()
    text = rng.integers(0, 256, n, dtype=np.uint8)       # flat: the route is taken
    monkeypatch.delenv("SA_AMD_FORCE_TOP32")
    assert np.array_equal(build(text), oracle.sais(text))
    assert sa.last_stats()["sort_passes"] == 3, sa.last_stats()


@pytest.mark.parametrize("gen,n,seed", [("english", 700_000, 11), ("dna", 1 << 20, 12), ("sigma3", 300_000, 13),
                                        ("dna_repeats", 500_000, 14)])
@pytest.mark.parametrize("cap", ["2", "5", "64", "1024"])
def test_group_sort_caps(oracle, monkeypatch, gen, n, seed, cap):
    """fused gather + in-LDS group sort (k_group_sort): with a tiny cap most groups are 'big' and go through
    the flag / global-sort / scatter-back path, groups cut by a 2048-element tile boundary go through
    k_group_sort_straddle; with the largest cap nearly everything is ordered in LDS"""
    text = corpus.sigma(n, seed, 3, 97) if gen == "sigma3" else getattr(corpus, gen)(n, seed)
    exp = oracle.sais(text)
    monkeypatch.setenv("SA_AMD_GROUP_CAP", cap)
    monkeypatch.setenv("SA_AMD_DENSE_REKEY_MIN", "1")          # whole-list global sorts keyed by group index at any size
    for top32 in ("SA_AMD_FORCE_TOP32", "SA_AMD_NO_TOP32"):
        monkeypatch.delenv("SA_AMD_FORCE_TOP32", raising=False)
        monkeypatch.delenv("SA_AMD_NO_TOP32", raising=False)
        monkeypatch.setenv(top32, "1")
        assert np.array_equal(build(text), exp)
        st = sa.last_stats()
        if st["unresolved_after_initial"] > 0 and cap == "1024":
            assert st["locally_sorted"] > 0


@pytest.mark.parametrize("gen,n,seed", [("english", 900_000, 31), ("sigma3", 400_000, 32), ("dna_repeats", 600_000, 33),
                                        ("periodic", 200_003, 34), ("unary", 50_000, 35)])
@pytest.mark.parametrize("cap", ["3", "100", "1024"])
def test_first_round_from_sorted_keys(oracle, monkeypatch, gen, n, seed, cap):
    """k_finish_sorted on 64-bit keys (opt-in SA_AMD_FUSED64): the first text-keyed round in one pass over the sorted keys,
    groups too large for a tile listed by k_todo_compact for the general route, survivors of both merged by k_surv_compact;
    and the same kernel finishing the 32-bit first stage (forced), with its fall-back to the general route"""
    if gen == "periodic":
        text = np.resize(np.frombuffer(b"abcabcabd", dtype=np.uint8), n)
    elif gen == "unary":
        text = np.full(n, 7, dtype=np.uint8)
    elif gen == "sigma3":
        text = corpus.sigma(n, seed, 3, 97)
    else:
        text = getattr(corpus, gen)(n, seed)
    exp = oracle.sais(text)
    monkeypatch.delenv("SA_AMD_NO_FUSED_FINISH", raising=False)      # (the suite may be run under global switches)
    monkeypatch.setenv("SA_AMD_GROUP_CAP", cap)
    monkeypatch.setenv("SA_AMD_FUSED64", "1")
    monkeypatch.setenv("SA_AMD_NO_TOP32", "1")
    assert np.array_equal(build(text), exp)
    assert sa.last_stats()["text_rounds"] >= 1
    monkeypatch.delenv("SA_AMD_NO_TOP32")
    monkeypatch.setenv("SA_AMD_FORCE_TOP32", "1")
    assert np.array_equal(build(text), exp)
    monkeypatch.setenv("SA_AMD_NO_FUSED_FINISH", "1")
    assert np.array_equal(build(text), exp)


@pytest.mark.parametrize("v64,v32", [("1", "1"), ("2", "2"), ("3", "3"), ("3", "4"), ("99", "-7")])
def test_sort_kernel_variants(oracle, monkeypatch, v64, v32):
    """the tile-scatter kernel shapes of the three-kernel pass (diagnostic library since round 4: no prefetch, 512 x 16, granule 8;
    out-of-range values fall back to the default) give the same arrays"""
    monkeypatch.setenv("SA_AMD_NO_ONESWEEP", "1")            # (the shapes belong to the three-kernel pass)
    monkeypatch.setenv("SA_AMD_SORT_VARIANT", v64)
    monkeypatch.setenv("SA_AMD_SORT32_VARIANT", v32)
    for gen, n, seed in (("english", 400_000, 21), ("uniform", 300_000, 22)):
        text = getattr(corpus, gen)(n, seed)
        exp = oracle.sais(text)
        for top32 in ("SA_AMD_FORCE_TOP32", "SA_AMD_NO_TOP32"):
            monkeypatch.delenv("SA_AMD_FORCE_TOP32", raising=False)
            monkeypatch.delenv("SA_AMD_NO_TOP32", raising=False)
            monkeypatch.setenv(top32, "1")
            assert np.array_equal(build_diag(text), exp)


@pytest.mark.parametrize("engine", ["single-pass", "shapes-1-1", "shapes-2-2", "shapes-0-3", "early-look", "three-kernel"])
def test_sort_engines_full_path(oracle, monkeypatch, engine):
    """the single-pass tile scatter in its default and its other tile shapes (two workgroups per CU, values through the keys'
    LDS buffer), with the early look-back, and the three-kernel pass it replaced give the same arrays on every route of the
    pipeline (several tiles per segment: 3 MiB texts)"""
    if engine == "three-kernel":
        monkeypatch.setenv("SA_AMD_NO_ONESWEEP", "1")
    if engine == "early-look":
        monkeypatch.setenv("SA_AMD_ONESWEEP_FLAGS", "1")
    if engine.startswith("shapes"):
        monkeypatch.setenv("SA_AMD_ONESWEEP64_SHAPE", engine.split("-")[1])
        monkeypatch.setenv("SA_AMD_ONESWEEP32_SHAPE", engine.split("-")[2])
    for gen, n, seed in (("english_corpus", 3_000_001, 31), ("uniform", 3_145_728, 32), ("dna", 2_999_999, 33), ("dna_repeats", 1_500_000, 34)):
        text = getattr(corpus, gen)(n, seed)
        exp = oracle.sais(text)
        for switch in (None, "SA_AMD_FORCE_TOP32", "SA_AMD_NO_TOP32", "SA_AMD_FORCE_DENSE", "SA_AMD_BINNED_ISA_ALWAYS"):
            if switch:
                monkeypatch.setenv(switch, "1")
            got = build_diag(text) if engine == "three-kernel" else build(text)      # (the three-kernel pass lives in the diagnostic library)
            if switch:
                monkeypatch.delenv(switch)
            assert np.array_equal(got, exp), (engine, gen, switch)


def test_randomised_inputs_and_regimes(oracle, monkeypatch, small_path):
    """short form of tools/stress.py: random sizes, alphabets and structures with the regime switches
    toggled at random; every array equals the oracle's"""
    rng = np.random.default_rng(2026)
    switches = ["SA_AMD_FORCE_TOP32", "SA_AMD_NO_LOCAL_SORT", "SA_AMD_NO_TEXT_ROUNDS", "SA_AMD_FORCE_DENSE",
                "SA_AMD_BINNED_ISA_ALWAYS", "SA_AMD_NO_TOP32"]
    for it in range(250):
        n = int(rng.choice([rng.integers(0, 200), rng.integers(200, 9000), rng.integers(9000, 120000)]))
        kind = it % 4
        if kind == 0:
            s = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1:
            s = (rng.integers(0, int(rng.integers(1, 6)), n) + 60).astype(np.uint8)
        elif kind == 2:
            s = np.resize(rng.integers(0, 256, int(rng.integers(1, 30)), dtype=np.uint8), n).astype(np.uint8)
        else:
            s = rng.integers(0, 256, n, dtype=np.uint8)
            if n > 100:
                ln = int(rng.integers(2, n // 3)); a = int(rng.integers(0, n - ln)); b = int(rng.integers(0, n - ln))
                s[b:b + ln] = s[a:a + ln]
        s = np.ascontiguousarray(s)
        for k in switches + ["SA_AMD_SPARSE_DIV"]:
            monkeypatch.delenv(k, raising=False)
        for k in switches:
            if rng.random() < 0.2:
                monkeypatch.setenv(k, "1")
        if rng.random() < 0.3:
            monkeypatch.setenv("SA_AMD_SPARSE_DIV", str(int(rng.choice([1, 4, 64, 10**9]))))
        assert np.array_equal(build(s), oracle.sais(s)), (it, n, kind)


# ---- the product library is bit-exact under ANY environment ---------------------------------------

ALL_KNOBS = ["SA_AMD_SORT_VARIANT", "SA_AMD_SORT32_VARIANT", "SA_AMD_KEY_BITS", "SA_AMD_GROUP_CAP", "SA_AMD_SPARSE_DIV",
             "SA_AMD_FORCE_DENSE", "SA_AMD_NO_TEXT_ROUNDS", "SA_AMD_NO_LOCAL_SORT", "SA_AMD_NO_TOP32", "SA_AMD_FORCE_TOP32",
             "SA_AMD_NO_FUSED_FINISH", "SA_AMD_FUSED64", "SA_AMD_NO_PACKED_TEXT", "SA_AMD_NO_BINNED_ISA",
             "SA_AMD_BINNED_ISA_ALWAYS", "SA_AMD_NO_RUN_SKIP", "SA_AMD_RUN_SKIP_MIN", "SA_AMD_TIMING_ONLY_INITIAL_SORT",
             "SA_AMD_MAX_TEXT_ROUNDS", "SA_AMD_BINNED_MIN", "SA_AMD_CHASE", "SA_AMD_NO_REPEAT_PROBE", "SA_AMD_NO_FIRST_TAIL", "SA_AMD_DENSE_REKEY_MIN", "SA_AMD_SCATTER_LEVELS",
             "SA_AMD_CACHE_MAX_BYTES", "SA_AMD_COPY_THREADS", "SA_AMD_STAGED_MIN_BYTES", "SA_AMD_BATCH_THREADS",
             "SA_AMD_NO_GRAM_KEYS", "SA_AMD_GRAM_MIN_N", "SA_AMD_GRAM_G", "SA_AMD_GRAM_TAIL", "SA_AMD_CHASE_BIG", "SA_AMD_CHASE_BIG_MIN", "SA_AMD_NO_SPLIT", "SA_AMD_SPLIT_MIN", "SA_AMD_SPLIT_GROUP_MIN",
             "SA_AMD_SMALL_MAX", "SA_AMD_TOP32_PROBE_MIN_N", "SA_AMD_TOP32_PARTNERS_X100", "SA_AMD_TOP32_COLLISIONS_X100", "SA_AMD_NO_ONESWEEP", "SA_AMD_ONESWEEP64_SHAPE", "SA_AMD_ONESWEEP32_SHAPE", "SA_AMD_ONESWEEP_FLAGS", "SA_AMD_NO_BIG_GROUP_SORT", "SA_AMD_NO_TEXT_KEYS", "SA_AMD_NO_VALUE_BITS", "SA_AMD_NO_BUCKET_SORT", "SA_AMD_NO_BUCKET_FINISH", "SA_AMD_BUCKET_FINISH_ALWAYS", "SA_AMD_BUCKET_BITS", "SA_AMD_BUCKET_MIN_N", "SA_AMD_BUCKET_SHAPE", "SA_AMD_NUMA", "SA_AMD_HELPER_THREADS", "SA_AMD_NO_LANES", "SA_AMD_LANES_MIN_N", "SA_AMD_NO_PREFAULT", "SA_AMD_PREFAULT_WAIT", "SA_AMD_PREFAULT_KEEP", "SA_AMD_PINNED_MAX_BYTES", "SA_AMD_NO_DEFER", "SA_AMD_NETWORK_MIN", "SA_AMD_NO_UPFRONT_COUNTS", "SA_AMD_NO_KERNEL_D2H", "SA_AMD_KERNEL_D2H_ALWAYS", "SA_AMD_NO_FLAT_RULE", "SA_AMD_NO_POSTED_READBACK", "SA_AMD_COUNT_NEXT_MIN_N", "SA_AMD_COUNT_NEXT_BELOW_N", "SA_AMD_EARLY_DIV", "SA_AMD_EARLY_MIN_BYTES", "SA_AMD_EARLY_CHUNK_BYTES", "SA_AMD_EARLY_WAIT_CHUNKS", "SA_AMD_SAMPLE_SORT", "SA_AMD_SAMPLE_SORT_MIN_N", "SA_AMD_SAMPLE_LOG", "SA_AMD_SAMPLE_MERGE", "SA_AMD_CACHE_IDLE_MS", "SA_AMD_NO_REDUCED", "SA_AMD_NO_UNARY_SHORTCUT",
             "SA_AMD_DEBUG_SYNC", "SA_AMD_VERBOSE"]


def test_knob_list_is_complete():
    """every SA_AMD_* name the sources read is in ALL_KNOBS (SA_AMD_DEVICE picks the GPU; an ordinal that does not exist
    is an error code, tested in test_abi.py, not a different array)"""
    import glob
    import re
    names = set()
    for f in glob.glob(os.path.join(ROOT, "suffix_array_amd", "csrc", "**", "*.*"), recursive=True):
        if f.endswith((".hpp", ".hip", ".inc")):
            names |= set(re.findall(r'"(SA_AMD_[A-Z0-9_]+)"', open(f).read()))
    assert names - {"SA_AMD_DEVICE"} <= set(ALL_KNOBS), names - set(ALL_KNOBS)


def test_any_environment_is_bit_exact(oracle, monkeypatch):
    """no value of any SA_AMD_* variable may change a result of the SHIPPED library -- the timing
    ablations and the truncated build exist only in libsuffix_array_amd_diag.so.  Random values (numbers in and out of
    range, negative, huge, empty, text) for every knob at once, 40 rounds over texts that reach every regime."""
    rng = np.random.default_rng(424242)
    texts = [corpus.english(260_000, 3), corpus.uniform(180_000, 2), corpus.dna(300_000, 4), corpus.dna_repeats(250_000, 5),
             np.resize(np.frombuffer(b"abcab", dtype=np.uint8), 90_001).copy(), np.full(40_000, 9, dtype=np.uint8),
             corpus.sigma(120_000, 7, 3, 97), np.concatenate([corpus.english(70_000, 8)] * 2)]
    expected = [oracle.sais(t) for t in texts]
    pool = ["0", "1", "2", "3", "5", "7", "9", "16", "33", "64", "100", "1024", "4096", "-1", "-77", "999999999999",
            "18446744073709551616", "", "abc", "1e9", " 4", "0x10", "yes"]
    for it in range(40):
        for k in ALL_KNOBS:
            monkeypatch.delenv(k, raising=False)
            if rng.random() < 0.55:
                monkeypatch.setenv(k, str(rng.choice(pool)))
        monkeypatch.setenv("SA_AMD_DEBUG_SYNC", "0")           # (read once per process; keep the suite fast)
        monkeypatch.setenv("SA_AMD_VERBOSE", "0")
        i = it % len(texts)
        env = {k: os.environ[k] for k in ALL_KNOBS if k in os.environ}
        assert np.array_equal(build(texts[i]), expected[i]), (it, i, env)


def test_device_ordinal_from_the_environment(oracle, monkeypatch):
    """SA_AMD_DEVICE picks the GPU of the host-pointer entry points; an ordinal that does not exist is an error code
    (SA_AMD_EINVAL), never another result; text that is not a number is ignored"""
    s = corpus.english(30_000, 2)
    exp = oracle.sais(s)
    monkeypatch.setenv("SA_AMD_DEVICE", "0")
    assert np.array_equal(build(s), exp)
    monkeypatch.setenv("SA_AMD_DEVICE", "not a number")
    assert np.array_equal(build(s), exp)
    monkeypatch.setenv("SA_AMD_DEVICE", str(sa.lib().sa_amd_device_count() + 5))
    with pytest.raises(sa.SuffixArrayError) as e:
        build(s)
    assert e.value.code == -1
    out = np.zeros(s.size + 1, dtype=np.uint32)
    st = (ctypes.c_int32 * 1)()
    T = (ctypes.c_void_p * 1)(s.ctypes.data); S = (ctypes.c_void_p * 1)(out.ctypes.data)
    N = (ctypes.c_int32 * 1)(s.size); D = (ctypes.c_int32 * 1)(77)
    assert sa.lib().sa_amd_saca_batch(T, S, N, D, 1, st) == -1 and st[0] == -1          # bad ordinal in the batch call


def test_unpack_rejects_truncated_data():
    """a packed file whose data section is shorter than the full blocks it must contain is InvalidData, not an array with
    zeros filled in; misaligned device scratch is rejected by sa_amd_saca_device"""
    rng = np.random.default_rng(3)
    arr = rng.permutation(1000).astype(np.uint32)
    blob = sa.pack(arr)
    cut = bytearray(blob[: 16 + 3 * 160])                     # three of eight 10-bit blocks
    cut[8:16] = int(len(cut) - 16).to_bytes(8, "little")      # a consistent header: only the block count is wrong
    with pytest.raises(ValueError):
        sa.unpack(bytes(cut))
    hip = ctypes.CDLL("libamdhip64.so")                       # (already in the process: the product library links it)
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    wb = sa.workspace_bytes(4096)
    ptrs = []
    for size in (4096, 4097 * 4, wb + 512):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), size) == 0
        assert hip.hipMemset(p, 0, size) == 0
        ptrs.append(p.value)
    t, o, w = ptrs
    aligned = (w + 255) & ~255
    L = sa.lib()
    assert L.sa_amd_saca_device(t, o, 4096, aligned, wb, None, None) == 0
    assert L.sa_amd_saca_device(t, o, 4096, aligned + 4, wb, None, None) == -1
    for p in ptrs:
        hip.hipFree(p)


def test_device_entry_point_with_unaligned_text(oracle, monkeypatch):
    """sa_amd_saca_device takes the text wherever the caller has it in HBM: offsets 1, 3 and 5 from an aligned block (the
    kernels read the text with 8- and 16-byte loads where the address allows and byte loads where it does not), gram keys
    and plain keys, small and word-structured alphabets"""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    L = sa.lib()
    monkeypatch.setenv("SA_AMD_GRAM_MIN_N", "1")
    monkeypatch.setenv("SA_AMD_FORCE_TOP32", "1")          # with all 256 byte values in the text the first pass of the bucket
    monkeypatch.setenv("SA_AMD_BUCKET_MIN_N", "1")         # route reads its keys from the text itself: two aligned words per key
    texts = [corpus.english(150_001, 3), corpus.dna(90_000, 4), corpus.uniform(70_003, 5), np.full(33_000, 9, dtype=np.uint8),
             corpus.uniform(300_007, 6)]
    for t in texts:
        n = int(t.size)
        exp = oracle.sais(t)
        wb = sa.workspace_bytes(n)
        dt, do, dw = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(dt), n + 64) == 0 and hip.hipMalloc(ctypes.byref(do), 4 * (n + 1) + 64) == 0
        assert hip.hipMalloc(ctypes.byref(dw), wb + 512) == 0
        work = (dw.value + 255) & ~255
        for off in (0, 1, 3, 5):
            assert hip.hipMemcpy(dt.value + off, t.ctypes.data, n, 1) == 0
            assert L.sa_amd_saca_device(dt.value + off, do.value, n, work, wb, None, None) == 0
            out = np.zeros(n + 1, dtype=np.uint32)
            assert hip.hipMemcpy(out.ctypes.data, do.value, 4 * (n + 1), 2) == 0
            assert np.array_equal(out, exp), (n, off)
        for p in (dt, do, dw):
            hip.hipFree(p)


# ---- BASELINE.json full-size configs: size-independent properties + oracle equality ----------

def _sampled_neighbours_ordered(text, arr, seed, samples=3000, width=256):
    rng = np.random.default_rng(seed)
    n = text.size
    for i in rng.integers(1, n, samples):
        a, b = int(arr[i]), int(arr[i + 1])
        x, y = text[a:a + width].tobytes(), text[b:b + width].tobytes()
        assert x < y or (x == y and len(x) == width), (i, a, b)


@pytest.mark.parametrize("name", ["c2_uniform_64m", "c3_english_256m", "c3_iid_256m", "c4_dna_1g"])
def test_full_size_configs(oracle, name):
    """BASELINE.json configs 2-4 through the C ABI (host pointers): SA[0] = n, the linear-time form of reference
    src/sa.rs:72-84 on the CPU (oracle) AND on the GPU (sa_amd_check_integrity), sampled direct comparisons;
    the 64 MiB config also equals the oracle's array bit for bit"""
    text = corpus.workload(name)
    arr = build(text)
    assert arr[0] == text.size
    assert oracle.verify_mt(text, arr) == 1
    assert sa.check_integrity(text, arr) is True
    _sampled_neighbours_ordered(text, arr, 11)
    if text.size <= (64 << 20):
        assert np.array_equal(arr, oracle.sais(text))
    sa.lib().sa_amd_release_cache()


def test_dna_1g_with_planted_repeats(oracle):
    """the harder C4 variant of SURVEY.md 8d: 1 GiB of DNA with copied segments of 1-100 KiB and 1 % point mutations"""
    text = corpus.dna_repeats(1 << 30, 4, 0.2)
    arr = build(text)
    assert arr[0] == text.size
    assert oracle.verify_mt(text, arr) == 1
    assert sa.check_integrity(text, arr) is True
    assert sa.last_stats()["rounds"] >= 2
    sa.lib().sa_amd_release_cache()


def test_c5_batch_of_512m_texts(oracle):
    """BASELINE.json config 5 through ONE call of sa_amd_saca_batch: 8 independent 512 MiB texts (seeds 50 .. 57), text i on
    device i mod G over all visible devices; every array verified on the device (reference src/sa.rs:72-84 in linear
    time), two of them also by the oracle's threaded verifier"""
    texts = [corpus.workload("c5_uniform_512m", rank=r) for r in range(8)]
    outs = sa.saca_batch(texts)
    for i, (t, o) in enumerate(zip(texts, outs)):
        assert o[0] == t.size
        assert sa.check_integrity(t, o) is True, i
        _sampled_neighbours_ordered(t, o, 5, samples=200)
    for i in (0, 7):
        assert oracle.verify_mt(texts[i], outs[i]) == 1
    assert not np.array_equal(outs[0][:1000], outs[1][:1000])           # independent texts, independent arrays
    sa.lib().sa_amd_release_cache()


# ---- bench.py: the N > 1 path with the real backend, two ranks rehearsed on this one GPU --------------

def test_bench_two_ranks_share_one_gpu():
    """`python bench.py --gpus 2` starts its own ranks (torch.distributed.run, gloo control plane because both ranks sit on
    device 0: SA_BENCH_SHARE_GPU=1), every rank builds its own text through libsuffix_array_amd.so, both arrays are
    verified on the device, and rank 0 prints one JSON line for the whole job"""
    import subprocess
    import sys
    env = dict(os.environ, SA_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                           "--workload", "c2_uniform_64m", "--n", str(8 << 20), "--e2e-calls", "2"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["verified"] is True and out["config"]["n_bytes"] == 8 << 20
    assert out["value"] == pytest.approx(2 * (8 << 20) / 1e6 / (out["ms_per_step"] / 1e3), rel=1e-3)
    assert out["batch_c5"]["verified"] is True and out["batch_c5"]["texts"] == 2
    assert out["end_to_end"]["reused_buffer"]["MB_per_s"] > 0
    rf = out["roofline"]
    assert rf["kernel"] == rf["kernels"][0]["name"] and rf["achieved"] > 0          # the class with the most measured time
    assert all(k["ms_per_step"] <= rf["kernels"][0]["ms_per_step"] for k in rf["kernels"])
    # who ran where: both ranks report the device they used (the same one here, which the line says) and their own check
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and all(r["verified"] for r in out["ranks"])
    assert out["ranks"][0]["pci_bus_id"] == out["ranks"][1]["pci_bus_id"] is not None and out["distinct_gpus"] is False
    assert out["ranks"][0]["pid"] != out["ranks"][1]["pid"]


def test_bench_refuses_ranks_that_share_a_gpu_unasked():
    """without SA_BENCH_SHARE_GPU=1 an N > 1 run whose ranks do not sit on N distinct GPUs is an error, not a number: here
    LOCAL_RANK 0 is forced on both ranks of a real two-rank launch"""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("SA_BENCH_SHARE_GPU", None)
    env["SA_BENCH_FORCE_LOCAL_RANK"] = "0"
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                           "--workload", "c2_uniform_64m", "--n", str(1 << 20), "--no-end-to-end", "--no-batch"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode != 0
    assert "not distinct" in proc.stderr and not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]


def test_bench_three_ranks_share_one_gpu():
    """the N > 1 path with more ranks than the two of the test above (three: a GPU box lets six processes on its card at once --
    the test runner is one of them, and a launch of six ranks was seen as eight GPU processes by the box's guard, which then
    ends the whole run): every rank builds and verifies its own text, one line for the whole job, three entries in ranks[]"""
    import subprocess
    import sys
    env = dict(os.environ, SA_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1",
                           "--workload", "c2_uniform_64m", "--n", str(4 << 20), "--e2e-calls", "1"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["verified"] is True and len(out["ranks"]) == 3
    assert sorted(r["rank"] for r in out["ranks"]) == list(range(3)) and all(r["verified"] for r in out["ranks"])
    assert out["value"] == pytest.approx(3 * (4 << 20) / 1e6 / (out["ms_per_step"] / 1e3), rel=1e-3)
    assert out["batch_c5"]["texts"] == 3 and out["batch_c5"]["verified"] is True


def test_bench_batch_leg_as_a_child_process():
    """`bench.py --batch-api-only`: what the N = 1 line starts as a child when more than one GPU is visible (one process then drives
    all of them through ONE sa_amd_saca_batch call) -- on this one-GPU box the same command, small texts"""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch-api-only", "--batch-texts", "3", "--text-bytes", str(4 << 20),
                           "--small-batch-texts", "64"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["entry_point"] == "sa_amd_saca_batch" and out["texts"] == 3 and out["verified"] is True and out["devices"] >= 1
    assert out["small_texts"]["texts"] == 64 and out["small_texts"]["verified"] is True


def test_bench_single_rank_line_with_batch_api():
    """`python bench.py` at N = 1 (small sizes): the dominant kernel is picked by measured time, the CPU baseline carries its
    probe log, and the config-5 leg calls sa_amd_saca_batch once with several texts"""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workload",
                           "c3_english_256m", "--n", str(8 << 20), "--e2e-calls", "2", "--batch-texts", "4", "--cpu-sample", str(1 << 20),
                           "--config-steps", "2"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert out["verified"] is True and out["n_gpus"] == 1
    names = [k["name"] for k in out["roofline"]["kernels"]]
    assert out["roofline"]["kernel"] == names[0] and len(names) >= 1
    assert out["batch_api"]["texts"] == 4 and out["batch_api"]["verified"] is True and out["batch_api"]["entry_point"] == "sa_amd_saca_batch"
    assert out["cpu_baseline"]["probe"][0]["step"].startswith("cargo") and out["cpu_baseline"]["pinned_cpu"] is not None
    # the other BASELINE configs behind the headline workload (here at 8 MiB each): verified, their own dominant class, end to end
    cf = out["configs"]
    assert [c for c in cf if not c.startswith("_")] == ["c2_uniform_64m", "c2_uniform_256m", "c4_dna_1g", "c5_uniform_512m"]
    for name, c in cf.items():
        if name.startswith("_"):
            continue
        assert c["verified"] is True and c["n_bytes"] == 8 << 20 and c["ms_per_build"] > 0, name
        assert c["roofline"]["kernel"] and 0 < c["roofline"]["frac"] < 1 and c["end_to_end"]["fresh_buffer"]["MB_per_s"] > 0, name
    assert cf["c4_dna_1g"]["sigma"] == 4 and cf["c2_uniform_64m"]["sigma"] == 256
    assert out["ranks"][0]["pci_bus_id"] and out["roofline"]["traffic_source"] is None      # (traffic.json is for the full-size text)


# ---- next rows: bucket table and integrity check --------------------------------

def _texts_for_extras():
    cases = {k: v for k, v in adversarial_cases().items() if len(v) < 3000}
    cases["english_200k"] = corpus.english(200_000, 3).tobytes()
    cases["uniform_300k"] = corpus.uniform(300_001, 2).tobytes()
    cases["dna_100k"] = corpus.dna(100_000, 4).tobytes()
    return cases


def test_bucket_table_matches_reference_restatement(oracle):
    """reference src/sa.rs:89-119 restated in oracle_bucket_table (bigram counts + prefix sum)"""
    for name, s in _texts_for_extras().items():
        arr = oracle.sais(s)
        got = sa.bucket_table(s, arr)
        assert got.size == 256 * 257 + 1
        assert np.array_equal(got, oracle.bucket_table(s)), name


def test_bucket_table_needs_the_text_only(oracle):
    """the reference's enable_buckets reads the text and nothing else (src/sa.rs:96-116): no suffix array is handed over (NULL),
    on every tiny length, on a text of all 256 byte values large enough for every pair of workgroups (k_bigram_hist) and on
    texts of one and two byte values (every count in one or four bins)"""
    L = sa.lib()
    rng = np.random.default_rng(77)
    cases = [np.zeros(0, dtype=np.uint8)] + [rng.integers(0, 256, size=k, dtype=np.uint8) for k in (1, 2, 3, 15, 16, 17, 18, 31, 33, 255, 4097)]
    cases += [corpus.uniform((40 << 20) + 13, 21), np.full(3_000_001, 255, dtype=np.uint8), (rng.integers(0, 2, size=5_000_003) * 255).astype(np.uint8),
              corpus.english_corpus(9_000_001, 5)]
    for t in cases:
        bkt = np.zeros(256 * 257 + 1, dtype=np.uint32)
        assert L.sa_amd_bucket_table(t.ctypes.data if t.size else None, t.size, None, bkt.ctypes.data) == 0
        assert np.array_equal(bkt, oracle.bucket_table(t)), t.size
        assert int(bkt[-1]) == t.size + 1


def test_bucket_table_device_form_on_unaligned_texts(oracle):
    """sa_amd_bucket_table_device with dSA == NULL (bigram counts) takes the text at any byte offset: 16-byte loads from the
    first aligned address, bytes in front of and behind them; with a suffix array it answers by binary search -- same table"""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    L = sa.lib()
    for t in (corpus.english(300_007, 3), corpus.uniform(1_000_003, 5), corpus.uniform(37, 6), corpus.dna(70_001, 4)):
        n = int(t.size)
        exp = oracle.bucket_table(t)
        arr = oracle.sais(t)
        dt, db, ds = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(dt), n + 64) == 0 and hip.hipMalloc(ctypes.byref(db), 4 * (256 * 257 + 1)) == 0
        assert hip.hipMalloc(ctypes.byref(ds), 4 * (n + 1)) == 0
        assert hip.hipMemcpy(ds.value, arr.ctypes.data, 4 * (n + 1), 1) == 0
        for off in (0, 1, 7, 15):
            assert hip.hipMemcpy(dt.value + off, t.ctypes.data, n, 1) == 0
            for dsa in (None, ds.value):
                got = np.zeros(256 * 257 + 1, dtype=np.uint32)
                assert L.sa_amd_bucket_table_device(dt.value + off, dsa, n, db.value, None) == 0
                assert hip.hipMemcpy(got.ctypes.data, db.value, got.nbytes, 2) == 0
                assert np.array_equal(got, exp), (n, off, dsa is None)
        for q in (dt, db, ds):
            hip.hipFree(q)


def test_enable_buckets_on_constructed_array(oracle):
    s = b"splendid splendor"
    obj = sa.SuffixArray(s)
    assert obj.buckets() is None
    obj.enable_buckets()
    bkt = obj.buckets()
    idx = ord("s") * 257 + (ord("p") + 1) + 1                       # sub-bucket (c0, c1), reference src/sa.rs:130
    arr = obj.into_parts()[1]
    assert sorted(int(p) for p in arr[bkt[idx - 1]:bkt[idx]]) == [0, 9]   # doc-test of reference src/lib.rs:28-29
    assert np.array_equal(bkt, oracle.bucket_table(s))


def test_saca_with_buckets_single_round_trip(oracle):
    t = corpus.english(150_000, 8)
    arr = np.zeros(t.size + 1, dtype=np.uint32)
    bkt = np.zeros(256 * 257 + 1, dtype=np.uint32)
    assert sa.lib().sa_amd_saca_u8_buckets(t.ctypes.data, arr.ctypes.data, t.size, bkt.ctypes.data) == 0
    assert np.array_equal(arr, oracle.sais(t)) and np.array_equal(bkt, oracle.bucket_table(t))


def test_check_integrity_matches_reference_restatement(oracle):
    """reference src/sa.rs:72-84 restated literally in oracle_check_integrity"""
    rng = np.random.default_rng(5)
    for name, s in _texts_for_extras().items():
        good = oracle.sais(s)
        assert sa.check_integrity(s, good) is True, name
        n = len(s)
        if n >= 2:
            for _ in range(4):
                bad = good.copy()
                i, j = rng.integers(0, n + 1, 2)
                bad[i], bad[j] = bad[j], bad[i]
                assert sa.check_integrity(s, bad) == (oracle.check_integrity(s, bad) == 1), name
            dup = good.copy(); dup[n // 2] = dup[n // 2 + 1]
            assert sa.check_integrity(s, dup) is False
            assert sa.check_integrity(s, good[:-1]) is False               # src/sa.rs:73-75
            oob = good.copy(); oob[1] = n + 7
            assert oracle.check_integrity(s, oob) == -1
            with pytest.raises(IndexError):
                sa.check_integrity(s, oob)


def test_check_integrity_device_forms_on_a_large_array(oracle):
    """both forms of sa_amd_check_integrity_device -- the small work block (random-store inverse permutation) and the
    streaming one (binned inverse permutation through two radix passes, one random rank line per slot) -- on a 40 M-entry
    array: the good array, adjacent swaps, a duplicate, the empty suffix in a wrong slot, an entry out of range"""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    L = sa.lib()
    t = corpus.english_corpus(40_000_003, 77)
    n = int(t.size)
    good = build(t)
    assert oracle.verify_mt(t, good) == 1
    big, small = int(L.sa_amd_check_integrity_work_bytes(n)), 4 * (n + 1) + 256
    assert big > 5 * small // 2
    dt, ds, dw = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dt), n + 64) == 0 and hip.hipMalloc(ctypes.byref(ds), 4 * (n + 1) + 64) == 0
    assert hip.hipMalloc(ctypes.byref(dw), big + 512) == 0
    work = (dw.value + 255) & ~255
    assert hip.hipMemcpy(dt.value, t.ctypes.data, n, 1) == 0
    rng = np.random.default_rng(3)

    def both(arr):
        assert hip.hipMemcpy(ds.value, arr.ctypes.data, 4 * (n + 1), 1) == 0
        return (L.sa_amd_check_integrity_device(dt.value, n, ds.value, work, big, None),
                L.sa_amd_check_integrity_device(dt.value, n, ds.value, work, small, None))

    assert both(good) == (1, 1)
    for _ in range(3):
        bad = good.copy()
        i = int(rng.integers(1, n))
        bad[i], bad[i + 1] = bad[i + 1], bad[i]
        assert both(bad) == (0, 0), i
    bad = good.copy(); bad[n // 3] = bad[n // 3 + 1]                     # a value twice, another one missing
    assert both(bad) == (0, 0)
    bad = good.copy(); bad[0], bad[5] = bad[5], bad[0]                   # the empty suffix not in slot 0
    assert both(bad) == (0, 0)
    bad = good.copy(); bad[n // 2] = n + 7                               # the reference panics on the slice index
    assert both(bad) == (-6, -6)
    bad = good.copy(); bad[1:] = np.roll(good[1:], 1)                    # a rotation: a permutation, almost everywhere in order
    assert both(bad) == (0, 0)
    for p in (dt, ds, dw):
        hip.hipFree(p)


def test_from_parts_uses_gpu_check(oracle):
    s = corpus.english(50_000, 4)
    good = oracle.sais(s)
    assert sa.SuffixArray.from_parts(s, good) is not None
    bad = good.copy(); bad[100], bad[101] = bad[101], bad[100]
    assert sa.SuffixArray.from_parts(s, bad) is None


# ---- next row 8f-4: batched search (reference src/sa.rs:164-253) --------------------------------
# naive checkers restated from the reference's own tests, src/tests.rs:104-132

def _lcp(a, b):
    k = 0
    while k < len(a) and k < len(b) and a[k] == b[k]:
        k += 1
    return k


def naive_contains(s, pat):
    return any(pat == s[i:min(len(s), i + len(pat))] for i in range(0, max(len(s) - len(pat), 0) + 1))


def naive_search_all(s, pat):
    return [i for i in range(0, max(len(s) - len(pat), 0) + 1) if pat == s[i:min(len(s), i + len(pat))]]


def naive_search_lcp(s, pat):
    best = 0
    for i in range(len(s) + 1):
        best = max(best, _lcp(pat, s[i:]))
    return pat[:best]


def _bytes_with_pat(rng, n):
    """reference src/tests.rs:79-102: no_junk / trail_junk / all_junk patterns"""
    s = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    m = int(n * rng.random())
    kind = int(rng.integers(0, 3))
    if kind == 0:
        i = int(rng.integers(0, n - m + 1)); pat = s[i:i + m]
    elif kind == 1:
        i = int(rng.integers(0, n - m + 1)); j = int(rng.integers(0, m + 1))
        pat = s[i:i + (m - j)] + rng.integers(0, 256, j, dtype=np.uint8).tobytes()
    else:
        pat = rng.integers(0, 256, m, dtype=np.uint8).tobytes()
    return s, pat


def test_search_properties_of_the_reference():
    """contains_correctness / search_all_correctness / search_lcp_correctness, reference src/tests.rs:19-59"""
    rng = np.random.default_rng(99)
    for it in range(60):
        n = int(rng.integers(0, 600)) if it else 0
        s, pat = _bytes_with_pat(rng, n)
        obj = sa.SuffixArray(s)
        assert obj.contains(pat) == naive_contains(s, pat)
        assert sorted(int(x) for x in obj.search_all(pat)) == naive_search_all(s, pat)
        r = obj.search_lcp(pat)
        assert s[r.start:r.stop] == naive_search_lcp(s, pat)


def test_doctests_of_the_reference():
    """reference src/lib.rs:16-41"""
    s = b"splendid splendor"
    obj = sa.SuffixArray(s)
    assert obj.contains(b"splend")
    assert sorted(int(x) for x in obj.search_all(b"splend")) == [0, 9]
    r = obj.search_lcp(b"splash")
    assert s[r.start:r.stop] == b"spl"


def test_batched_search_on_device_resident_index(oracle):
    text = corpus.english(200_000, 12)
    s = text.tobytes()
    ix = sa.DeviceIndex(text)                      # array built on the device, never downloaded for the search
    arr = oracle.sais(text)
    assert np.array_equal(ix.suffix_array(), arr)
    assert ix.check_integrity() and np.array_equal(ix.buckets(), oracle.bucket_table(text))
    rng = np.random.default_rng(3)
    pats = [b"", b"e", b" ras ", s[-7:], s[:300], b"zzzzzzzzzz", b"\xff"]
    for _ in range(300):
        i = int(rng.integers(0, len(s) - 40)); ln = int(rng.integers(1, 40))
        p = s[i:i + ln]
        if rng.random() < 0.4:
            p = p[:-1] + bytes([int(rng.integers(0, 256))])
        pats.append(p)
    res = ix.search(pats)
    for q, p in enumerate(pats):
        occ = naive_search_all(s, p) if len(p) else list(range(len(s) + 1))
        got = sorted(int(x) for x in arr[res["lo"][q]:res["hi"][q]])
        assert got == occ, p
        assert bool(res["contains"][q]) == (len(occ) > 0)
        st, ln = int(res["lcp_start"][q]), int(res["lcp_len"][q])
        assert s[st:st + ln] == p[:ln]
        if q < 40:
            assert ln == len(naive_search_lcp(s, p))
    # With the bucket table built (enable_buckets) the searches start from the pattern's (c0, c1) bucket like the
    # reference's get_bucket (src/sa.rs:123-144); search_lcp then has the reference's empty-bucket branch (src/sa.rs:211-222):
    # first suffix of the top-level bucket of c0 with length 1, or s.len()..s.len() -- the RANGE, not only the substring
    bkt = oracle.bucket_table(text)

    def ref_search_lcp_with_buckets(p):
        if len(p) > 1:
            idx = p[0] * 257 + p[1] + 2
            blo, bhi = int(bkt[idx - 1]), int(bkt[idx])
        elif len(p) == 1:
            blo, bhi = int(bkt[p[0] * 257]), int(bkt[p[0] * 257 + 257])
        else:
            blo, bhi = 0, 1
        if blo == bhi:
            tlo, thi = int(bkt[p[0] * 257]), int(bkt[p[0] * 257 + 257])
            return (int(arr[tlo]), 1) if thi > tlo else (len(s), 0)
        return None                                             # (non-empty bucket: the substring is checked above)
    pats2 = [b"e\x00", b"e\xff", b"t\x01x", b"\xff", b"\xffa", b"\x00\x00", b"q!", b"e"]
    res2 = ix.search(pats2)
    hit = 0
    for q, p in enumerate(pats2):
        exp = ref_search_lcp_with_buckets(p)
        if exp is not None:
            assert (int(res2["lcp_start"][q]), int(res2["lcp_len"][q])) == exp, p
            assert not res2["contains"][q] and res2["lo"][q] == res2["hi"][q]
            hit += 1
    assert hit >= 5
    ix.close()


# ---- next row 8f-3: packed format (reference src/packed_sa.rs; byte-level parity unpinned) ---------

def test_pack_matches_model_and_round_trips(oracle):
    import pack_model
    rng = np.random.default_rng(11)
    for length in (1, 2, 5, 127, 128, 129, 1000, 4096, 4097, 100_001):
        arr = rng.permutation(length).astype(np.uint32)
        blob = sa.pack(arr)
        if length <= 5000:
            assert blob == pack_model.pack(arr), length
        assert np.array_equal(sa.unpack(blob), arr), length
    with pytest.raises(ValueError):
        sa.unpack(b"XXXX" + blob[4:])
    with pytest.raises(ValueError):
        sa.unpack(blob[:-3])


def test_pack_correctness_property_of_the_reference(oracle):
    """reference src/tests.rs:61-76: dump == dump_bytes, load_bytes round-trips to the same array"""
    import io
    rng = np.random.default_rng(8)
    for it in range(25):
        n = int(rng.integers(0, 4096)) if it else 0
        s = rng.integers(0, 256, n, dtype=np.uint8)
        sa1 = sa.SuffixArray(s)
        bytes1 = sa1.dump_bytes()
        buf = io.BytesIO()
        sa1.dump(buf)
        assert buf.getvalue() == bytes1
        sa2 = sa.SuffixArray.load_bytes(s, bytes1)
        assert np.array_equal(sa1.into_parts()[1], sa2.into_parts()[1])
    bad = sa.SuffixArray(b"banana").dump_bytes()
    with pytest.raises(ValueError):
        sa.SuffixArray.load_bytes(b"bananb", bad)       # integrity check fails: the reference's InvalidData


def test_c_program_through_the_abi(tmp_path):
    """a plain C caller of the drop-in symbols: sa_amd_divsufsort has the signature of the C engine the crate's
    src/saca.rs:14 binds (T, SA, n -> 0), sa_amd_saca_u8 the contract of saca() itself (src/saca.rs:9-15)"""
    import subprocess
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include "suffix_array_amd.h"
#include <stdio.h>
#include <string.h>
int main(void) {
    const unsigned char *t = (const unsigned char *)"mississippi";
    int32_t sa[11]; uint32_t full[12];
    static const int32_t want[11] = { 10, 7, 4, 1, 0, 9, 8, 6, 3, 5, 2 };
    if (sa_amd_divsufsort(t, sa, 11) != 0) return 1;
    if (memcmp(sa, want, sizeof(want)) != 0) return 2;
    if (sa_amd_saca_u8(t, full, 11) != 0) return 3;
    if (full[0] != 11 || memcmp(full + 1, want, sizeof(want)) != 0) return 4;
    if (sa_amd_divsufsort(t, sa, 0) != 0) return 5;                 /* n = 0: success, nothing written */
    if (sa_amd_divsufsort((const unsigned char *)0, sa, 3) == 0) return 6;   /* null text: an error code, no abort */
    puts("ok");
    return 0;
}
''')
    exe = tmp_path / "caller"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", os.path.join(ROOT, "suffix_array_amd"), "-lsuffix_array_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "suffix_array_amd")])
    assert subprocess.call([str(exe)]) == 0


def test_cpp_program_through_the_mirror(tmp_path):
    """a C++ caller of include/suffix_array_amd.hpp, the host-side mirror of the crate's interface: SuffixArray::new_ / set /
    from_parts (the literal check_integrity of src/sa.rs:72-84) / enable_buckets on the doc-test text of src/lib.rs:28-29"""
    import subprocess
    src = tmp_path / "caller.cpp"
    src.write_text(r'''
#include "suffix_array_amd.hpp"
#include <cstdio>
#include <cstring>
using suffix_array::SuffixArray;
int main() {
    const char *txt = "splendid splendor";
    const auto *t = reinterpret_cast<const std::uint8_t *>(txt);
    const std::size_t n = std::strlen(txt);
    SuffixArray a = SuffixArray::new_(t, n);
    static const std::uint32_t want[18] = { 17, 8, 7, 5, 14, 3, 12, 6, 2, 11, 4, 13, 15, 1, 10, 16, 0, 9 };
    if (a.sa().size() != 18 || std::memcmp(a.sa().data(), want, sizeof(want)) != 0) return 1;
    a.enable_buckets();
    const auto &b = a.buckets();
    if (b.size() != 256 * 257 + 1 || b[0] != 1 || b.back() != 18) return 2;
    const std::size_t idx = std::size_t('s') * 257 + (std::size_t('p') + 1) + 1;     // sub-bucket ('s', 'p'), src/sa.rs:130
    if (b[idx] - b[idx - 1] != 2) return 3;                                          // "splend" occurs twice (src/lib.rs:28-29)
    std::vector<std::uint32_t> good(a.sa()), bad(a.sa());
    std::swap(bad[3], bad[4]);
    if (!SuffixArray::from_parts(t, n, good)) return 4;
    if (SuffixArray::from_parts(t, n, bad)) return 5;
    a.set(t, 6);                                                                     // "splend": set() re-runs construction, src/sa.rs:30-33
    if (a.sa().size() != 7 || a.sa()[0] != 6) return 6;
    try { suffix_array::saca(t, n, good.data(), 5); return 7; } catch (const std::logic_error &) { }   // the assert of src/saca.rs:11
    std::puts("ok");
    return 0;
}
''')
    exe = tmp_path / "caller_cpp"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", os.path.join(ROOT, "suffix_array_amd"), "-lsuffix_array_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "suffix_array_amd")])
    assert subprocess.call([str(exe)]) == 0
