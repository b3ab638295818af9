"""Seeded synthetic corpora (ctypes binding of csrc/textgen.c) for BASELINE.json's configs.

C1/C2/C5 uniform bytes, C3 English-like Zipf word model, C4 DNA-like sigma=4 -- the offline
analogues of the reference's benchmark data (reference benches/utils.rs:17-45, :206-210).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _gen():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libsa_textgen.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: run __graft_entry__.build()")
        L = ctypes.CDLL(path)
        L.sa_gen_uniform.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64]
        L.sa_gen_sigma.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32]
        L.sa_gen_dna.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64]
        L.sa_gen_dna_repeats.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double]
        L.sa_gen_english.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int32]
        L.sa_gen_english.restype = ctypes.c_int32
        L.sa_gen_english_corpus.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int32, ctypes.c_double]
        L.sa_gen_english_corpus.restype = ctypes.c_int32
        _lib = L
    return _lib


def uniform(n: int, seed: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    _gen().sa_gen_uniform(out.ctypes.data, n, seed)
    return out


def sigma(n: int, seed: int, sigma: int, base: int = 0) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    _gen().sa_gen_sigma(out.ctypes.data, n, seed, sigma, base)
    return out


def dna(n: int, seed: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    _gen().sa_gen_dna(out.ctypes.data, n, seed)
    return out


def dna_repeats(n: int, seed: int, fraction: float = 0.2) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    _gen().sa_gen_dna_repeats(out.ctypes.data, n, seed, fraction)
    return out


def english(n: int, seed: int, vocab: int = 50000) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    rc = _gen().sa_gen_english(out.ctypes.data, n, seed, vocab)
    if rc:
        raise MemoryError("sa_gen_english")
    return out


def english_iid(n: int, seed: int, vocab: int = 50000) -> np.ndarray:
    """round-1 C3 model: iid Zipf words, no phrase reuse (mean LCP 9, max 29 at 32 MiB) -- kept as `c3_iid_256m`"""
    return english(n, seed, vocab)


def english_corpus(n: int, seed: int, vocab: int = 50000, dup_fraction: float = 0.08) -> np.ndarray:
    """C3: English-like text with the repeat structure of a real corpus (word bigrams, stock sentences, copied passages
    with mutations; csrc/textgen.c sa_gen_english_corpus) -- the offline stand-in for Pizza&Chili `english`
    (reference benches/utils.rs:24-45)"""
    out = np.empty(n, dtype=np.uint8)
    rc = _gen().sa_gen_english_corpus(out.ctypes.data, n, seed, vocab, dup_fraction)
    if rc:
        raise MemoryError("sa_gen_english_corpus")
    return out


#: BASELINE.md section 2 workloads: name -> (generator, n, seed)
WORKLOADS = {
    "c1_uniform_1k": (uniform, 1 << 10, 1),
    "c2_uniform_64m": (uniform, 64 << 20, 2),
    "c3_english_256m": (english_corpus, 256 << 20, 3),
    "c3_iid_256m": (english_iid, 256 << 20, 3),
    "c2_uniform_256m": (uniform, 256 << 20, 2),          # north_star's literal "256 MiB random-byte text"
    "c4_dna_1g": (dna, 1 << 30, 4),
    "c4_dna_repeats_1g": (lambda n, seed: dna_repeats(n, seed, 0.2), 1 << 30, 4),   # the harder C4 variant (planted repeats of 1-100 KiB, 1 % mutations)
    "c5_uniform_512m": (uniform, 512 << 20, 50),
}


def workload(name: str, rank: int = 0, n_override: int | None = None) -> np.ndarray:
    gen, n, seed = WORKLOADS[name]
    return gen(n_override if n_override is not None else n, seed + rank)
