import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _one_byte_texts_take_the_general_path(monkeypatch):
    """a text of ONE byte value has a closed form that the pipeline takes (k_fill_descending); the suite's many one-byte texts are
    there to exercise giant groups in the sort and the rounds, so every test runs with the shortcut off -- the closed form has a
    test of its own (test_text_of_one_byte_value_takes_the_closed_form), which switches it back on"""
    monkeypatch.setenv("SA_AMD_NO_UNARY_SHORTCUT", "1")


class Oracle:
    """ctypes view of oracle/liboracle.so -- the CPU checker (never the thing under test)."""

    def __init__(self):
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "all"], cwd=os.path.join(ROOT, "oracle"))
        L = ctypes.CDLL(path)
        for fn in ("oracle_naive_sa", "oracle_sais"):
            getattr(L, fn).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
            getattr(L, fn).restype = ctypes.c_int32
        for fn in ("oracle_check_integrity", "oracle_verify_sa"):
            getattr(L, fn).argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
            getattr(L, fn).restype = ctypes.c_int32
        L.oracle_verify_sa_mt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]
        L.oracle_verify_sa_mt.restype = ctypes.c_int32
        L.oracle_divsufsort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
        L.oracle_divsufsort.restype = ctypes.c_int32
        L.oracle_bucket_table.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        self.L = L

    @staticmethod
    def _u8(s):
        if isinstance(s, np.ndarray):
            return np.ascontiguousarray(s, dtype=np.uint8)
        return np.frombuffer(bytes(s), dtype=np.uint8) if len(s) else np.zeros(0, dtype=np.uint8)

    def naive(self, s):
        t = self._u8(s)
        sa = np.zeros(t.size + 1, dtype=np.uint32)
        assert self.L.oracle_naive_sa(t.ctypes.data, sa.ctypes.data, t.size) == 0
        return sa

    def sais(self, s):
        t = self._u8(s)
        sa = np.zeros(t.size + 1, dtype=np.uint32)
        assert self.L.oracle_sais(t.ctypes.data, sa.ctypes.data, t.size) == 0
        return sa

    def check_integrity(self, s, sa):
        t = self._u8(s)
        a = np.ascontiguousarray(sa, dtype=np.uint32)
        return self.L.oracle_check_integrity(t.ctypes.data, t.size, a.ctypes.data, a.size)

    def verify(self, s, sa):
        t = self._u8(s)
        a = np.ascontiguousarray(sa, dtype=np.uint32)
        return self.L.oracle_verify_sa(t.ctypes.data, t.size, a.ctypes.data, a.size)

    def verify_mt(self, s, sa, threads=16):
        """oracle_verify_sa on several host threads (the 512 MiB / 1 GiB configs)"""
        t = self._u8(s)
        a = np.ascontiguousarray(sa, dtype=np.uint32)
        return self.L.oracle_verify_sa_mt(t.ctypes.data, t.size, a.ctypes.data, a.size, threads)

    def bucket_table(self, s):
        t = self._u8(s)
        b = np.zeros(256 * 257 + 1, dtype=np.uint32)
        self.L.oracle_bucket_table(t.ctypes.data, t.size, b.ctypes.data)
        return b


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


def fibonacci_word(k):
    a, b = b"a", b"ab"
    for _ in range(k):
        a, b = b, b + a
    return b


def thue_morse(n):
    return bytes(bin(i).count("1") & 1 for i in range(n))


def de_bruijn(k, n):
    a = [0] * k * n
    seq = []

    def db(t, p):
        if t > n:
            if n % p == 0:
                seq.extend(a[1:p + 1])
        else:
            a[t] = a[t - p]
            db(t + 1, p)
            for j in range(a[t - p] + 1, k):
                a[t] = j
                db(t + 1, t)
    db(1, 1)
    return bytes(seq)


def adversarial_cases():
    """inputs the reference's random tests do not reach."""
    cases = {
        "empty": b"", "one": b"a", "two_eq": b"aa", "two_inc": b"ab", "two_dec": b"ba", "three": b"aba",
        "zeros": b"\x00" * 777, "ffs": b"\xff" * 513, "zeros_ffs": b"\x00" * 300 + b"\xff" * 300,
        "ffs_zeros": b"\xff" * 257 + b"\x00" * 255, "mix00ff": (b"\x00\xff" * 400) + b"\x00" * 7,
        "a_run": b"a" * 4097, "ab_period": b"ab" * 2049, "abc_period": b"abc" * 1366,
        "ramp_up": bytes(range(256)), "ramp_down": bytes(range(255, -1, -1)),
        "ramp_up_rep": bytes(range(256)) * 9, "fib": fibonacci_word(16), "thue_morse": thue_morse(6000),
        "de_bruijn_2_12": de_bruijn(2, 12), "de_bruijn_4_6": de_bruijn(4, 6),
        "tail_zeros": b"abc" * 50 + b"\x00" * 70, "banana": b"banana", "mississippi": b"mississippi",
    }
    for k in (63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097):
        cases[f"run_{k}"] = b"b" * k + b"a" + b"b" * k
        cases[f"n_{k}"] = bytes((i * 7 + 3) & 0xFF for i in range(k))
    return cases


KNOWN_ANSWERS = [   # SURVEY.md section 8a; the last one agrees with the doc-test at reference src/lib.rs:28-29
    (b"", [0]), (b"a", [1, 0]), (b"aa", [2, 1, 0]), (b"banana", [6, 5, 3, 1, 0, 4, 2]),
    (b"mississippi", [11, 10, 7, 4, 1, 0, 9, 8, 6, 3, 5, 2]), (b"\xff\x00\xff", [3, 1, 2, 0]),
    (b"splendid splendor", [17, 8, 7, 5, 14, 3, 12, 6, 2, 11, 4, 13, 15, 1, 10, 16, 0, 9]),
]
