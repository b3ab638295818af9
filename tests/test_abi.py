"""CPU suite: the C-ABI library loads, exports every symbol include/*.h declares, and the host
mirror of the reference interface behaves like the reference (no compute without a GPU)."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

import suffix_array_amd as sa
from conftest import KNOWN_ANSWERS, ROOT


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(sa_amd_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("sa_amd_divsufsort", "sa_amd_saca_u8", "sa_amd_saca_batch", "sa_amd_saca_device",
                 "sa_amd_workspace_bytes", "sa_amd_max_length"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(sa.library_path())
    for name in declared_symbols():
        assert hasattr(L, name), f"{name} declared in include/ but not exported"


def test_constants_and_error_strings():
    L = sa.lib()
    assert L.sa_amd_max_length() == sa.MAX_LENGTH == 2**31 - 1      # reference src/saca.rs:6
    assert L.sa_amd_strerror(0) == b"ok"
    assert b"memory" in L.sa_amd_strerror(-2)
    assert L.sa_amd_version().startswith(b"suffix_array_amd")


def test_workspace_size_grows_linearly():
    assert sa.workspace_bytes(0) > 0
    a, b = sa.workspace_bytes(1 << 20), sa.workspace_bytes(1 << 21)
    assert a < b < 2.2 * a
    assert sa.workspace_bytes(2**31 - 1) < 200 * 2**30       # fits one MI355X (288 GB) with the text and SA
    assert sa.lib().sa_amd_workspace_bytes(-1) == -1


def test_saca_preconditions_mirror_reference_asserts():
    with pytest.raises(AssertionError):                      # reference src/saca.rs:11
        sa.saca(b"abc", np.zeros(3, dtype=np.uint32))
    with pytest.raises(TypeError):
        sa.saca(b"abc", np.zeros(4, dtype=np.int64))


def test_invalid_arguments_return_codes():
    L = sa.lib()
    assert L.sa_amd_saca_u8(None, None, -1) == -1
    assert L.sa_amd_divsufsort(None, None, 5) == -1
    assert L.sa_amd_saca_device(None, None, 5, None, 0, None, None) == -1


@pytest.mark.skipif(sa.lib().sa_amd_device_count() > 0, reason="GPU present")
def test_no_gpu_fails_loudly_no_cpu_fallback():
    with pytest.raises(sa.SuffixArrayError) as e:
        sa.SuffixArray(b"banana")
    assert e.value.code == -4


def test_from_parts_checks_integrity(oracle):
    """host logic of reference src/sa.rs:57-84 on oracle-built arrays"""
    for text, expected in KNOWN_ANSWERS:
        arr = np.array(expected, dtype=np.uint32)
        obj = sa.SuffixArray.from_parts(text, arr)
        assert obj is not None and obj.len() == len(text) and obj.is_empty() == (len(text) == 0)
        s, got = obj.into_parts()
        assert got.tolist() == expected and s.tobytes() == text
    text = b"splendid splendor" * 20
    good = oracle.sais(text)
    assert sa.SuffixArray.from_parts(text, good) is not None
    bad = good.copy(); bad[10], bad[11] = bad[11], bad[10]
    assert sa.SuffixArray.from_parts(text, bad) is None
    assert sa.SuffixArray.from_parts(text, good[:-1]) is None
    shifted = good.copy(); shifted[0] = 0
    assert sa.SuffixArray.from_parts(text, shifted) is None
    assert sa.SuffixArray.unchecked_from_parts(text, bad).len() == len(text)


def test_cpp_mirror_header_compiles(tmp_path):
    """include/suffix_array_amd.hpp (C++ host mirror of the reference interface) is valid C++17"""
    import subprocess
    src = tmp_path / "t.cpp"
    src.write_text('#include "suffix_array_amd.hpp"\n'
                   'int main() { const unsigned char t[] = "banana";\n'
                   '  auto a = suffix_array::SuffixArray::from_parts(t, 6, {6, 5, 3, 1, 0, 4, 2});\n'
                   '  auto b = suffix_array::SuffixArray::from_parts(t, 6, {6, 5, 3, 1, 0, 2, 4});\n'
                   '  return (a.has_value() && !b.has_value() && a->len() == 6) ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", os.path.join(ROOT, "suffix_array_amd"), "-lsuffix_array_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "suffix_array_amd")])
    assert subprocess.call([str(exe)]) == 0
