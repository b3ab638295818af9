"""suffix_array_amd -- host-side mirror of the reference's construction interface.

The product is the C-ABI library ``libsuffix_array_amd.so`` (include/suffix_array_amd.h); this
package is the thin Python binding used by tests and bench.py.  Names, argument meaning and
error behaviour follow the reference crate for the one path that is in scope:

    SuffixArray.new / set / len / is_empty / into_parts / from_parts / unchecked_from_parts
        reference src/sa.rs:23-70, integrity check src/sa.rs:72-84
    saca(s, sa), MAX_LENGTH
        reference src/saca.rs:6-15

There is no CPU fallback: if the HIP library is missing or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

__all__ = ["MAX_LENGTH", "saca", "SuffixArray", "SuffixArrayError", "lib", "diag_lib", "library_path", "Stats", "last_host_timing",
           "saca_batch", "workspace_bytes", "device_pci_bus_id", "saca_device_ptr", "bucket_table", "check_integrity", "last_stats", "DeviceIndex", "pack", "unpack"]

#: reference src/saca.rs:6
MAX_LENGTH = 2**31 - 1

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libsuffix_array_amd.so"
_lib: Optional[ctypes.CDLL] = None


class SuffixArrayError(RuntimeError):
    """Engine failure (the reference panics on a non-zero return of its C engine)."""

    def __init__(self, code: int, what: str):
        super().__init__(f"suffix_array_amd: {what} (status {code})")
        self.code = code


class Stats(ctypes.Structure):
    """sa_amd_stats of include/suffix_array_amd.h"""
    _fields_ = [("sigma", ctypes.c_int32), ("bits_per_symbol", ctypes.c_int32),
                ("symbols_per_key", ctypes.c_int32), ("rounds", ctypes.c_int32),
                ("sort_passes", ctypes.c_int32), ("sparse_mode", ctypes.c_int32),
                ("sorted_elements", ctypes.c_int64), ("unresolved_after_initial", ctypes.c_int64),
                ("text_rounds", ctypes.c_int32), ("top32_first", ctypes.c_int32), ("locally_sorted", ctypes.c_int64),
                ("readbacks", ctypes.c_int32), ("reserved", ctypes.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def library_path() -> str:
    return os.path.join(_HERE, _LIB_NAME)


def lib() -> ctypes.CDLL:
    """Load the C-ABI library (built in-tree by __graft_entry__.build()); fail loudly if absent."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)")
        L = ctypes.CDLL(path)
        c_u8p, c_vp = ctypes.c_void_p, ctypes.c_void_p
        L.sa_amd_max_length.restype = ctypes.c_int32
        L.sa_amd_divsufsort.argtypes = [c_u8p, c_vp, ctypes.c_int32]
        L.sa_amd_divsufsort.restype = ctypes.c_int32
        L.sa_amd_saca_u8.argtypes = [c_u8p, c_vp, ctypes.c_int32]
        L.sa_amd_saca_u8.restype = ctypes.c_int32
        L.sa_amd_saca_batch.argtypes = [c_vp, c_vp, c_vp, c_vp, ctypes.c_int32, c_vp]
        L.sa_amd_saca_batch.restype = ctypes.c_int32
        L.sa_amd_workspace_bytes.argtypes = [ctypes.c_int32]
        L.sa_amd_workspace_bytes.restype = ctypes.c_int64
        L.sa_amd_saca_device.argtypes = [c_vp, c_vp, ctypes.c_int32, c_vp, ctypes.c_int64, c_vp, c_vp]
        L.sa_amd_saca_device.restype = ctypes.c_int32
        L.sa_amd_device_count.restype = ctypes.c_int32
        L.sa_amd_device_pci_bus_id.argtypes = [ctypes.c_int32, ctypes.c_char_p, ctypes.c_int32]
        L.sa_amd_device_pci_bus_id.restype = ctypes.c_int32
        L.sa_amd_bucket_table_device.argtypes = [c_vp, c_vp, ctypes.c_int32, c_vp, c_vp]
        L.sa_amd_bucket_table_device.restype = ctypes.c_int32
        L.sa_amd_last_stats.argtypes = [c_vp]
        L.sa_amd_index_create.argtypes = [c_vp, ctypes.c_int32, c_vp, ctypes.POINTER(ctypes.c_void_p)]
        L.sa_amd_index_create.restype = ctypes.c_int32
        L.sa_amd_index_destroy.argtypes = [c_vp]
        L.sa_amd_index_destroy.restype = None
        for fn in ("sa_amd_index_sa", "sa_amd_index_buckets"):
            getattr(L, fn).argtypes = [c_vp, c_vp]
            getattr(L, fn).restype = ctypes.c_int32
        L.sa_amd_index_check_integrity.argtypes = [c_vp]
        L.sa_amd_index_check_integrity.restype = ctypes.c_int32
        L.sa_amd_index_search.argtypes = [c_vp, c_vp, c_vp, ctypes.c_int32, c_vp, c_vp, c_vp, c_vp, c_vp]
        L.sa_amd_index_search.restype = ctypes.c_int32
        L.sa_amd_pack_bound.argtypes = [ctypes.c_int64]
        L.sa_amd_pack_bound.restype = ctypes.c_int64
        L.sa_amd_pack.argtypes = [c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp]
        L.sa_amd_pack.restype = ctypes.c_int32
        L.sa_amd_unpack.argtypes = [c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp]
        L.sa_amd_unpack.restype = ctypes.c_int32
        L.sa_amd_bucket_table.argtypes = [c_vp, ctypes.c_int32, c_vp, c_vp]
        L.sa_amd_bucket_table.restype = ctypes.c_int32
        L.sa_amd_saca_u8_buckets.argtypes = [c_vp, c_vp, ctypes.c_int32, c_vp]
        L.sa_amd_saca_u8_buckets.restype = ctypes.c_int32
        L.sa_amd_check_integrity.argtypes = [c_vp, ctypes.c_int32, c_vp, ctypes.c_int64]
        L.sa_amd_check_integrity.restype = ctypes.c_int32
        L.sa_amd_last_stats.restype = None
        L.sa_amd_strerror.argtypes = [ctypes.c_int32]
        L.sa_amd_strerror.restype = ctypes.c_char_p
        L.sa_amd_version.restype = ctypes.c_char_p
        L.sa_amd_profile_begin.restype = None
        L.sa_amd_profile_end.argtypes = [c_vp, c_vp, c_vp, ctypes.c_int32]
        L.sa_amd_profile_end.restype = ctypes.c_int32
        L.sa_amd_profile_kernel_name.argtypes = [ctypes.c_int32]
        L.sa_amd_profile_kernel_name.restype = ctypes.c_char_p
        L.sa_amd_last_host_timing.argtypes = [c_vp, ctypes.c_int32]
        L.sa_amd_last_host_timing.restype = ctypes.c_int32
        L.sa_amd_profile_begin_classes.argtypes = [ctypes.c_uint64]
        L.sa_amd_profile_begin_classes.restype = None
        L.sa_amd_check_integrity_device.argtypes = [c_vp, ctypes.c_int32, c_vp, c_vp, ctypes.c_int64, c_vp]
        L.sa_amd_check_integrity_device.restype = ctypes.c_int32
        L.sa_amd_check_integrity_work_bytes.argtypes = [ctypes.c_int32]
        L.sa_amd_check_integrity_work_bytes.restype = ctypes.c_int64
        _lib = L
    return _lib


_diag: Optional[ctypes.CDLL] = None


def diag_lib() -> ctypes.CDLL:
    """libsuffix_array_amd_diag.so (csrc/sa_diag.h): the same sources built with -DSA_AMD_DIAG -- primitive test hooks,
    phase stamps and the timing ablations.  For tests/ and tools/ only; nothing in the product path loads it."""
    global _diag
    if _diag is None:
        path = os.path.join(_HERE, "libsuffix_array_amd_diag.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: run __graft_entry__.build()")
        L = ctypes.CDLL(path)
        c_vp = ctypes.c_void_p
        L.sa_amd_test_sort_pairs.argtypes = [c_vp, c_vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
        L.sa_amd_test_sort_pairs.restype = ctypes.c_int32
        L.sa_amd_test_sort_pairs32.argtypes = [c_vp, c_vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
        L.sa_amd_test_sort_pairs32.restype = ctypes.c_int32
        L.sa_amd_test_sample_sort64.argtypes = [c_vp, c_vp, ctypes.c_int64, ctypes.c_int32, c_vp]
        L.sa_amd_test_sample_sort64.restype = ctypes.c_int32
        L.sa_amd_test_bucket_sort32.argtypes = [c_vp, c_vp, ctypes.c_int64, ctypes.c_int32, c_vp]
        L.sa_amd_test_bucket_sort32.restype = ctypes.c_int32
        L.sa_amd_test_build_keys.argtypes = [c_vp, ctypes.c_int32, c_vp, c_vp, c_vp]
        L.sa_amd_test_build_keys.restype = ctypes.c_int32
        L.sa_amd_debug_phase_cycles.argtypes = [c_vp, ctypes.c_int32]
        L.sa_amd_debug_phase_cycles.restype = ctypes.c_int32
        L.sa_amd_debug_group_sort_stamps.argtypes = [ctypes.c_int32]
        L.sa_amd_debug_group_sort_stamps.restype = ctypes.c_int32
        L.sa_amd_debug_sort_variant_count.restype = ctypes.c_int32
        L.sa_amd_debug_sort_variant_name.argtypes = [ctypes.c_int32]
        L.sa_amd_debug_sort_variant_name.restype = ctypes.c_char_p
        L.sa_amd_saca_u8.argtypes = [c_vp, c_vp, ctypes.c_int32]
        L.sa_amd_saca_u8.restype = ctypes.c_int32
        L.sa_amd_saca_device.argtypes = [c_vp, c_vp, ctypes.c_int32, c_vp, ctypes.c_int64, c_vp, c_vp]
        L.sa_amd_saca_device.restype = ctypes.c_int32
        L.sa_amd_workspace_bytes.argtypes = [ctypes.c_int32]
        L.sa_amd_workspace_bytes.restype = ctypes.c_int64
        L.sa_amd_version.restype = ctypes.c_char_p
        _diag = L
    return _diag


def device_pci_bus_id(device: int = 0) -> str:
    """PCI address of HIP device `device` ("0000:c1:00.0"): which physical GPU an ordinal is"""
    buf = ctypes.create_string_buffer(64)
    _check(lib().sa_amd_device_pci_bus_id(int(device), buf, 64))
    return buf.value.decode()


def _check(code: int) -> None:
    if code != 0:
        raise SuffixArrayError(code, lib().sa_amd_strerror(code).decode())


def _as_u8(s) -> np.ndarray:
    if isinstance(s, np.ndarray):
        if s.dtype != np.uint8 or not s.flags.c_contiguous:
            raise TypeError("text must be a contiguous uint8 array")
        return s
    return np.frombuffer(bytes(s), dtype=np.uint8) if len(s) else np.zeros(0, dtype=np.uint8)


def saca(s, sa: np.ndarray) -> None:
    """``pub fn saca(s: &[u8], sa: &mut [u32])`` -- reference src/saca.rs:9-15.

    ``sa`` is a caller-owned uint32 array of ``len(s) + 1`` entries whose prior contents are
    irrelevant; on return ``sa[0] == len(s)`` and ``sa[1:]`` holds the sorted suffix offsets.
    The reference's two ``assert!``s (src/saca.rs:10-11) are AssertionErrors here.
    """
    t = _as_u8(s)
    assert t.size <= MAX_LENGTH                       # src/saca.rs:10
    assert t.size + 1 == sa.size                      # src/saca.rs:11
    if sa.dtype != np.uint32 or not sa.flags.c_contiguous or not sa.flags.writeable:
        raise TypeError("sa must be a writable contiguous uint32 array")
    _check(lib().sa_amd_saca_u8(t.ctypes.data, sa.ctypes.data, t.size))


def divsufsort(s, sa: np.ndarray) -> None:
    """The C engine's own signature (n int32 entries, no sentinel) -- call site reference src/saca.rs:14."""
    t = _as_u8(s)
    assert t.size == sa.size and sa.dtype == np.int32
    _check(lib().sa_amd_divsufsort(t.ctypes.data, sa.ctypes.data, t.size))


def saca_batch(texts, devices=None):
    """Independent texts, one device each; returns the list of uint32 arrays."""
    ts = [_as_u8(t) for t in texts]
    cnt = len(ts)
    if cnt >= 64:
        # many texts: the arrays are views of ONE allocation and the pointer tables come out of numpy -- a ctypes pointer per
        # text costs more than the library needs to build a small text (k_small_sa_batch: 0.04-1.6 us per text)
        sizes = np.fromiter((t.size for t in ts), dtype=np.int64, count=cnt)
        offs = np.zeros(cnt + 1, dtype=np.int64)
        np.cumsum(sizes + 1, out=offs[1:])
        buf = np.empty(int(offs[-1]), dtype=np.uint32)
        outs = [buf[a:b] for a, b in zip(offs[:-1].tolist(), offs[1:].tolist())]
        tp = np.fromiter((t.__array_interface__["data"][0] for t in ts), dtype=np.uint64, count=cnt)
        sp = (np.uint64(buf.ctypes.data) + (offs[:-1] * 4).astype(np.uint64)).astype(np.uint64)
        nn = sizes.astype(np.int32)
        dd = np.asarray(devices, dtype=np.int32) if devices is not None else None
        st = np.zeros(cnt, dtype=np.int32)
        rc = lib().sa_amd_saca_batch(tp.ctypes.data, sp.ctypes.data, nn.ctypes.data, dd.ctypes.data if dd is not None else None, cnt,
                                     st.ctypes.data)
        _check(rc)
        return outs
    outs = [np.empty(t.size + 1, dtype=np.uint32) for t in ts]
    T = (ctypes.c_void_p * cnt)(*[t.ctypes.data for t in ts])
    S = (ctypes.c_void_p * cnt)(*[o.ctypes.data for o in outs])
    N = (ctypes.c_int32 * cnt)(*[t.size for t in ts])
    D = (ctypes.c_int32 * cnt)(*devices) if devices is not None else None
    st = (ctypes.c_int32 * cnt)()
    rc = lib().sa_amd_saca_batch(T, S, N, D, cnt, st)
    _check(rc)
    return outs


def last_host_timing() -> dict:
    """wall-clock phases (ms) of this thread's most recent host-pointer build (sa_amd_last_host_timing)"""
    v = (ctypes.c_double * 9)()
    lib().sa_amd_last_host_timing(v, 9)
    return {"acquire": v[0], "h2d": v[1], "build": v[2], "d2h": v[3], "release": v[4], "total": v[5], "staged_threads": int(v[6]),
            "early_fraction": v[7], "workspace_bytes_in_host_memory": int(v[8])}


def last_stats() -> dict:
    """statistics of the most recent build issued by this thread"""
    st = Stats()
    lib().sa_amd_last_stats(ctypes.byref(st))
    return st.as_dict()


BUCKET_TABLE_LEN = 256 * 257 + 1        # reference src/sa.rs:95


def bucket_table(s, sa: np.ndarray | None = None) -> np.ndarray:
    """the table `enable_buckets` builds (reference src/sa.rs:89-119), computed on the GPU the way the reference computes
    it: bigram counts of the TEXT + prefix sum (src/sa.rs:96-116).  `sa` is not needed (and not uploaded); it is accepted for
    callers of the earlier form and only its length is checked"""
    t = _as_u8(s)
    if sa is not None:
        assert np.asarray(sa).size == t.size + 1
    bkt = np.empty(BUCKET_TABLE_LEN, dtype=np.uint32)
    _check(lib().sa_amd_bucket_table(t.ctypes.data, t.size, None, bkt.ctypes.data))
    return bkt


def check_integrity(s, sa: np.ndarray) -> bool:
    """`check_integrity` (reference src/sa.rs:72-84) on the GPU; raises IndexError where the
    reference panics (an entry beyond the text)"""
    t = _as_u8(s)
    a = np.ascontiguousarray(sa, dtype=np.uint32)
    rc = lib().sa_amd_check_integrity(t.ctypes.data, t.size, a.ctypes.data, a.size)
    if rc == -6:
        raise IndexError("suffix offset out of range (the reference panics here, src/sa.rs:77-78)")
    if rc < 0:
        _check(rc)
    return rc == 1


def pack(sa: np.ndarray) -> bytes:
    """`PackedSuffixArray::from_sa(..).dump_bytes()` -- reference src/packed_sa.rs:17-53, :99-106 (bit packing on the GPU)"""
    a = np.ascontiguousarray(sa, dtype=np.uint32)
    cap = int(lib().sa_amd_pack_bound(a.size))
    out = np.empty(cap, dtype=np.uint8)
    n_out = ctypes.c_int64()
    _check(lib().sa_amd_pack(a.ctypes.data, a.size, out.ctypes.data, cap, ctypes.byref(n_out)))
    return out[:n_out.value].tobytes()


def unpack(blob: bytes) -> np.ndarray:
    """`PackedSuffixArray::load_bytes(..).into_sa()` -- reference src/packed_sa.rs:55-88, :117-124;
    ValueError where the reference returns an InvalidData error"""
    b = np.frombuffer(blob, dtype=np.uint8)
    if b.size < 16:
        raise ValueError("packed suffix array: truncated header")
    length = int(np.frombuffer(blob[4:8], dtype="<u4")[0])
    out = np.empty(max(length, 1), dtype=np.uint32)
    got = ctypes.c_int64()
    rc = lib().sa_amd_unpack(b.ctypes.data, b.size, out.ctypes.data, out.size, ctypes.byref(got))
    if rc == -1:
        raise ValueError("packed suffix array: invalid data")
    _check(rc)
    return out[:got.value]


class DeviceIndex:
    """Text + suffix array resident in HBM (sa_amd_index of include/suffix_array_amd.h): batched
    `contains` / `search_all` / `search_lcp` (reference src/sa.rs:164-253), bucket table, integrity check.
    `sa=None` builds the array on the device (SuffixArray::new without downloading it)."""

    def __init__(self, s, sa: Optional[np.ndarray] = None):
        self._s = _as_u8(s)
        h = ctypes.c_void_p()
        a = None if sa is None else np.ascontiguousarray(sa, dtype=np.uint32)
        if a is not None:
            assert a.size == self._s.size + 1
        _check(lib().sa_amd_index_create(self._s.ctypes.data, self._s.size, None if a is None else a.ctypes.data,
                                         ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().sa_amd_index_destroy(self._h)
            self._h = None

    __del__ = close

    def suffix_array(self) -> np.ndarray:
        out = np.empty(self._s.size + 1, dtype=np.uint32)
        _check(lib().sa_amd_index_sa(self._h, out.ctypes.data))
        return out

    def buckets(self) -> np.ndarray:
        bkt = np.empty(BUCKET_TABLE_LEN, dtype=np.uint32)
        _check(lib().sa_amd_index_buckets(self._h, bkt.ctypes.data))
        return bkt

    def check_integrity(self) -> bool:
        rc = lib().sa_amd_index_check_integrity(self._h)
        if rc < 0 and rc != -6:
            _check(rc)
        return rc == 1

    def search(self, patterns):
        """-> dict of arrays over the patterns: contains (bool), lo/hi (search_all == sa[lo:hi]),
        lcp_start/lcp_len (search_lcp == start..start+len)"""
        pats = [bytes(p) for p in patterns]
        cnt = len(pats)
        off = np.zeros(cnt + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(p) for p in pats])
        data = np.frombuffer(b"".join(pats), dtype=np.uint8) if off[-1] else np.zeros(1, dtype=np.uint8)
        c = np.zeros(cnt, dtype=np.uint8)
        lo, hi, ls, ll = (np.zeros(cnt, dtype=np.uint32) for _ in range(4))
        _check(lib().sa_amd_index_search(self._h, data.ctypes.data, off.ctypes.data, cnt, c.ctypes.data, lo.ctypes.data,
                                         hi.ctypes.data, ls.ctypes.data, ll.ctypes.data))
        return {"contains": c.astype(bool), "lo": lo, "hi": hi, "lcp_start": ls, "lcp_len": ll}


def workspace_bytes(n: int) -> int:
    return int(lib().sa_amd_workspace_bytes(n))


def saca_device_ptr(text_ptr: int, sa_ptr: int, n: int, work_ptr: int, work_bytes: int, stream: int = 0,
                    stats: Optional[Stats] = None) -> None:
    """Device-resident build (raw device pointers, e.g. torch ``tensor.data_ptr()``)."""
    _check(lib().sa_amd_saca_device(text_ptr, sa_ptr, n, work_ptr, work_bytes, stream,
                                    ctypes.byref(stats) if stats is not None else None))


def _check_integrity(s: np.ndarray, sa: np.ndarray) -> bool:
    """reference src/sa.rs:72-84, in its linear-time equivalent form:
    a length check, then every adjacent pair must be strictly increasing as byte slices."""
    n = s.size
    if n + 1 != sa.size:                              # src/sa.rs:73-75
        return False
    if n == 0:
        return int(sa[0]) == 0                        # sa[0] > n would panic in the reference
    if sa.max() > n:
        raise IndexError("suffix offset out of range (the reference panics here, src/sa.rs:77-78)")
    if n <= 64:                                       # literal form for tiny inputs
        b = s.tobytes()
        return all(b[int(sa[i - 1]):] < b[int(sa[i]):] for i in range(1, n + 1))
    if int(sa[0]) != n:
        return False
    rank = np.full(n + 1, -1, dtype=np.int64)
    rank[sa] = np.arange(n + 1)
    if (rank < 0).any():
        return False
    a, b = sa[1:-1].astype(np.int64), sa[2:].astype(np.int64)
    ca, cb = s[a], s[b]
    ok = (ca < cb) | ((ca == cb) & (rank[a + 1] < rank[b + 1]))
    return bool(ok.all())


class SuffixArray:
    """Mirror of ``SuffixArray<'a>`` for the construction path (reference src/sa.rs:13-70)."""

    def __init__(self, s):
        """``SuffixArray::new`` -- reference src/sa.rs:23-27."""
        self._s = _as_u8(s)
        self._sa = np.zeros(self._s.size + 1, dtype=np.uint32)     # vec![0; s.len() + 1]
        saca(self._s, self._sa)
        self._bkt = None
        self._ix = None

    @classmethod
    def new(cls, s) -> "SuffixArray":
        return cls(s)

    def set(self, s) -> None:
        """``SuffixArray::set`` -- reference src/sa.rs:30-33 (like the reference it re-runs
        construction into the resized buffer and leaves the stored text and buckets alone)."""
        t = _as_u8(s)
        self._sa = np.resize(self._sa, t.size + 1)
        saca(t, self._sa)
        self._ix = None                                # the device-resident index (a cache of this object) held the old array

    def fit(self) -> None:                             # src/sa.rs:36-38 (shrink_to_fit: numpy arrays carry no slack)
        self._sa = np.ascontiguousarray(self._sa)

    def as_ref(self) -> np.ndarray:                    # AsRef<[u8]>, src/sa.rs:370-374
        return self._s

    def len(self) -> int:                              # src/sa.rs:41-43
        return int(self._s.size)

    def is_empty(self) -> bool:                        # src/sa.rs:46-48
        return self.len() == 0

    def into_parts(self):                              # src/sa.rs:51-53
        return self._s, self._sa

    @classmethod
    def from_parts(cls, s, sa) -> Optional["SuffixArray"]:
        """reference src/sa.rs:57-64: compose and check integrity; None when the check fails."""
        obj = cls.unchecked_from_parts(s, sa)
        if lib().sa_amd_device_count() > 0:
            ok = check_integrity(obj._s, obj._sa)             # HIP kernels k_ci_scatter / k_ci_check
        else:
            ok = _check_integrity(obj._s, obj._sa)            # host glue when no device is visible
        return obj if ok else None

    @classmethod
    def unchecked_from_parts(cls, s, sa) -> "SuffixArray":   # src/sa.rs:68-70
        obj = cls.__new__(cls)
        obj._s = _as_u8(s)
        obj._sa = np.ascontiguousarray(sa, dtype=np.uint32)
        obj._bkt = None
        obj._ix = None
        return obj

    # feature `pack`: reference src/sa.rs:255-361
    def dump_bytes(self) -> bytes:
        return pack(self._sa)

    def dump(self, file) -> None:
        file.write(self.dump_bytes())

    @classmethod
    def load_bytes(cls, s, blob: bytes) -> "SuffixArray":
        """reference src/sa.rs:349-361: unpack, then check the integrity; ValueError = the reference's InvalidData"""
        obj = cls.from_parts(s, unpack(blob))
        if obj is None:
            raise ValueError("inconsistent suffix array")
        return obj

    @classmethod
    def load(cls, s, file) -> "SuffixArray":
        return cls.load_bytes(s, file.read())

    def enable_buckets(self) -> None:
        """reference src/sa.rs:89-119; a no-op when the table exists (src/sa.rs:90-92)"""
        if self._bkt is None:
            self._bkt = bucket_table(self._s, self._sa)
            if getattr(self, "_ix", None) is not None:
                self._ix.buckets()                     # the resident index now narrows its searches (get_bucket, src/sa.rs:123-144)

    def buckets(self) -> Optional[np.ndarray]:
        return self._bkt

    # search: reference src/sa.rs:164-253, one pattern per call as in the reference (a batch of one on
    # the device-resident index; use DeviceIndex.search for many patterns)
    def _index(self) -> "DeviceIndex":
        if getattr(self, "_ix", None) is None:
            self._ix = DeviceIndex(self._s, self._sa)
            if self._bkt is not None:
                self._ix.buckets()
        return self._ix

    def contains(self, pat) -> bool:
        return bool(self._index().search([pat])["contains"][0])

    def search_all(self, pat) -> np.ndarray:
        r = self._index().search([pat])
        return self._sa[int(r["lo"][0]):int(r["hi"][0])]

    def search_lcp(self, pat) -> range:
        r = self._index().search([pat])
        return range(int(r["lcp_start"][0]), int(r["lcp_start"][0]) + int(r["lcp_len"][0]))

    def __array__(self, dtype=None):                   # From<SuffixArray> for Vec<u32>, src/sa.rs:364-368
        return self._sa if dtype is None else self._sa.astype(dtype)
