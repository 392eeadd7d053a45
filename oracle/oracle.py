"""ctypes front-end of the plain-C oracle (``oracle/psk_oracle.c``).

TEST INFRASTRUCTURE ONLY -- the checker, never the thing measured or shipped.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; nothing under ``pyprobables_amd/`` does.

Parity status: pinned (tests/test_oracle_golden.py checks every entry point
against the reference's own known-answer vectors and against fixtures generated
from the real reference by tests/golden/gen_golden.py).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libpsk_oracle.so"

Q_MIN, Q_MEAN, Q_MEANMIN = 0, 1, 2
_QUERY = {"min": Q_MIN, None: Q_MIN, "mean": Q_MEAN, "mean-min": Q_MEANMIN}


def build(force: bool = False) -> Path:
    """compile oracle/psk_oracle.c with gcc (seconds)"""
    src = _HERE / "psk_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(
            ["gcc", "-O2", "-fPIC", "-std=gnu11", "-pthread", "-fno-strict-aliasing", "-shared", "-o", str(_LIB_PATH), str(src), "-lm"],
            check=True,
        )
    return _LIB_PATH


_lib = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        L.psk_o_fnv1a.restype = C.c_uint64
        L.psk_o_fnv1a.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.psk_o_fnv1a_u32.restype = C.c_uint64
        L.psk_o_fnv1a_u32.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.psk_o_default_fnv1a.restype = None
        L.psk_o_default_fnv1a.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.psk_o_default_fnv1a_u32.restype = None
        L.psk_o_default_fnv1a_u32.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.psk_o_bloom_params.restype = C.c_int
        L.psk_o_bloom_params.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), u64p, u64p]
        L.psk_o_cms_params.restype = None
        L.psk_o_cms_params.argtypes = [C.c_double, C.c_double, u64p, u64p]
        vp = C.c_void_p
        u64, u32, i64, ci = C.c_uint64, C.c_uint32, C.c_int64, C.c_int
        L.psk_o_bloom_add_keys.restype = None
        L.psk_o_bloom_add_keys.argtypes = [vp, u64, u32, vp, u64, u64]
        L.psk_o_bloom_check_keys.restype = None
        L.psk_o_bloom_check_keys.argtypes = [vp, u64, u32, vp, u64, u64, vp]
        L.psk_o_bloom_add_varlen.restype = None
        L.psk_o_bloom_add_varlen.argtypes = [vp, u64, u32, vp, vp, u64]
        L.psk_o_bloom_check_varlen.restype = None
        L.psk_o_bloom_check_varlen.argtypes = [vp, u64, u32, vp, vp, u64, vp]
        L.psk_o_bloom_add_hashes.restype = None
        L.psk_o_bloom_add_hashes.argtypes = [vp, u64, u32, vp, u64, u64]
        L.psk_o_bloom_check_hashes.restype = None
        L.psk_o_bloom_check_hashes.argtypes = [vp, u64, u32, vp, u64, u64, vp]
        L.psk_o_bloom_bits_set.restype = u64
        L.psk_o_stack_add_varlen.restype = ci
        L.psk_o_stack_add_varlen.argtypes = [vp, u64, u64, u64p, vp, u64, u32, u64, u64, vp, vp, u64, ci, u64p]
        L.psk_o_stack_check_varlen.restype = None
        L.psk_o_stack_check_varlen.argtypes = [vp, u64, u64, u64, u32, vp, vp, u64, vp]
        L.psk_o_bloom_bits_set.argtypes = [vp, u64]
        L.psk_o_cbf_add_alt.restype = u32
        L.psk_o_cbf_add_alt.argtypes = [vp, u64, u32, vp, u64, u64p]
        L.psk_o_cbf_check_alt.restype = u32
        L.psk_o_cbf_check_alt.argtypes = [vp, u64, u32, vp]
        L.psk_o_cbf_remove_alt.restype = u32
        L.psk_o_cbf_remove_alt.argtypes = [vp, u64, u32, vp, u64, u64p]
        L.psk_o_cbf_update_keys.restype = None
        L.psk_o_cbf_update_keys.argtypes = [vp, u64, u32, vp, u64, u64, vp, u64p, vp]
        L.psk_o_cbf_check_keys.restype = None
        L.psk_o_cbf_check_keys.argtypes = [vp, u64, u32, vp, u64, u64, vp]
        L.psk_o_cms_add_alt.restype = i64
        L.psk_o_cms_add_alt.argtypes = [vp, u64, u32, vp, i64, i64p, ci]
        L.psk_o_cms_remove_alt.restype = i64
        L.psk_o_cms_remove_alt.argtypes = [vp, u64, u32, vp, i64, i64p, ci]
        L.psk_o_cms_check_alt.restype = i64
        L.psk_o_cms_check_alt.argtypes = [vp, u64, u32, vp, i64, ci]
        L.psk_o_cms_add_keys.restype = None
        L.psk_o_cms_add_keys.argtypes = [vp, u64, u32, vp, u64, u64, vp, i64p, ci, vp]
        L.psk_o_cms_remove_keys.restype = None
        L.psk_o_cms_remove_keys.argtypes = [vp, u64, u32, vp, u64, u64, vp, i64p, ci, vp]
        L.psk_o_cms_check_keys.restype = None
        L.psk_o_cms_check_keys.argtypes = [vp, u64, u32, vp, u64, u64, i64, ci, vp]
        L.psk_o_cms_join.restype = None
        L.psk_o_cms_join.argtypes = [vp, vp, u64]
        L.psk_o_estimate_elements.restype = i64
        L.psk_o_estimate_elements.argtypes = [u64, u64, u32]
        L.psk_o_bloom_combine.restype = None
        L.psk_o_bloom_combine.argtypes = [vp, vp, vp, u64, ci]
        L.psk_o_bloom_jaccard.restype = C.c_double
        L.psk_o_bloom_jaccard.argtypes = [vp, vp, u64, u64p]
        L.psk_o_cbf_combine.restype = u64
        L.psk_o_cbf_combine.argtypes = [vp, vp, vp, u64, ci]
        L.psk_o_cbf_jaccard.restype = C.c_double
        L.psk_o_cbf_jaccard.argtypes = [vp, vp, u64, u64p]
        L.psk_o_cbf_nonzero.restype = u64
        L.psk_o_cbf_nonzero.argtypes = [vp, u64]
        L.psk_o_bloom_insert_check_mt.restype = u64
        L.psk_o_bloom_insert_check_mt.argtypes = [vp, u64, u32, u64, u64, u64, u32, C.POINTER(C.c_double)]
        L.psk_o_bloom_insert_check_mt_shared.restype = u64
        L.psk_o_bloom_insert_check_mt_shared.argtypes = [vp, u64, u32, u64, u64, u64, u32, C.POINTER(C.c_double)]
        L.psk_o_splitmix64.restype = u64
        L.psk_o_splitmix64.argtypes = [u64]
        L.psk_o_gen_keys16.restype = None
        L.psk_o_gen_keys16.argtypes = [vp, u64, u64, u64]
        L.psk_o_gen_weights.restype = None
        L.psk_o_gen_weights.argtypes = [vp, u64, u64, u64]
        _lib = L
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _as_elems(key):
    """bytes-like -> (uint8 array, False); str -> (uint32 code points, True)  [hashes.py:98]"""
    if isinstance(key, str):
        return np.array([ord(c) for c in key], dtype=np.uint32), True
    return np.frombuffer(bytes(key), dtype=np.uint8), False


# ------------------------------------------------------------------ hashing
def fnv_1a(key, seed: int = 0) -> int:
    a, wide = _as_elems(key)
    f = lib().psk_o_fnv1a_u32 if wide else lib().psk_o_fnv1a
    return int(f(_ptr(a) if a.size else None, a.size, seed))


def default_fnv_1a(key, depth: int = 1) -> list[int]:
    a, wide = _as_elems(key)
    out = np.zeros(depth, dtype=np.uint64)
    f = lib().psk_o_default_fnv1a_u32 if wide else lib().psk_o_default_fnv1a
    f(_ptr(a) if a.size else None, a.size, depth, _ptr(out))
    return [int(x) for x in out]


# ------------------------------------------------------------------- sizing
def bloom_params(est_elements, fpr):
    """-> (fpr32, k, m) or raises ValueError(code)"""
    f, k, m = C.c_double(), C.c_uint64(), C.c_uint64()
    rc = lib().psk_o_bloom_params(float(est_elements), float(fpr), C.byref(f), C.byref(k), C.byref(m))
    if rc:
        raise ValueError(rc)
    return f.value, k.value, m.value


def cms_params(confidence, error_rate):
    w, d = C.c_uint64(), C.c_uint64()
    lib().psk_o_cms_params(confidence, error_rate, C.byref(w), C.byref(d))
    return w.value, d.value


# ---------------------------------------------------------- key generators
def gen_keys16(start: int, n: int, seed: int = 0x5EED) -> np.ndarray:
    out = np.empty((n, 16), dtype=np.uint8)
    lib().psk_o_gen_keys16(_ptr(out), start, n, seed)
    return out


def gen_weights(start: int, n: int, seed: int = 0x5EED) -> np.ndarray:
    out = np.empty(n, dtype=np.int32)
    lib().psk_o_gen_weights(_ptr(out), start, n, seed)
    return out


def _keys2d(keys) -> np.ndarray:
    a = np.ascontiguousarray(keys, dtype=np.uint8)
    assert a.ndim == 2
    return a


def pack_varlen(keys) -> tuple[np.ndarray, np.ndarray]:
    """list of bytes -> (blob uint8, offsets uint64[n+1])"""
    offs = np.zeros(len(keys) + 1, dtype=np.uint64)
    np.cumsum([len(k) for k in keys], out=offs[1:])
    blob = np.frombuffer(b"".join(bytes(k) for k in keys), dtype=np.uint8).copy()
    if blob.size == 0:
        blob = np.zeros(1, dtype=np.uint8)
    return blob, offs


# -------------------------------------------------------------- structures
class OracleBloom:
    """sequential BloomFilter table: bloom.py:241-272"""

    def __init__(self, m_bits: int, k: int):
        self.m, self.k = int(m_bits), int(k)
        self.bloom = np.zeros((self.m + 7) // 8, dtype=np.uint8)
        self.els_added = 0

    def add_keys(self, keys):
        a = _keys2d(keys)
        lib().psk_o_bloom_add_keys(_ptr(self.bloom), self.m, self.k, _ptr(a), a.shape[0], a.shape[1])
        self.els_added += a.shape[0]

    def check_keys(self, keys) -> np.ndarray:
        a = _keys2d(keys)
        out = np.empty(a.shape[0], dtype=np.uint8)
        lib().psk_o_bloom_check_keys(_ptr(self.bloom), self.m, self.k, _ptr(a), a.shape[0], a.shape[1], _ptr(out))
        return out

    def add_varlen(self, keys):
        blob, offs = pack_varlen(keys)
        lib().psk_o_bloom_add_varlen(_ptr(self.bloom), self.m, self.k, _ptr(blob), _ptr(offs), len(keys))
        self.els_added += len(keys)

    def check_varlen(self, keys) -> np.ndarray:
        blob, offs = pack_varlen(keys)
        out = np.empty(len(keys), dtype=np.uint8)
        lib().psk_o_bloom_check_varlen(_ptr(self.bloom), self.m, self.k, _ptr(blob), _ptr(offs), len(keys), _ptr(out))
        return out

    def add_hashes(self, hashes):
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        lib().psk_o_bloom_add_hashes(_ptr(self.bloom), self.m, self.k, _ptr(h), h.shape[0], h.shape[1])
        self.els_added += h.shape[0]

    def check_hashes(self, hashes) -> np.ndarray:
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        out = np.empty(h.shape[0], dtype=np.uint8)
        lib().psk_o_bloom_check_hashes(_ptr(self.bloom), self.m, self.k, _ptr(h), h.shape[0], h.shape[1], _ptr(out))
        return out

    def bits_set(self) -> int:
        return int(lib().psk_o_bloom_bits_set(_ptr(self.bloom), self.bloom.size))

    def estimate_elements(self) -> int:
        """bloom.py:340-352"""
        return int(lib().psk_o_estimate_elements(self.bits_set(), self.m, self.k))

    def _combine(self, other: "OracleBloom", op: int) -> "OracleBloom":
        res = OracleBloom(self.m, self.k)
        lib().psk_o_bloom_combine(_ptr(res.bloom), _ptr(self.bloom), _ptr(other.bloom), self.bloom.size, op)
        res.els_added = res.estimate_elements()  # bloom.py:398 / :427
        return res

    def union(self, other: "OracleBloom") -> "OracleBloom":
        """bloom.py:401-428"""
        return self._combine(other, 0)

    def intersection(self, other: "OracleBloom") -> "OracleBloom":
        """bloom.py:371-399"""
        return self._combine(other, 1)

    def jaccard_index(self, other: "OracleBloom") -> float:
        """bloom.py:430-460"""
        return float(lib().psk_o_bloom_jaccard(_ptr(self.bloom), _ptr(other.bloom), self.bloom.size, None))

    def insert_check_mt(self, start: int, n: int, nthreads: int, seed: int = 0x5EED) -> int:
        """all-cores baseline leg: per-thread replica + OR merge + lookups; returns the number of keys found
        (``self.mt_seconds``: wall time of the three phases, replicas allocated and touched beforehand)"""
        sec = C.c_double(0.0)
        found = int(lib().psk_o_bloom_insert_check_mt(_ptr(self.bloom), self.m, self.k, start, n, seed, nthreads, C.byref(sec)))
        self.mt_seconds = sec.value
        return found

    def insert_check_mt_shared(self, start: int, n: int, nthreads: int, seed: int = 0x5EED) -> int:
        """all-cores baseline leg, the shape one would ship on a CPU: ONE shared table, relaxed atomic OR per bit, then lookups"""
        sec = C.c_double(0.0)
        found = int(lib().psk_o_bloom_insert_check_mt_shared(_ptr(self.bloom), self.m, self.k, start, n, seed, nthreads, C.byref(sec)))
        self.mt_seconds = sec.value
        return found


def splitmix64(x: int) -> int:
    """the stream generator of SURVEY.md 8(d)"""
    return int(lib().psk_o_splitmix64(C.c_uint64(x & 0xFFFFFFFFFFFFFFFF)))


class OracleStack:
    """sequential ExpandingBloomFilter (queue=0) / RotatingBloomFilter (queue>0): expandingbloom.py:140-184, :320-358"""

    def __init__(self, est_elements: int, false_positive_rate: float, queue: int = 0, max_filters: int = 64):
        self.est, self.fpr_given, self.queue = int(est_elements), false_positive_rate, int(queue)
        self.fpr32, self.k, self.m = bloom_params(est_elements, false_positive_rate)
        self.filter_bytes = (self.m + 7) // 8
        self.max_filters = max_filters
        self.stack = np.zeros((max_filters, self.filter_bytes), dtype=np.uint8)
        self.counts = np.zeros(max_filters, dtype=np.uint64)
        self._n = C.c_uint64(1)       # starts with one empty filter (expandingbloom.py:66-68)
        self._added = C.c_uint64(0)

    @property
    def nfilters(self) -> int:
        return self._n.value

    @property
    def els_added(self) -> int:
        return self._added.value

    def add_keys(self, keys, force: bool = False):
        blob, offs = pack_varlen(keys)
        rc = lib().psk_o_stack_add_varlen(_ptr(self.stack), self.filter_bytes, self.max_filters, C.byref(self._n),
                                          _ptr(self.counts), self.m, self.k, self.est, self.queue, _ptr(blob), _ptr(offs),
                                          len(keys), int(force), C.byref(self._added))
        if rc != 0:
            raise RuntimeError("OracleStack: max_filters exceeded")

    def check_keys(self, keys) -> np.ndarray:
        blob, offs = pack_varlen(keys)
        out = np.empty(len(keys), dtype=np.uint8)
        lib().psk_o_stack_check_varlen(_ptr(self.stack), self.filter_bytes, self.nfilters, self.m, self.k, _ptr(blob),
                                       _ptr(offs), len(keys), _ptr(out))
        return out

    def export_bytes(self) -> bytes:
        """expandingbloom.py:186-210: per filter uint64 elements_added + bit array, then the QQQf footer"""
        import struct

        parts = []
        for f in range(self.nfilters):
            parts.append(struct.pack("Q", int(self.counts[f])))
            parts.append(self.stack[f].tobytes())
        parts.append(struct.pack("QQQf", self.nfilters, self.est, self.els_added, self.fpr_given))
        return b"".join(parts)


class OracleCBF:
    """sequential CountingBloomFilter table: countingbloom.py:135-208"""

    def __init__(self, m: int, k: int):
        self.m, self.k = int(m), int(k)
        self.bloom = np.zeros(self.m, dtype=np.uint32)
        self._els = C.c_uint64(0)

    @property
    def els_added(self) -> int:
        return self._els.value

    def add_alt(self, hashes, num_els=1) -> int:
        h = np.ascontiguousarray(hashes[: self.k], dtype=np.uint64)
        return int(lib().psk_o_cbf_add_alt(_ptr(self.bloom), self.m, self.k, _ptr(h), num_els, C.byref(self._els)))

    def remove_alt(self, hashes, num_els=1) -> int:
        h = np.ascontiguousarray(hashes[: self.k], dtype=np.uint64)
        return int(lib().psk_o_cbf_remove_alt(_ptr(self.bloom), self.m, self.k, _ptr(h), num_els, C.byref(self._els)))

    def check_alt(self, hashes) -> int:
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        return int(lib().psk_o_cbf_check_alt(_ptr(self.bloom), self.m, len(h), _ptr(h)))

    def update_keys(self, keys, weights=None, want_out=False):
        """weights[i] > 0: add(key_i, w); < 0: remove(key_i, -w); None: add(key_i, 1)"""
        a = _keys2d(keys)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.int64)
        out = np.empty(a.shape[0], dtype=np.uint32) if want_out else None
        lib().psk_o_cbf_update_keys(
            _ptr(self.bloom), self.m, self.k, _ptr(a), a.shape[0], a.shape[1],
            None if w is None else _ptr(w), C.byref(self._els), None if out is None else _ptr(out),
        )
        return out

    def check_keys(self, keys) -> np.ndarray:
        a = _keys2d(keys)
        out = np.empty(a.shape[0], dtype=np.uint32)
        lib().psk_o_cbf_check_keys(_ptr(self.bloom), self.m, self.k, _ptr(a), a.shape[0], a.shape[1], _ptr(out))
        return out

    def bits_set(self) -> int:
        """countingbloom.py:302-304"""
        return int(lib().psk_o_cbf_nonzero(_ptr(self.bloom), self.m))

    def estimate_elements(self) -> int:
        """bloom.py:340-352 on the non-zero count"""
        return int(lib().psk_o_estimate_elements(self.bits_set(), self.m, self.k))

    def _combine(self, other: "OracleCBF", op: int) -> "OracleCBF":
        res = OracleCBF(self.m, self.k)
        if lib().psk_o_cbf_combine(_ptr(res.bloom), _ptr(self.bloom), _ptr(other.bloom), self.m, op):
            raise OverflowError("unsigned int is greater than maximum")  # what array('I') raises in the reference
        est = res.estimate_elements()
        res._els.value = est & 0xFFFFFFFFFFFFFFFF
        res.els_estimate = est  # (may be -1: countingbloom.py:239 / :299 store it as is)
        return res

    def union(self, other: "OracleCBF") -> "OracleCBF":
        """countingbloom.py:271-300"""
        return self._combine(other, 0)

    def intersection(self, other: "OracleCBF") -> "OracleCBF":
        """countingbloom.py:210-240"""
        return self._combine(other, 1)

    def jaccard_index(self, other: "OracleCBF") -> float:
        """countingbloom.py:242-269"""
        return float(lib().psk_o_cbf_jaccard(_ptr(self.bloom), _ptr(other.bloom), self.m, None))


class OracleCMS:
    """sequential CountMinSketch table: countminsketch.py:267-340, 429-453"""

    def __init__(self, width: int, depth: int, query="min"):
        self.width, self.depth = int(width), int(depth)
        self.bins = np.zeros(self.width * self.depth, dtype=np.int32)
        self._els = C.c_int64(0)
        self.query = _QUERY[query]

    @property
    def els_added(self) -> int:
        return self._els.value

    def add_alt(self, hashes, num_els=1) -> int:
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        return int(lib().psk_o_cms_add_alt(_ptr(self.bins), self.width, len(h), _ptr(h), num_els, C.byref(self._els), self.query))

    def remove_alt(self, hashes, num_els=1) -> int:
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        return int(lib().psk_o_cms_remove_alt(_ptr(self.bins), self.width, len(h), _ptr(h), num_els, C.byref(self._els), self.query))

    def check_alt(self, hashes) -> int:
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        return int(lib().psk_o_cms_check_alt(_ptr(self.bins), self.width, len(h), _ptr(h), self._els.value, self.query))

    def _batch(self, fn, keys, weights, want_out):
        a = _keys2d(keys)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.int32)
        out = np.empty(a.shape[0], dtype=np.int64) if want_out else None
        fn(
            _ptr(self.bins), self.width, self.depth, _ptr(a), a.shape[0], a.shape[1],
            None if w is None else _ptr(w), C.byref(self._els), self.query, None if out is None else _ptr(out),
        )
        return out

    def add_keys(self, keys, weights=None, want_out=False):
        return self._batch(lib().psk_o_cms_add_keys, keys, weights, want_out)

    def remove_keys(self, keys, weights=None, want_out=False):
        return self._batch(lib().psk_o_cms_remove_keys, keys, weights, want_out)

    def check_keys(self, keys) -> np.ndarray:
        a = _keys2d(keys)
        out = np.empty(a.shape[0], dtype=np.int64)
        lib().psk_o_cms_check_keys(_ptr(self.bins), self.width, self.depth, _ptr(a), a.shape[0], a.shape[1], self._els.value, self.query, _ptr(out))
        return out

    def join(self, other: "OracleCMS"):
        lib().psk_o_cms_join(_ptr(self.bins), _ptr(other.bins), self.bins.size)
        e = self._els.value + other._els.value
        self._els.value = max(min(e, 2**63 - 1), -(2**63))
