"""Host side of the ``hash_function`` plugin surface (reference: ``probables/hashes.py``).

``HashFuncT = Callable[[str | bytes, int], list[int]]`` (hashes.py:10-15).  A sketch built with
``hash_function=None`` or ``default_fnv_1a`` never calls anything in this file on its data path: the
FNV-1a family is fused into the HIP kernels.  These host functions exist for ``sketch.hashes(key)``
(one key, python ints), for comparing two sketches' hash families (bloom.py:563-568) and as building
blocks for user-defined strategies, which are evaluated per key on the host and handed to the GPU as
pre-computed hashes (PSK_KEYS_HASHES).
"""

from __future__ import annotations

import hashlib
import struct
from collections.abc import Callable
from functools import wraps

KeyT = str | bytes
HashResultsT = list[int]
HashFuncT = Callable[[KeyT, int], HashResultsT]

_MASK64 = (1 << 64) - 1
_MASK32 = (1 << 32) - 1
_FNV64_BASIS, _FNV64_PRIME = 0xCBF29CE484222325, 0x100000001B3
_FNV32_BASIS, _FNV32_PRIME = 0x811C9DC5, 0x01000193


def _elements(key: KeyT):
    """a str is hashed by code point, anything else by byte value (hashes.py:98)"""
    return map(ord, key) if isinstance(key, str) else key


def fnv_1a(key: KeyT, seed: int = 0) -> int:
    """64-bit FNV-1a with offset basis ``basis + 31*seed`` (hashes.py:86-103)"""
    h = (_FNV64_BASIS + 31 * seed) & _MASK64
    for e in _elements(key):
        h = ((h ^ e) * _FNV64_PRIME) & _MASK64
    return h


def fnv_1a_32(key: KeyT, seed: int = 0) -> int:
    """32-bit FNV-1a (hashes.py:106-122)"""
    h = (_FNV32_BASIS + 31 * seed) & _MASK32
    for e in _elements(key):
        h = ((h ^ e) * _FNV32_PRIME) & _MASK32
    return h


def default_fnv_1a(key: KeyT, depth: int = 1) -> HashResultsT:
    """``depth`` independent FNV-1a passes, seeds 0..depth-1 (hashes.py:71-83).

    This object is also the *marker* for the fused GPU path: a sketch whose ``hash_function`` is this
    function (or None) hashes inside the kernel."""
    return [fnv_1a(key, seed) for seed in range(depth)]


def hash_with_depth_bytes(func) -> HashFuncT:
    """decorator: ``func(key_bytes, idx) -> digest bytes`` becomes a chained k-hash family, each round
    re-hashing the previous digest and keeping its first 8 bytes (hashes.py:18-41)"""

    @wraps(func)
    def hashing_func(key, depth=1):
        cur = key.encode("utf-8") if isinstance(key, str) else key
        out = []
        for idx in range(depth):
            cur = func(cur, idx)
            out.append(struct.unpack("Q", cur[:8])[0])
        return out

    return hashing_func


def hash_with_depth_int(func) -> HashFuncT:
    """decorator: ``func(key, idx) -> int`` becomes a chained family, round i hashing the hex text of
    round i-1 (hashes.py:44-68)"""

    @wraps(func)
    def hashing_func(key, depth=1):
        cur = func(key, 0)
        out = [cur]
        for idx in range(1, depth):
            cur = func(f"{cur:x}", idx)
            out.append(cur)
        return out

    return hashing_func


@hash_with_depth_bytes
def default_md5(key: KeyT, *args, **kwargs) -> bytes:
    """chained md5 family (hashes.py:125-136)"""
    return hashlib.md5(key).digest()


@hash_with_depth_bytes
def default_sha256(key: KeyT, *args, **kwargs) -> bytes:
    """chained sha256 family (hashes.py:139-150)"""
    return hashlib.sha256(key).digest()


_PROBE_KEY, _PROBE_DEPTH = "this is a test \u20ac", 3  # a code point > 255: FNV walks code points, the digests UTF-8 bytes


def _same_family(hash_function, ours) -> bool:
    """``hash_function`` is ``ours``, or the reference's own function of the same name (``probables.hashes.<name>``,
    what a user who switches to this package passes in: bloom.py:496-499 installs exactly that object by default).
    The foreign function is accepted by NAME and then probed, like the reference compares two filters' hash families
    through ``hashes("test")`` (bloom.py:563-568): a look-alike that disagrees stays on the per-key host route."""
    if hash_function is ours:
        return True
    mod = getattr(hash_function, "__module__", None)
    name = getattr(hash_function, "__qualname__", getattr(hash_function, "__name__", None))
    if mod != "probables.hashes" or name != ours.__name__:
        return False
    try:
        return list(hash_function(_PROBE_KEY, _PROBE_DEPTH)) == ours(_PROBE_KEY, _PROBE_DEPTH)
    except Exception:
        return False


def is_fused_fnv(hash_function) -> bool:
    """True when the kernels may compute the hashes themselves: the default FNV-1a family, ours or the reference's"""
    return hash_function is None or _same_family(hash_function, default_fnv_1a)


def device_digest(hash_function):
    """the engine's digest id (include/psk.h ``psk_digest``) when ``hash_function`` is one of the built-in digest
    families (ours or the reference's ``probables.hashes.default_md5`` / ``default_sha256``), else None: those run as
    HIP kernels instead of per key on the host"""
    if _same_family(hash_function, default_md5):
        return 0
    if _same_family(hash_function, default_sha256):
        return 1
    return None
