"""Pure-Python mirror of the reference's per-key hot path.  TEST INFRASTRUCTURE ONLY -- not product code.

A from-scratch restatement, in interpreted Python with arbitrary-precision ints (the way the reference itself runs),
of ``default_fnv_1a`` (probables/hashes.py:71-103) and ``BloomFilter.add_alt`` / ``check_alt`` (probables/blooms/
bloom.py:241-272).  It exists for ONE purpose: ``bench.py``'s ``cpu_baseline.legs.python_mirror`` -- the number that
speaks for what the reference's interpreted per-key loop costs on the GPU box's host cores, since the
reference itself cannot travel there.  Pinned by tests/test_oracle_golden.py::test_python_mirror_* against fixtures
generated from the real reference.  Only tests/ and bench.py's cpu_baseline leg import it.
"""

from __future__ import annotations

_M64 = 0xFFFFFFFFFFFFFFFF
_BASIS = 14695981039346656037   # hashes.py:96
_PRIME = 1099511628211          # hashes.py:97


def fnv_1a(key, seed: int = 0) -> int:
    """hashes.py:86-103: a str is walked by code point, anything else by byte value"""
    hval = (_BASIS + 31 * seed) & _M64
    for e in (map(ord, key) if isinstance(key, str) else key):
        hval ^= e
        hval = (hval * _PRIME) & _M64
    return hval


def default_fnv_1a(key, depth: int = 1) -> list:
    """hashes.py:71-83"""
    return [fnv_1a(key, i) for i in range(depth)]


class MirrorBloom:
    """bloom.py:241-272 on a bytearray (bit b -> byte b // 8, mask 1 << (b % 8))"""

    def __init__(self, m_bits: int, k: int):
        self.m, self.k = m_bits, k
        self.bloom = bytearray((m_bits + 7) // 8)
        self.els_added = 0

    def add(self, key) -> None:
        for h in default_fnv_1a(key, self.k):          # bloom.py:234-239 -> add_alt
            b = h % self.m                              # :247
            self.bloom[b // 8] |= 1 << (b % 8)          # :248-249
        self.els_added += 1                             # :250

    def check(self, key) -> bool:
        for h in default_fnv_1a(key, self.k):          # bloom.py:252-259 -> check_alt
            b = h % self.m
            if (self.bloom[b // 8] & (1 << (b % 8))) == 0:   # :269-271 early exit
                return False
        return True


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def key16(i: int, seed: int = 0x5EED) -> bytes:
    """SURVEY.md 8(d) synthetic key i"""
    return splitmix64(seed + 2 * i).to_bytes(8, "little") + splitmix64(seed + 2 * i + 1).to_bytes(8, "little")
