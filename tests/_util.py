"""shared helpers for the test-suite (no reference, no product imports)"""

import hashlib

import numpy as np


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def unpackbits(hexstr: str, n: int) -> np.ndarray:
    """LSB-first packed bits (hex) -> uint8[n] of 0/1"""
    raw = np.frombuffer(bytes.fromhex(hexstr), dtype=np.uint8)
    return np.unpackbits(raw, bitorder="little")[:n]


def as_key(case):
    """golden hash case -> python key object"""
    return case["key"] if case["type"] == "str" else bytes.fromhex(case["key"])
