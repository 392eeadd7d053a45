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


import pytest  # noqa: E402


def knob(name: str, value: int) -> None:
    """set an option that exists only in the bench build (-DPSK_BENCH_KNOBS=1: the A/B switches of experiments that were measured and dropped);
    the shipped library answers "unknown option" and the test is skipped -- it exercised a path nothing selects any more"""
    from pyprobables_amd import _native as N

    try:
        N.set_option(name, value)
    except ValueError as e:
        if "unknown option" in str(e):
            pytest.skip(f"option {name!r} exists only in the bench build (libpsk_hip_knobs.so)")
        raise


def knob_value(name: str, default: int) -> int:
    """the value of a bench-build option, or `default` where the shipped library does not know the name"""
    from pyprobables_amd import _native as N

    try:
        return N.get_option(name)
    except ValueError:
        return default
