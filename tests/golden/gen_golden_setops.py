#!/usr/bin/env python3
"""Generate tests/golden/golden_setops.json by running the REAL reference (pyprobables v0.7.0).

Set algebra of SURVEY.md 8(f) N2: BloomFilter.union / intersection / jaccard_index (bloom.py:371-460) and the
CountingBloomFilter variants (countingbloom.py:210-304), including `elements_added = estimate_elements()` on the
result, the CBF "sum only where both are non-zero" rule (countingbloom.py:236-239), the empty-vs-empty 1.0 and a CBF
case where only one side is non-zero.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_setops.py [/root/reference]

Data only: the recipe of the synthetic key stream (same generator as gen_golden.py) and the outputs
the reference produced.
"""

import json
import struct
import sys
from pathlib import Path

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import probables  # noqa: E402
from probables import BloomFilter, CountingBloomFilter  # noqa: E402

M64 = 2**64 - 1
SEED = 0x5EED


def sm(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def key(i):
    return struct.pack("<QQ", sm(SEED + 2 * i), sm(SEED + 2 * i + 1))


def w(i):
    return 1 + sm((SEED ^ 0xC0FFEE) + i) % 7


G = {"reference_version": probables.__version__, "seed": SEED}


def bloom_case(est, fpr, a_range, b_range):
    a = BloomFilter(est_elements=est, false_positive_rate=fpr)
    b = BloomFilter(est_elements=est, false_positive_rate=fpr)
    for i in range(*a_range):
        a.add(key(i))
    for i in range(*b_range):
        b.add(key(i))
    u, x = a.union(b), a.intersection(b)
    try:
        union_bytes = bytes(u).hex()
    except struct.error:  # elements_added == -1 does not fit the footer's "Q" (bloom.py:299-303)
        union_bytes = "struct.error"
    return {
        "est_elements": est, "fpr": fpr, "m": a.number_bits, "k": a.number_hashes,
        "a_keys": list(a_range), "b_keys": list(b_range),
        "a_hex": bytes(a.bloom).hex(), "b_hex": bytes(b.bloom).hex(),
        "union_hex": bytes(u.bloom).hex(), "union_elements_added": u.elements_added,
        "union_estimate_elements": u.estimate_elements(), "union_bytes_hex": union_bytes,
        "intersection_hex": bytes(x.bloom).hex(), "intersection_elements_added": x.elements_added,
        "intersection_estimate_elements": x.estimate_elements(),
        "jaccard": a.jaccard_index(b), "jaccard_self": a.jaccard_index(a),
        "jaccard_ba": b.jaccard_index(a),
    }


G["bloom"] = [
    bloom_case(1000, 0.01, (0, 500), (250, 750)),     # overlapping halves; m = 9586 (not a multiple of 8 * 4)
    bloom_case(100, 0.05, (0, 60), (1000, 1060)),     # disjoint key sets
    bloom_case(10, 0.05, (0, 0), (0, 0)),             # empty vs empty: jaccard 1.0 (bloom.py:458-459)
    bloom_case(10, 0.05, (0, 8), (0, 0)),             # one side empty
    bloom_case(5, 0.3, (0, 400), (400, 800)),         # saturated: estimate_elements() == -1 (bloom.py:348-349)
]


def cbf_case(est, fpr, a_ops, b_ops):
    """ops: list of (start, stop, weighted?)"""
    a = CountingBloomFilter(est_elements=est, false_positive_rate=fpr)
    b = CountingBloomFilter(est_elements=est, false_positive_rate=fpr)
    for flt, ops in ((a, a_ops), (b, b_ops)):
        for lo, hi, weighted in ops:
            for i in range(lo, hi):
                flt.add(key(i), w(i) if weighted else 1)
    u, x = a.union(b), a.intersection(b)
    return {
        "est_elements": est, "fpr": fpr, "m": a.number_bits, "k": a.number_hashes,
        "a_ops": a_ops, "b_ops": b_ops,
        "a_table": list(a.bloom), "b_table": list(b.bloom),
        "a_elements_added": a.elements_added, "b_elements_added": b.elements_added,
        "union_table": list(u.bloom), "union_elements_added": u.elements_added,
        "union_estimate_elements": u.estimate_elements(),
        "intersection_table": list(x.bloom), "intersection_elements_added": x.elements_added,
        "intersection_estimate_elements": x.estimate_elements(),
        "jaccard": a.jaccard_index(b), "jaccard_self": a.jaccard_index(a), "jaccard_ba": b.jaccard_index(a),
        "a_bits_set": a._cnt_number_bits_set(), "b_bits_set": b._cnt_number_bits_set(),
    }


G["cbf"] = [
    cbf_case(100, 0.01, [(0, 60, True)], [(30, 90, True)]),          # overlap, weighted counts
    cbf_case(100, 0.01, [(0, 50, False)], []),                       # only one side non-zero: intersection empty
    cbf_case(10, 0.05, [], []),                                      # empty vs empty: jaccard 1.0 (countingbloom.py:267-268)
    cbf_case(50, 0.05, [(0, 40, False), (0, 20, True)], [(100, 140, True)]),  # disjoint keys, some shared cells
]

out = Path(__file__).resolve().parent / "golden_setops.json"
out.write_text(json.dumps(G, indent=0, ensure_ascii=True))
print("wrote", out, out.stat().st_size, "bytes")
