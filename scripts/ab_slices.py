#!/usr/bin/env python3
"""A/B of the slice-size heuristic (option slice_bias) over a few geometries: M keys/s update / lookup"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop  # noqa: E402

import torch  # noqa: E402

import pyprobables_amd as pa  # noqa: E402
from pyprobables_amd import _native as N  # noqa: E402

n = 10_000_000
keys = gen_keys(n)
w = gen_weights(n)
f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
for _ in range(1500):
    f.add_many(keys)
torch.cuda.synchronize()
cases = [("bloom 2^28", lambda: pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)),
         ("bloom 96Mbit np2", lambda: pa.BloomFilter(est_elements=10_000_000, false_positive_rate=0.01)),
         ("bloom 2^24 k=10", lambda: pa.BloomFilter(est_elements=1160981, false_positive_rate=0.0009653916676755292)),
         ("bloom 2^26 k=4", lambda: pa.BloomFilter(est_elements=10_760_000, false_positive_rate=0.05)),
         ("cms 2^20x5", lambda: pa.CountMinSketch(width=2**20, depth=5)),
         ("cms 2^18x4", lambda: pa.CountMinSketch(width=2**18, depth=4)),
         ("cms 2^22x3", lambda: pa.CountMinSketch(width=2**22, depth=3)),
         ("cbf 2^25", lambda: pa.CountingBloomFilter(est_elements=3_500_701, false_positive_rate=0.01)),
         ("cbf 2^23", lambda: pa.CountingBloomFilter(est_elements=875_000, false_positive_rate=0.01))]
for label, make in cases:
    row = []
    for bias in (0, 1):
        N.set_option("slice_bias", bias)
        s = make()
        if isinstance(s, pa.CountMinSketch):
            a = timed_loop(lambda: s.add_many(keys, w), 8)
        else:
            a = timed_loop(lambda: s.add_many(keys), 8)
        c = timed_loop(lambda: s.check_many(keys), 8)
        row.append((n / a / 1e3, n / c / 1e3))
        del s
    print(f"{label:20s} bias0 {row[0][0]:8.0f} / {row[0][1]:8.0f}   bias+1 {row[1][0]:8.0f} / {row[1][1]:8.0f}")
N.set_option("slice_bias", 0)
