#!/usr/bin/env python3
"""throughput against the batch size (cfg 2 geometry: Bloom m = 2^28, k = 7; CMS 2^20 x 5): us per call and M keys/s"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop  # noqa: E402

import pyprobables_amd as pa  # noqa: E402

f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
c = pa.CountMinSketch(width=2**20, depth=5)
allk = gen_keys(1 << 24)
allw = gen_weights(1 << 24)
f.add_many(allk)
print(f"{'keys':>10s} {'bloom add':>22s} {'bloom check':>22s} {'cms add':>22s} {'cms check':>22s}")
for lg in range(14, 25, 2):
    n = 1 << lg
    k, w = allk[:n], allw[:n]
    it = 50 if lg <= 20 else 10
    t = [timed_loop(lambda: f.add_many(k), it), timed_loop(lambda: f.check_many(k), it), timed_loop(lambda: c.add_many(k, w), it), timed_loop(lambda: c.check_many(k), it)]
    print(f"{n:10d} " + " ".join(f"{x*1e3:9.1f} us {n/x/1e3:8.0f} M/s" for x in t))
