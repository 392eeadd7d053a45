#!/usr/bin/env python3
"""direct vs partitioned path as a function of batch size (sets partition_min_keys)"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop  # noqa: E402
import torch

import bench
import pyprobables_amd as pa
from pyprobables_amd import _native as N

keys = gen_keys(4_000_000, 0, 0)
w = gen_weights(4_000_000, 0, 0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
cms = pa.CountMinSketch(width=2**20, depth=5)
print(f"{'n':>9s} | {'bloom add us: direct':>21s} {'part':>8s} | {'bloom check: direct':>20s} {'part':>8s} | {'cms add: direct':>16s} {'part':>8s}")
for n in (16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 4000000):
    k = keys[:n]
    row = []
    for mode in (0, 1):
        N.set_option("partition", mode)
        N.set_option("partition_min_keys", 1)
        row.append(timed_loop(lambda: blm.add_many(k), 20, warm=3) * 1e3)
        row.append(timed_loop(lambda: blm.check_many(k), 20, warm=3) * 1e3)
        row.append(timed_loop(lambda: cms.add_many(k, w[:n]), 20, warm=3) * 1e3)
    print(f"{n:9d} | {row[0]:21.1f} {row[3]:8.1f} | {row[1]:20.1f} {row[4]:8.1f} | {row[2]:16.1f} {row[5]:8.1f}")
