#!/usr/bin/env python3
"""A/B of the lookup round structure at the headline size (10 M keys, m = 2^28): M keys/s insert / lookup"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, timed_loop  # noqa: E402

import torch  # noqa: E402

import pyprobables_amd as pa  # noqa: E402
from pyprobables_amd import _native as N  # noqa: E402

n = 10_000_000
keys = gen_keys(n)
s = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
for _ in range(1500):
    s.add_many(keys)
torch.cuda.synchronize()
cache0 = N.get_option("partition_cache_bytes")
for rep in range(2):
    for wgs, tt, cache in ((0, 0, cache0), (512, 0, cache0), (512, 0, 0), (0, 512, cache0), (1024, 512, cache0), (1024, 512, 0), (0, 0, cache0)):
        N.set_option("scatter_workgroups", wgs)
        N.set_option("tile_threads", tt)
        N.set_option("partition_cache_bytes", cache)
        c = timed_loop(lambda: s.check_many(keys), 20)
        a = timed_loop(lambda: s.add_many(keys), 20)
        print(f"   wgs {wgs:4d} tile_threads {tt:4d} cache {cache:10d}: {n / a / 1e3:8.0f} / {n / c / 1e3:8.0f}   ({c * 1e3:.1f} us)")
N.set_option("tile_threads", 0)
N.set_option("scatter_workgroups", 0)
N.set_option("partition_cache_bytes", cache0)
