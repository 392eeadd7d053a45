#!/usr/bin/env python3
"""A/B of the pass-1 workgroup count (option scatter_workgroups) on the large Bloom geometries: M keys/s insert / lookup"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, timed_loop  # noqa: E402

import torch  # noqa: E402

import pyprobables_amd as pa  # noqa: E402
from pyprobables_amd import _native as N  # noqa: E402

n = 1 << 25
keys = gen_keys(n)
f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
for _ in range(400):
    f.add_many(keys)
torch.cuda.synchronize()
cases = [("bloom 2^31", lambda: pa.BloomFilter(est_elements=224044920, false_positive_rate=0.01)),
         ("bloom 2^30", lambda: pa.BloomFilter(est_elements=112022460, false_positive_rate=0.01)),
         ("bloom 2^28", lambda: pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01))]
for label, make in cases:
    s = make()
    print(label, s.number_bits, s.number_hashes)
    for wgs, tt in ((0, 0), (256, 0), (512, 0), (0, 512), (512, 512), (1024, 512)):
        N.set_option("scatter_workgroups", wgs)
        N.set_option("tile_threads", tt)
        a = timed_loop(lambda: s.add_many(keys), 6)
        c = timed_loop(lambda: s.check_many(keys), 6)
        print(f"   wgs {wgs:4d} tile_threads {tt:4d}: {n / a / 1e3:8.0f} / {n / c / 1e3:8.0f}")
    N.set_option("tile_threads", 0)
    del s
N.set_option("scatter_workgroups", 0)
