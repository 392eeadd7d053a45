#!/usr/bin/env python3
"""A/B of the pass-1 workgroup shape (option tile_threads: 1024 = one 1024-thread workgroup per CU, 0 = auto: two 512-thread
workgroups with 32 probes per thread where two LDS stages fit) at 10 M keys: us per call"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop  # noqa: E402

import torch  # noqa: E402

import pyprobables_amd as pa  # noqa: E402
from pyprobables_amd import _native as N  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
keys = gen_keys(n)
w = gen_weights(n)
f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
c = pa.CountMinSketch(width=2**20, depth=5)
for _ in range(1500):
    f.add_many(keys)
torch.cuda.synchronize()
for rep in range(3):
    for even in (1024, 0):
        N.set_option("tile_threads", even)
        t = [timed_loop(lambda: f.add_many(keys), 20), timed_loop(lambda: f.check_many(keys), 20),
             timed_loop(lambda: c.add_many(keys, w), 20), timed_loop(lambda: c.check_many(keys), 20), timed_loop(lambda: c.add_many(keys), 20)]
        print(f"tile_threads {even:4d}: bloom add {t[0]*1e3:7.1f}  check {t[1]*1e3:7.1f}   cms add {t[2]*1e3:7.1f}  check {t[3]*1e3:7.1f}  unit add {t[4]*1e3:7.1f} us")
N.set_option("tile_threads", 0)
