#!/usr/bin/env python3
"""A/B: one 1024-thread workgroup per CU (2 keys per thread) against two 512-thread workgroups per CU with 4 keys per thread
(same 2048-key tiles; build with PSK_FAT512): us per Bloom insert of 10 M keys"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, timed_loop  # noqa: E402

import torch  # noqa: E402

import pyprobables_amd as pa  # noqa: E402
from pyprobables_amd import _native as N  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
keys = gen_keys(n)
f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
for _ in range(1500):
    f.add_many(keys)
torch.cuda.synchronize()
for rep in range(4):
    for tt in (0, 512):
        N.set_option("tile_threads", tt)
        t = timed_loop(lambda: f.add_many(keys), 20)
        print(f"tile_threads {tt:4d}: bloom add {t*1e3:7.1f} us")
N.set_option("tile_threads", 0)
