#!/usr/bin/env python3
"""Ablation timing of the partitioned Bloom insert (bench-only debug bits; results are NOT valid filters)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import bench
import pyprobables_amd as pa
from pyprobables_amd import _native as N

n = 10_000_000
keys = bench.gen_keys(n, 0, 0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
for dbg, label in [(0, "full"), (1, "no stores"), (2, "no reservation atomics"), (3, "no stores, no atomics"), (4, "no hashing"), (7, "skeleton only")]:
    N.set_option("part_debug", dbg)
    ms = bench.timed_loop(lambda: blm.add_many(keys), 10, warm=3)
    print(f"dbg={dbg} {label:28s} insert {ms*1e3:8.1f} us  -> {n/ms/1e3:9.0f} Mkeys/s", flush=True)
N.set_option("part_debug", 0)
