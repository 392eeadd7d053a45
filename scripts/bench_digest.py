#!/usr/bin/env python3
"""Bloom insert / lookup throughput with the digest hash families (md5 / sha256 chains on the GPU)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop  # noqa: E402
import bench
import pyprobables_amd as pa

n = 10_000_000
keys = gen_keys(n, 0, 0)
for name, fn in (("fnv_1a (fused)", None), ("md5", pa.default_md5), ("sha256", pa.default_sha256)):
    blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, hash_function=fn, device=0)
    ms_a = timed_loop(lambda: blm.add_many(keys), 5, warm=2)
    ms_c = timed_loop(lambda: blm.check_many(keys), 5, warm=2)
    print(f"{name:16s} insert {n/ms_a/1e3:9.0f} Mkeys/s   lookup {n/ms_c/1e3:9.0f} Mkeys/s", flush=True)
