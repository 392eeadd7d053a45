#!/usr/bin/env python3
"""How does the round size (partition_max_keys) move the partitioned Bloom insert / lookup?"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop  # noqa: E402
import bench
import pyprobables_amd as pa
from pyprobables_amd import _native as N

n = 10_000_000
keys = gen_keys(n, 0, 0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
blm.add_many(keys)
for mk in (1 << 25, 5_000_000, 3_400_000, 2_500_000, 1_250_000, 1 << 25):
    N.set_option("partition_max_keys", mk)
    ms_c = timed_loop(lambda: blm.check_many(keys), 10, warm=3)
    ms_a = timed_loop(lambda: blm.add_many(keys), 10, warm=3)
    print(f"max_keys={mk:9d} check {ms_c*1e3:8.1f} us ({n/ms_c/1e3:7.0f} Mkeys/s)   insert {ms_a*1e3:8.1f} us ({n/ms_a/1e3:7.0f} Mkeys/s)", flush=True)
