#!/usr/bin/env python3
"""Ablation of the keyed Bloom lookup's pass 2 (k_bloom_test; bench-only debug bits of the knobs build; answers are NOT valid):
64 = no LDS reads (every probe 'hits'), 128 = no slice load, 192 = both -> what is left is the probe stream + the walk."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, timed_loop, use_knobs_build  # noqa: E402

use_knobs_build()
import pyprobables_amd as pa
from pyprobables_amd import _native as N

n = 10_000_000
keys = gen_keys(n, 0, 0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
blm.add_many(keys)
N.set_option("bloom_lookup", 0)
for dbg, label in [(0, "full"), (64, "no LDS reads"), (128, "no slice load"), (192, "stream + walk only"), (1, "pass 1 without stores (pass 2 walks stale segments)")]:
    N.set_option("part_debug", dbg)
    ms = timed_loop(lambda: blm.check_many(keys), 10, warm=3)
    print(f"dbg={dbg:3d} {label:40s} check {ms*1e3:8.1f} us", flush=True)
N.set_option("part_debug", 0)
