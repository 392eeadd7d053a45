#!/usr/bin/env python3
"""Ablation timing of the partitioned Bloom lookup (bench-only debug bits; results are NOT valid)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop, use_knobs_build  # noqa: E402

use_knobs_build()  # the part_debug bits exist only in the -DPSK_BENCH_KNOBS=1 build
import torch

import bench
import pyprobables_amd as pa
from pyprobables_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
keys = gen_keys(n, 0, 0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
blm.add_many(keys)
for dbg, label in [(0, "full"), (64, "no LDS reads"), (128, "no slice load"), (192, "neither"), (1, "no pass-1 stores"), (0, "full")]:
    N.set_option("part_debug", dbg)
    ms = timed_loop(lambda: blm.check_many(keys), 10, warm=3)
    print(f"dbg={dbg:3d} {label:20s} check {ms*1e3:8.1f} us  -> {n/ms/1e3:9.0f} Mkeys/s", flush=True)
N.set_option("part_debug", 0)
