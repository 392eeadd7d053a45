#!/usr/bin/env python3
"""Throughput of the stacked filters (ExpandingBloomFilter / RotatingBloomFilter) on the engine."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop  # noqa: E402
import torch

import bench
import pyprobables_amd as pa

n = 10_000_000
keys = gen_keys(n, 0, 0)
dup = keys[torch.randint(0, n // 4, (n,), device=keys.device)]  # every key ~4 times


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for label, est, fpr, kw, data in [
    ("EBF est=1M  fpr=0.01 distinct", 1_000_000, 0.01, {}, keys),
    ("EBF est=1M  fpr=0.01 4x repeats", 1_000_000, 0.01, {}, dup),
    ("EBF est=10M fpr=0.01 distinct", 10_000_000, 0.01, {}, keys),
    ("RBF est=1M  fpr=0.01 queue 4", 1_000_000, 0.01, {"max_queue_size": 4}, keys),
]:
    cls = pa.RotatingBloomFilter if kw else pa.ExpandingBloomFilter
    state = {}

    def run():
        blm = cls(est_elements=est, false_positive_rate=fpr, **kw)
        blm.add_many(data)
        state["blm"] = blm

    t = timed(run)
    blm = state["blm"]
    tc = timed(lambda: blm.check_many(data))
    print(f"{label:34s} add_many {n/t/1e6:8.1f} Mkeys/s ({t*1e3:7.1f} ms, {len(blm._blooms)} filters, {blm.last_batch_stats['chunks']} chunks)"
          f"   check_many {n/tc/1e6:8.1f} Mkeys/s", flush=True)

blm = pa.ExpandingBloomFilter(est_elements=1000, false_positive_rate=0.01)
t0 = time.perf_counter()
for i in range(2000):
    blm.add(f"key-{i}")
dt = (time.perf_counter() - t0) / 2000
t0 = time.perf_counter()
for i in range(2000):
    blm.check(f"key-{i}")
dc = (time.perf_counter() - t0) / 2000
print(f"per-key add {dt*1e6:.1f} us, check {dc*1e6:.1f} us ({len(blm._blooms)} filters)")
