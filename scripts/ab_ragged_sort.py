#!/usr/bin/env python3
"""Same-box A/B of pass 1's per-tile length sort of ragged keys (option "ragged_sort"): M keys/s with the sort / in batch order, for three
length distributions (wide: 4 + Exp(12.6) capped at 40; narrow: 4 + Poisson(12); words: 2 .. 15 letters)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import bench
import pyprobables_amd as pa
from pyprobables_amd import _native as N

n = 10_000_000


def ragged(kind, seed=7):
    rng = np.random.default_rng(seed)
    if kind == "wide":
        lens = 4 + np.minimum(36, np.floor(rng.exponential(12.6, n))).astype(np.int64)
    elif kind == "narrow":
        lens = 4 + np.minimum(36, rng.poisson(12, n)).astype(np.int64)
    elif kind == "const16":
        lens = np.full(n, 16, dtype=np.int64)
    else:
        lens = np.clip(np.round(rng.normal(8, 2.5, n)), 2, 15).astype(np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    return (torch.from_numpy(rng.integers(0, 256, int(offs[-1]), dtype=np.uint8)).cuda(), torch.from_numpy(offs).cuda())


f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
k16 = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
N.check(N.lib().psk_gen_keys16(k16.data_ptr(), 0, n, bench.SEED, 0, torch.cuda.current_stream().cuda_stream or None))
for _ in range(1500):
    f.add_many(k16)
torch.cuda.synchronize()
a = bench.timed_loop(torch, lambda: f.add_many(k16), 10, warm=2)
c = bench.timed_loop(torch, lambda: f.check_many(k16), 10, warm=2)
print(f"{'fixed 16-byte keys':28s} insert {n / a / 1e3:8.0f}  check {n / c / 1e3:8.0f} M keys/s")
for kind in ("wide", "narrow", "words", "const16"):
    pair = ragged(kind)
    for srt in (1, 0, 1, 0):
        N.set_option("ragged_sort", srt)
        a = bench.timed_loop(torch, lambda: f.add_many(pair), 10, warm=2)
        c = bench.timed_loop(torch, lambda: f.check_many(pair), 10, warm=2)
        print(f"{kind:10s} ragged_sort={srt}        insert {n / a / 1e3:8.0f}  check {n / c / 1e3:8.0f} M keys/s", flush=True)
    del pair
N.set_option("ragged_sort", 1)
