#!/usr/bin/env python3
"""Ablation + phase profile of pass 1 on ragged byte keys (bench-only debug bits; results are NOT valid filters):
ablate_varlen.py [wide|narrow|words|fixed16]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, timed_loop, use_knobs_build  # noqa: E402

use_knobs_build()
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import pyprobables_amd as pa  # noqa: E402
from pyprobables_amd import _native as N  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "wide"
n = 10_000_000
rng = np.random.default_rng(7)
if kind == "fixed16":
    keys = gen_keys(n, 0, 0)
    N.set_option("tile_threads", 1024)  # the shape the ragged keys run in
else:
    if kind == "wide":
        lens = 4 + np.minimum(36, np.floor(rng.exponential(12.6, n))).astype(np.int64)
    elif kind == "narrow":
        lens = 4 + np.minimum(36, rng.poisson(12, n)).astype(np.int64)
    else:
        lens = np.clip(np.round(rng.normal(8, 2.5, n)), 2, 15).astype(np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    keys = (torch.from_numpy(rng.integers(0, 256, int(offs[-1]), dtype=np.uint8)).cuda(), torch.from_numpy(offs).cuda())
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
for dbg, label in [(0, "full"), (1, "no stores"), (4, "no hashing"), (5, "skeleton only"), (2, "hashing only (+ key loads)")]:
    N.set_option("part_debug", dbg)
    ms = timed_loop(lambda: blm.add_many(keys), 10, warm=3)
    print(f"{kind} dbg={dbg} {label:28s} insert {ms*1e3:8.1f} us  -> {n/ms/1e3:9.0f} Mkeys/s", flush=True)
N.set_option("part_debug", 32)
blm.add_many(keys)
torch.cuda.synchronize()
buf = (C.c_uint64 * 12)()
N.check(N.lib().psk_debug_phase_profile(blm._tab.handle, 256, 256, buf))
for _ in range(3):
    blm.add_many(keys)
torch.cuda.synchronize()
N.check(N.lib().psk_debug_phase_profile(blm._tab.handle, 256, 256, buf))
names = {1: "top of tile", 9: "hash + hist (own work)", 2: "wait at barrier 1", 6: "scan: read hist + zero", 7: "scan: wave scan", 8: "scan: cursors",
         3: "wait at barrier 2", 10: "length sort of the next tile", 4: "stage sort + barrier", 5: "write-out"}
print("raw", list(buf))
tot = max(1, sum(buf[1:12]))
for i in [1, 9, 2, 6, 7, 8, 3, 10, 4, 5]:
    print(f"{kind} phase {names[i]:30s} {buf[i]/max(1, buf[0])/3:10.0f} ticks per WG per launch  ({100.0*buf[i]/tot:5.1f} %)")
N.set_option("part_debug", 0)
