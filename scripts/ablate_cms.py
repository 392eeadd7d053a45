#!/usr/bin/env python3
"""Ablation timing of pass 1 of the partitioned CountMinSketch add / lookup (cfg 3: width 2^20, depth 5; bench-only debug bits,
the results are NOT valid sketches) and of the Bloom keyed lookup of cfg 2 beside it."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop, use_knobs_build  # noqa: E402

use_knobs_build()
import ctypes as C

import torch

import pyprobables_amd as pa
from pyprobables_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
keys = gen_keys(n, 0, 0)
w = gen_weights(n, 0, 0)
cms = pa.CountMinSketch(width=2**20, depth=5, device=0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
ops = [("cms add (weighted)", lambda: cms.add_many(keys, w), 160, 256),
       ("cms add (unit)", lambda: cms.add_many(keys), 160, 256),
       ("cms check", lambda: cms.check_many(keys), 160, 256),
       ("bloom check (keyed)", lambda: blm.check_many(keys), 256, 256)]
labels = [(0, "full"), (1, "no stores"), (4, "no hashing"), (5, "skeleton only"), (2, "hashing only (+ key loads)")]
names = ["(wgs)", "zero+bar", "wait at barrier 1", "wait at barrier 2", "sort+bar", "writeout(+bar)", "  scan: read hist+zero", "  scan: wave scan",
         "  scan: cursor+pads", "  hash+hist (own work)"]
blm.add_many(keys)
for label, fn, nslices, nwg in ops:
    for dbg, what in labels:
        N.set_option("part_debug", dbg)
        ms = timed_loop(fn, 8, warm=2)
        print(f"{label:22s} dbg={dbg} {what:28s} {ms*1e3:8.1f} us", flush=True)
    N.set_option("part_debug", 0)
for tt in (0, 512, 1024):
    N.set_option("tile_threads", tt)
    for label, fn, nslices, nwg in ops:
        ms = timed_loop(fn, 8, warm=2)
        print(f"tile_threads {tt:4d}: {label:22s} {ms*1e3:8.1f} us", flush=True)
N.set_option("tile_threads", 0)
for label, fn, nslices, nwg in ops:
    for wgs in (nwg, 512):
        N.set_option("part_debug", 32)
        buf = (C.c_uint64 * 12)()
        h = (cms if label.startswith("cms") else blm)._tab.handle
        fn(); torch.cuda.synchronize()
        try:
            N.check(N.lib().psk_debug_phase_profile(h, nslices, wgs, buf))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            N.check(N.lib().psk_debug_phase_profile(h, nslices, wgs, buf))
        except Exception as e:
            print(label, wgs, "phase profile failed:", e)
            continue
        tot = sum(buf[1:12])
        if buf[0] == 0 or tot == 0:
            continue
        print(f"-- {label}: phase profile read behind {nslices} x {wgs} segment counts (wgs seen {buf[0] / 3:.0f})")
        for i in [1, 9, 2, 6, 7, 8, 3, 4, 5]:
            print(f"   {names[i]:24s} {buf[i]/buf[0]:10.0f} ticks per WG per launch  ({100.0*buf[i]/tot:5.1f} %)")
    N.set_option("part_debug", 0)
