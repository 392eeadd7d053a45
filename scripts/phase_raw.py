#!/usr/bin/env python3
"""Raw phase-profile slots (part_debug 32) and ablation timings of pass 1, insert and lookup (bench-only)."""
import ctypes as C
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

n = 10_000_000
keys = gen_keys(n, 0, 0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
blm.add_many(keys)
for extra, label in ((0, "scatter"),):
    for which, fn in (("insert", lambda: blm.add_many(keys)), ("check", lambda: blm.check_many(keys))):
        for dbg, tag in ((0, "full"), (4, "no hashing"), (1, "no stores"), (2, "hash only")):
            N.set_option("part_debug", dbg | extra)
            ms = timed_loop(fn, 10, warm=3)
            print(f"{label} {which:6s} {tag:10s} {ms*1e3:8.1f} us", flush=True)
        N.set_option("part_debug", 32 | extra)
        fn(); torch.cuda.synchronize()
        buf = (C.c_uint64 * 12)()
        nwg = 256
        N.check(N.lib().psk_debug_phase_profile(blm._tab.handle, 256, nwg, buf))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        N.check(N.lib().psk_debug_phase_profile(blm._tab.handle, 256, nwg, buf))
        tot = sum(buf[1:12]) or 1
        print(f"{label} {which:6s} wgs={buf[0]} " + " ".join(f"[{i}]={100.0*buf[i]/tot:.1f}%" for i in range(1, 12) if buf[i]) + f" total={tot/max(buf[0],1):.0f}")
N.set_option("part_debug", 0)
