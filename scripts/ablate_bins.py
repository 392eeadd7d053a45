#!/usr/bin/env python3
"""Ablation timing and phase profile of pass 1 at HEAD, both forms (bench-only debug bits of the -DPSK_BENCH_KNOBS=1 build; the ablated runs
do NOT build valid filters): k_part_bins (pass1_bins = 1, what ships for Bloom inserts / tile-flag lookups) and k_part_scatter (pass1_bins = 0).
    python scripts/ablate_bins.py  ->  profiles/r06_ablation_phase_profile.txt (via gpurun)"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, timed_loop, use_knobs_build  # noqa: E402

import os  # noqa: E402

if not os.environ.get("PSK_LIB_PATH"):  # (a bench build handed over explicitly: PSK_LIB_PATH=ab/libpsk_knobs.so)
    use_knobs_build()
import torch  # noqa: E402

import pyprobables_amd as pa  # noqa: E402
from pyprobables_amd import _native as N  # noqa: E402

n, nslices, nwg = 10_000_000, 256, 512
keys = gen_keys(n, 0, 0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
print("m =", blm.number_bits, "k =", blm.number_hashes, "n =", n, "(insert = pass 1 + k_bloom_apply; apply alone ~51 us)")
for bins, kern in ((1, "k_part_bins"), (0, "k_part_scatter")):
    N.set_option("pass1_bins", bins)
    for dbg, label in [(0, "full"), (1, "no segment stores"), (4, "no hashing (stand-in indices)"), (5, "neither"), (2, "hashing only (+ key loads)")]:
        N.set_option("part_debug", dbg)
        ms = timed_loop(lambda: blm.add_many(keys), 20, warm=5)
        print(f"{kern:15s} dbg={dbg} {label:30s} insert {ms * 1e3:8.1f} us", flush=True)
    N.set_option("part_debug", 32)
    blm.add_many(keys)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 12)()
    N.check(N.lib().psk_debug_phase_profile(blm._tab.handle, nslices, nwg, buf))  # clear what the warm-up left
    for _ in range(3):
        blm.add_many(keys)
    torch.cuda.synchronize()
    N.check(N.lib().psk_debug_phase_profile(blm._tab.handle, nslices, nwg, buf))
    if bins:
        for who, base in (("first wave (oldest of its SIMD)", 1), ("last wave", 5)):
            tot = sum(buf[base:base + 4])
            names = ("hash + slots (own work)", "wait at barrier 1", "write-out (own work)", "wait at barrier 2")
            print(f"{kern} phase profile, lane 0 of the {who}: " + "; ".join(f"{nm} {100.0 * buf[base + i] / tot:4.1f} %" for i, nm in enumerate(names))
                  + f"  ({tot / buf[0] / 3:.0f} ticks per workgroup and launch)")
    else:
        names = {1: "zero+bar", 9: "hash+hist (own work)", 2: "wait at barrier 1", 6: "scan: read hist+zero", 7: "scan: wave scan", 8: "scan: cursor+pads", 3: "wait at barrier 2",
                 4: "sort+bar", 5: "writeout(+bar)"}
        tot = sum(buf[1:12])
        print(f"{kern} phase profile, lane 0 of wave 0: " + "; ".join(f"{names[i]} {100.0 * buf[i] / tot:4.1f} %" for i in (1, 9, 2, 6, 7, 8, 3, 4, 5)))
    N.set_option("part_debug", 0)
N.set_option("pass1_bins", 1)
