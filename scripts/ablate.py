#!/usr/bin/env python3
"""Ablation timing of the partitioned Bloom insert (bench-only debug bits; results are NOT valid filters)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop, use_knobs_build  # noqa: E402

import os

if not os.environ.get("PSK_LIB_PATH"):  # (a bench build handed over explicitly: PSK_LIB_PATH=ab/libpsk_knobs.so -- the in-tree one does not travel to the GPU box)
    use_knobs_build()  # the part_debug bits exist only in the -DPSK_BENCH_KNOBS=1 build
import torch

import bench
import pyprobables_amd as pa
from pyprobables_amd import _native as N

# usage: ablate.py [est_elements [n_keys [slices]]]  (defaults: the cfg-2 filter, 10 M keys, 256 slices)
est = int(sys.argv[1]) if len(sys.argv) > 1 else 28005615
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
nslices = int(sys.argv[3]) if len(sys.argv) > 3 else 256
# pass-1 workgroups of the insert (the phase counters sit behind the slices x workgroups segment counts): 512 since the insert
# runs two 512-thread workgroups per CU (256 for tables whose LDS stage does not fit twice, e.g. 2048 slices); lookups: 256
nwg_insert = int(sys.argv[4]) if len(sys.argv) > 4 else (512 if nslices <= 512 else 256)
keys = gen_keys(n, 0, 0)
blm = pa.BloomFilter(est_elements=est, false_positive_rate=0.01, device=0)
print("m =", blm.number_bits, "k =", blm.number_hashes, "n =", n)
for dbg, label in [(0, "full"), (1, "no stores"), (4, "no hashing"), (5, "skeleton only"), (2, "hashing only (+ key loads)"), (10, "hashing only, 1 WG/CU"), (8, "1 WG/CU")]:
    N.set_option("part_debug", dbg)
    ms = timed_loop(lambda: blm.add_many(keys), 10, warm=3)
    print(f"dbg={dbg} {label:28s} insert {ms*1e3:8.1f} us  -> {n/ms/1e3:9.0f} Mkeys/s", flush=True)
N.set_option("part_debug", 0)

# phase profile of pass 1 (s_memtime ticks of lane 0 per workgroup, summed over workgroups)
import ctypes as C
N.set_option("part_debug", 32)
blm.add_many(keys); torch.cuda.synchronize()
buf = (C.c_uint64 * 12)()
N.check(N.lib().psk_debug_phase_profile(blm._tab.handle, nslices, nwg_insert, buf))   # clear whatever warm-up left
for _ in range(3):
    blm.add_many(keys)
torch.cuda.synchronize()
N.check(N.lib().psk_debug_phase_profile(blm._tab.handle, nslices, nwg_insert, buf))
names = ["(wgs)", "zero+bar", "hash+hist+bar", "scan+bar", "sort+bar", "writeout(+bar)"]
tot = sum(buf[1:12])
names += ["  scan: read hist+zero", "  scan: wave scan", "  scan: cursor+pads", "  hash+hist (own work)"]
names[2], names[3] = "wait at barrier 1", "wait at barrier 2"
for i in [1, 9, 2, 6, 7, 8, 3, 4, 5]:
    print(f"phase {names[i]:24s} {buf[i]/buf[0]/3:10.0f} ticks per WG per launch  ({100.0*buf[i]/tot:5.1f} %)")
N.set_option("part_debug", 0)

# same phase profile for the lookup (keyed) variant
N.set_option("part_debug", 32)
blm.check_many(keys); torch.cuda.synchronize()
N.check(N.lib().psk_debug_phase_profile(blm._tab.handle, nslices, 256, buf))
for _ in range(3):
    blm.check_many(keys)
torch.cuda.synchronize()
N.check(N.lib().psk_debug_phase_profile(blm._tab.handle, nslices, 256, buf))
tot = sum(buf[1:12])
for i in [1, 9, 2, 6, 7, 8, 3, 4, 5]:
    print(f"check phase {names[i]:24s} {buf[i]/buf[0]/3:10.0f} ticks per WG per launch  ({100.0*buf[i]/tot:5.1f} %)")
N.set_option("part_debug", 0)
