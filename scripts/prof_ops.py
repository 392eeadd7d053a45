#!/usr/bin/env python3
"""Run one engine operation repeatedly (for rocprofv3 --kernel-trace --stats / --pmc):  prof_ops.py <op> [n] [iters]
ops: bloom_add bloom_check bloom_check_fresh bloom31_add bloom31_check cms_add cms_add_unit cms_check cbf_add cbf_check cbf_check_kept cbf_remove cbf25_check cbf25_add
     cfg4_stream (n = keys per batch, 50 batches: BASELINE cfg 4's add / remove stream through the DEFAULT API -- update windows, k_win_fold --,
                  flush included; PSK_CFG4_MODE = window (default) | borrow_window (borrow_keys=True) | combine | combine_borrow (the round-2/3 opt-ins))
The Bloom lookups are PINNED to the scheme the timed step of bench.py settles on for that kind of batch (the per-call automatic choice starts on
keyed probes and needs up to four calls to move: a profile of the first calls measures kernels the step does not run):
     bloom_check / bloom31_check (all keys present) -> tile flags (bloom_lookup = 3);  bloom_check_fresh (all absent) -> lazy gathers (4);
     bloom_check_half (every other key absent) -> the automatic choice, warmed over 8 calls before the profiled ones.
PSK_OPTS (e.g. PSK_OPTS=bloom_lookup=0) overrides the pin."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pyprobables_amd as pa  # noqa: E402
from pyprobables_amd import _native as N  # noqa: E402

import os  # noqa: E402

for kv in os.environ.get("PSK_OPTS", "").split(","):  # engine options for this run, e.g. PSK_OPTS=bloom_lookup=3
    if "=" in kv:
        N.set_option(kv.split("=")[0], int(kv.split("=")[1], 0))
op = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
st = lambda: torch.cuda.current_stream().cuda_stream or None  # noqa: E731


def gen(n, start):
    t = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    N.check(N.lib().psk_gen_keys16(t.data_ptr(), start, n, 0x5EED, 0, st()))
    return t


if op == "cfg4_stream":
    B, NB = n, 50
    allk = gen(B * NB, 0)
    mode = os.environ.get("PSK_CFG4_MODE", "window")
    kw = {"window": {}, "borrow_window": {"borrow_keys": True}, "combine": {"combine_updates": True}, "combine_borrow": {"combine_updates": "borrow"}}[mode]
    s = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01, **kw)  # "window" = bench.py --config cfg4 as it runs by default

    def fn():
        s.clear()
        for b in range(NB):
            s.add_many(allk[b * B:(b + 1) * B])
            if b >= 1:
                s.remove_many(allk[(b - 1) * B:(b - 1) * B + B // 2])
        s.synchronize()

    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b_.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b_) / iters
    ops = B * NB + (NB - 1) * (B // 2)
    print(f"{op} ({mode}): {ms * 1e3:.1f} us per {ops} ops -> {ops / ms / 1e3:.0f} M/s  launches={2 + iters} n={ops}")
    sys.exit(0)
keys = gen(n, 0)
w = torch.empty(n, dtype=torch.int32, device="cuda")
N.check(N.lib().psk_gen_weights(w.data_ptr(), 0, n, 0x5EED, 0, st()))
if op.startswith("bloomvar"):  # ragged byte keys (the reference's native key type): PSK_VARLEN_KIND = wide (4 + Exp(12.6), default) | narrow | words
    import numpy as np

    kind = os.environ.get("PSK_VARLEN_KIND", "wide")
    rng = np.random.default_rng(7)
    if kind == "wide":
        lens = 4 + np.minimum(36, np.floor(rng.exponential(12.6, n))).astype(np.int64)
    elif kind == "narrow":
        lens = 4 + np.minimum(36, rng.poisson(12, n)).astype(np.int64)
    else:
        lens = np.clip(np.round(rng.normal(8, 2.5, n)), 2, 15).astype(np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    pair = (torch.from_numpy(rng.integers(0, 256, int(offs[-1]), dtype=np.uint8)).cuda(), torch.from_numpy(offs).cuda())
    s = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    s.add_many(pair)
    fn = {"bloomvar_add": lambda: s.add_many(pair), "bloomvar_check": lambda: s.check_many(pair)}[op]
elif op.startswith("bloom"):
    s = pa.BloomFilter(est_elements=224044920 if op.startswith("bloom31") else 28005615, false_positive_rate=0.01)  # 2^31 / 2^28 bits
    s.add_many(keys)
    fresh = gen(n, 7 * n)
    kind = op.split("_", 1)[1]
    pinned = "bloom_lookup=" in os.environ.get("PSK_OPTS", "")
    if kind == "check" and not pinned:
        s.set_engine_option("bloom_lookup", 3)   # tile flags: what the timed step runs on batches of present keys
    if kind == "check_fresh" and not pinned:
        s.set_engine_option("bloom_lookup", 4)   # lazy gathers: what the automatic choice settles on for batches of absent keys
    if kind == "check_half":
        half = keys.clone()
        half[1::2] = fresh[1::2]
        for _ in range(8):                       # the automatic choice settles (counted: launches = 8 + 2 + iters)
            s.check_many(half)
        torch.cuda.synchronize()
    fn = {"add": lambda: s.add_many(keys), "check": lambda: s.check_many(keys), "check_fresh": lambda: s.check_many(fresh),
          "check_half": lambda: s.check_many(half)}[kind]
elif op.startswith("cms"):
    s = pa.CountMinSketch(width=2**20, depth=5)
    s.add_many(keys, w)
    fn = {"cms_add": lambda: s.add_many(keys, w), "cms_add_unit": lambda: s.add_many(keys), "cms_check": lambda: s.check_many(keys)}[op]
else:
    s = pa.CountingBloomFilter(est_elements=3_500_000, false_positive_rate=0.01) if op.startswith("cbf25") else \
        pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
    s.add_many(keys)
    # check: a lookup of a table that has just changed (the whole 32-bit table is read: the kept 4-bit images are switched off);
    # check_kept: the third and later lookups in a row of an unchanged table (psk_sketch::shadow; the two set-up calls below build the images)
    if op.endswith("_check"):
        N.set_option("cbf_lookup_shadow", 0)
    fn = {"add": lambda: s.add_many(keys), "check": lambda: s.check_many(keys), "check_kept": lambda: s.check_many(keys),
          "remove": lambda: (s.add_many(keys), s.remove_many(keys))}[op.split("_", 1)[1]]  # (remove: the keys go back in first, so that every remove finds its key)
launches = 2 + iters + (8 if op == "bloom_check_half" else 0) + (1 if op in ("bloomvar_add", "bloom_add", "bloom31_add", "cms_add", "cbf_add", "cbf25_add") else 0)  # the set-up insert runs the same kernels
for _ in range(2):
    fn()
    torch.cuda.synchronize()  # (formats chosen from the PREVIOUS call's published tally -- PayWeightSmall, the lookup scheme -- settle before the profiled calls)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    fn()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
print(f"{op}: {ms * 1e3:.1f} us per {n} keys -> {n / ms / 1e3:.0f} M/s  launches={launches} n={n}")
