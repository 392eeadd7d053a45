#!/usr/bin/env python3
"""Throughput matrix over table geometry / key layout (side evidence, not the headline): M keys/s per op."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import bench
import pyprobables_amd as pa
from pyprobables_amd import _native as N

n = 10_000_000
_st = lambda: torch.cuda.current_stream().cuda_stream or None  # noqa: E731
keys16 = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
N.check(N.lib().psk_gen_keys16(keys16.data_ptr(), 0, n, bench.SEED, 0, _st()))
w = torch.empty(n, dtype=torch.int32, device="cuda")
N.check(N.lib().psk_gen_weights(w.data_ptr(), 0, n, bench.SEED, 0, _st()))
rows = []
# clock ramp: a second of work before the first measurement
_f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
for _ in range(2000):
    _f.add_many(keys16)
torch.cuda.synchronize()
del _f


def t(fn, it=5):
    return bench.timed_loop(torch, fn, it, warm=2)


def bloom_case(label, est, fpr, keys):
    f = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
    nn = keys.shape[0]
    a = t(lambda: f.add_many(keys))
    c = t(lambda: f.check_many(keys))
    rows.append((label, f"m={f.number_bits} k={f.number_hashes}", nn / a / 1e3, nn / c / 1e3))
    del f


import os

ONLY = os.environ.get("PSK_MATRIX_ONLY", "")  # "varlen": the headline row + the variable-length rows (development)
bloom_case("bloom pow2 2^28 (headline)", 28005615, 0.01, keys16)
if ONLY == "varlen":
    _real_bloom_case = bloom_case
    bloom_case = lambda *a, **k: None  # noqa: E731
bloom_case("bloom non-pow2 ~96 Mbit", 10_000_000, 0.01, keys16)
bloom_case("bloom non-pow2 ~1.9 Gbit", 200_000_000, 0.01, keys16)
bloom_case("bloom pow2 2^31", 224044920, 0.01, keys16)
bloom_case("bloom k=4 (fpr 0.05)", 40_000_000, 0.05, keys16)
bloom_case("bloom k=10 (fpr 0.001)", 10_000_000, 0.001, keys16)
bloom_case("bloom k=17 (fpr 1e-5)", 10_000_000, 1e-5, keys16)
k8 = torch.randint(0, 256, (n, 8), dtype=torch.uint8, device="cuda")
k32 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
k13 = torch.randint(0, 256, (n, 13), dtype=torch.uint8, device="cuda")
bloom_case("bloom 8-byte keys", 28005615, 0.01, k8)
bloom_case("bloom 32-byte keys", 28005615, 0.01, k32)
bloom_case("bloom 13-byte keys (unaligned)", 28005615, 0.01, k13)
if ONLY == "varlen":
    bloom_case = _real_bloom_case
# ---- the reference's native key type: variable-length byte / str keys (hashes.py:98 walks code points) ----
def ragged(kind, nn, seed=7):
    """(blob, offsets) on the device.  wide: 4 + min(36, floor(Exp(12.6))) bytes, mean ~16; narrow: 4 + min(36, Poisson(12));
    words: 2 .. 15 lowercase letters, mean ~8; wide32: the wide lengths as 4-byte code points, one per key above 255"""
    rng = np.random.default_rng(seed)
    if kind in ("wide", "wide32"):
        lens = 4 + np.minimum(36, np.floor(rng.exponential(12.6, nn))).astype(np.int64)
    elif kind == "narrow":
        lens = 4 + np.minimum(36, rng.poisson(12, nn)).astype(np.int64)
    else:
        lens = np.clip(np.round(rng.normal(8, 2.5, nn)), 2, 15).astype(np.int64)
    offs = np.zeros(nn + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    tot = int(offs[-1])
    if kind == "words":
        blob = rng.integers(97, 123, tot, dtype=np.uint8)
    elif kind == "wide32":
        blob = rng.integers(32, 127, tot, dtype=np.uint32)
        blob[offs[:-1]] = rng.integers(0x400, 0x2000, nn, dtype=np.uint32)
    else:
        blob = rng.integers(0, 256, tot, dtype=np.uint8)
    return blob, offs, float(lens.mean()), int(lens.max())


for kind, label in [("wide", "bloom ragged bytes 4-40 (exp, device)"), ("narrow", "bloom ragged bytes 4-40 (poisson, device)"),
                    ("words", "bloom ascii words 2-15 (device)"), ("wide32", "bloom str code points > 255 (device)")]:
    blob, offs, mean, mx = ragged(kind, n)
    dk = (torch.from_numpy(blob.view(np.int32) if blob.dtype == np.uint32 else blob).cuda(), torch.from_numpy(offs).cuda())
    f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    for _ in range(40):  # (the batch was made on the host: the clocks have dropped meanwhile)
        f.add_many(dk)
    a = t(lambda: f.add_many(dk))
    c = t(lambda: f.check_many(dk))
    rows.append((label, f"mean {mean:.1f} max {mx}", n / a / 1e3, n / c / 1e3))
    del f, dk
if ONLY == "varlen":
    print(f"{'case':48s} {'geometry':22s} {'update Mkeys/s':>15s} {'lookup Mkeys/s':>15s}")
    for r in rows:
        print(f"{r[0]:48s} {r[1]:22s} {r[2]:15.0f} {r[3]:15.0f}")
    sys.exit(0)
# host lists of bytes / str (packing + PCIe inclusive), 1 M keys
nh = 1_000_000
blob, offs, mean, mx = ragged("wide", nh)
raw = blob.tobytes()
lst = [raw[offs[i]:offs[i + 1]] for i in range(nh)]
f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
a = t(lambda: f.add_many(lst), 3)
c = t(lambda: f.check_many(lst), 3)
rows.append(("bloom list of ragged bytes (host, 1 M)", f"mean {mean:.1f}", nh / a / 1e3, nh / c / 1e3))
blob, offs, mean, mx = ragged("words", nh)
raw = blob.tobytes().decode("ascii")
lst = [raw[offs[i]:offs[i + 1]] for i in range(nh)]
a = t(lambda: f.add_many(lst), 3)
c = t(lambda: f.check_many(lst), 3)
rows.append(("bloom list of ascii str words (host, 1 M)", f"mean {mean:.1f}", nh / a / 1e3, nh / c / 1e3))
del f, lst
blob, offs, mean, mx = ragged("wide", n)
dk = (torch.from_numpy(blob).cuda(), torch.from_numpy(offs).cuda())
cms = pa.CountMinSketch(width=2**20, depth=5)
a = t(lambda: cms.add_many(dk, w))
c = t(lambda: cms.check_many(dk))
rows.append(("cms 2^20 x 5 ragged bytes 4-40 (device)", f"mean {mean:.1f}", n / a / 1e3, n / c / 1e3))
del cms
cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
a = t(lambda: cbf.add_many(dk), 3)
c = t(lambda: cbf.check_many(dk), 3)
rows.append(("cbf 2^28 ragged bytes 4-40 (device)", f"mean {mean:.1f}", n / a / 1e3, n / c / 1e3))
del cbf, dk
# host-staged (PCIe inclusive) 16-byte keys
kh = keys16.cpu().numpy()
f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
a = t(lambda: f.add_many(kh), 3)
c = t(lambda: f.check_many(kh), 3)
rows.append(("bloom host buffers (PCIe incl.)", "m=2^28 k=7", n / a / 1e3, n / c / 1e3))
del f
for label, width, depth in [("cms 2^20 x 5 (headline)", 2**20, 5), ("cms 1e6+3 x 5 (non-pow2)", 1_000_003, 5), ("cms 2^24 x 8", 2**24, 8)]:
    cms = pa.CountMinSketch(width=width, depth=depth)
    a = t(lambda: cms.add_many(keys16, w))
    u = t(lambda: cms.add_many(keys16))
    c = t(lambda: cms.check_many(keys16))
    rows.append((label + " weighted add / check", f"{width}x{depth}", n / a / 1e3, n / c / 1e3))
    rows.append((label + " unit add", f"{width}x{depth}", n / u / 1e3, float("nan")))
    del cms
for label, est in [("cbf 2^28 counters (1 GiB, cfg 4)", 28005615), ("cbf ~9.6e7 counters", 10_000_000), ("cbf 2^25 counters", 3_500_701)]:
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01)
    a = t(lambda: cbf.add_many(keys16), 3)
    c = t(lambda: cbf.check_many(keys16), 3)
    r = t(lambda: (cbf.add_many(keys16), cbf.remove_many(keys16)), 2) - a  # a remove needs its keys back in: (add + remove) - add
    rows.append((label + " add / check", f"m={cbf.number_bits}", n / a / 1e3, n / c / 1e3))
    rows.append((label + " remove", f"m={cbf.number_bits}", n / r / 1e3, float("nan")))
    del cbf
print(f"{'case':48s} {'geometry':22s} {'update Mkeys/s':>15s} {'lookup Mkeys/s':>15s}")
for r in rows:
    print(f"{r[0]:48s} {r[1]:22s} {r[2]:15.0f} {r[3]:15.0f}")
