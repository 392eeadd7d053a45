#!/usr/bin/env python3
"""us per call of Bloom insert / lookup on a big table (default m = 2^31: 2048 slices) at 10 M and 2^25 keys per call, for the
engine build named by PSK_LIB_PATH (scripts/build_variant.sh); also CMS / CBF lookups at 10 M keys (the other pass-2 kernels)"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, timed_loop  # noqa: E402

import torch  # noqa: E402

import pyprobables_amd as pa  # noqa: E402

est = int(sys.argv[1]) if len(sys.argv) > 1 else 224044920
tag = os.path.basename(os.environ.get("PSK_LIB_PATH", "default")) + " " + os.environ.get("PSK_OPTIONS", "")  # (engine options: PSK_OPTIONS=name=value,..., pyprobables_amd/_native.py)
f = pa.BloomFilter(est_elements=est, false_positive_rate=0.01)
out = []
for n in (10_000_000, 1 << 25):
    keys = gen_keys(n)
    ti = timed_loop(lambda: f.add_many(keys), 5)
    tc = timed_loop(lambda: f.check_many(keys), 5)
    ok = bool(f.check_many(keys).all().item())
    out.append(f"n={n}: add {ti*1e3:8.1f} us ({n/ti/1e3:7.0f} M/s)  check {tc*1e3:8.1f} us ({n/tc/1e3:7.0f} M/s) ok={ok}")
    del keys
del f
keys = gen_keys(10_000_000)
c = pa.CountMinSketch(width=2**20, depth=5)
c.add_many(keys)
tc = timed_loop(lambda: c.check_many(keys), 5)
out.append(f"cms check {tc*1e3:8.1f} us")
b = pa.CountingBloomFilter(est_elements=3_000_000, false_positive_rate=0.01)
b.add_many(keys)
tc = timed_loop(lambda: b.check_many(keys), 5)
tr = timed_loop(lambda: (b.add_many(keys), b.remove_many(keys)), 3)
out.append(f"cbf(3M) check {tc*1e3:8.1f} us  add+remove {tr*1e3:8.1f} us")
print(tag, " | ".join(out))
