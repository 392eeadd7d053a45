#!/usr/bin/env python3
"""Same-box A/B of pass-1 builds (round 6): for the library PSK_LIB_PATH names (and the PSK_PASS1_BINS / PSK_BINS_* environment of the call),
   1. parity: Bloom inserts + lookups at three geometries against the plain-C oracle (tables bit for bit, answers incl. false positives),
   2. timing (HIP events, 10 M 16-byte keys, m = 2^28, k = 7): insert, lookup of present keys (tile flags), CMS add, CBF add.
One line per figure; scripts/ab_pass1.sh runs it once per variant on ONE box."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import oracle  # noqa: E402  (the checker)

import bench  # noqa: E402
import pyprobables_amd as pa  # noqa: E402
from pyprobables_amd import _native as N  # noqa: E402

tag = os.environ.get("AB_TAG", Path(os.environ.get("PSK_LIB_PATH", "default")).stem)
ok = True
for est, fpr, n in ((28005615, 0.01, 3_000_017), (3_000_000, 0.01, 1_500_000), (50_000_000, 0.001, 2_000_003)):
    keys = oracle.gen_keys16(0, n)
    probe = oracle.gen_keys16(n // 2, n)  # half present, half fresh
    d, dp = torch.from_numpy(keys).cuda(), torch.from_numpy(probe).cuda()
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr, device=0)
    blm.add_many(d)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    ob.add_keys(keys)
    same = np.array_equal(np.frombuffer(bytes(blm.bloom), dtype=np.uint8), ob.bloom)
    want = ob.check_keys(probe).astype(np.uint8)
    got_all = True
    for scheme in (3, 0, 1):  # tile flags, keyed, return trip
        blm.set_engine_option("bloom_lookup", scheme)
        got = blm.check_many(dp).cpu().numpy().astype(np.uint8)
        hits = blm.check_many(d).cpu().numpy()
        got_all = got_all and np.array_equal(got, want) and bool(hits.all())
    print(f"{tag}: parity m={blm.number_bits} k={blm.number_hashes} n={n}: table {'OK' if same else 'DIFFERS'}, lookups {'OK' if got_all else 'DIFFER'}", flush=True)
    ok = ok and same and got_all
    del blm

n = 10_000_000
keys = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream or None
N.check(N.lib().psk_gen_keys16(keys.data_ptr(), 0, n, 0x5EED, 0, st))
w = torch.empty(n, dtype=torch.int32, device="cuda")
N.check(N.lib().psk_gen_weights(w.data_ptr(), 0, n, 0x5EED, 0, st))
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
blm.add_many(keys)
blm.set_engine_option("bloom_lookup", 3)
res = {}
import time  # noqa: E402

t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:  # clock ramp: a fresh process starts at idle clocks
    blm.add_many(keys)
    blm.check_many(keys)
    torch.cuda.synchronize()
res["insert"] = bench.timed_loop(torch, lambda: blm.add_many(keys), 100, warm=5)
res["check"] = bench.timed_loop(torch, lambda: blm.check_many(keys), 100, warm=5)


def step():
    blm.clear()
    blm.add_many(keys)
    blm.check_many(keys)


res["step"] = bench.timed_loop(torch, step, 100, warm=5)
cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
res["cbf_add"] = bench.timed_loop(torch, lambda: cbf.add_many(keys), 8, warm=2)
del cbf
print(f"{tag}: " + "  ".join(f"{k} {v * 1e3:7.1f} us" for k, v in res.items()) + f"  parity {'OK' if ok else 'FAILED'}", flush=True)
