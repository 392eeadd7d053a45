#!/usr/bin/env python3
"""return-trip lookups at mid-size geometries (A/B of the 4096-key pass-1 tiles' threshold, kLookupFatSlices): us per 10 M keys"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
import pyprobables_amd as pa  # noqa: E402
from pyprobables_amd import _native as N  # noqa: E402

n = 10_000_000
keys = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
N.check(N.lib().psk_gen_keys16(keys.data_ptr(), 0, n, 0x5EED, 0, torch.cuda.current_stream().cuda_stream or None))
out = []
for est in (7_000_000, 14_000_000, 28_005_615):  # 6.7e7 / 1.3e8 / 2.7e8 counters: 256 / 512 / 1024 nibble slices
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01)
    cbf.add_many(keys[: n // 2])
    N.set_option("cbf_lookup_shadow", 0)
    out.append(f"cbf m={cbf.number_bits:.3g}: {bench.timed_loop(torch, lambda: cbf.check_many(keys), 10, warm=3) * 1e3:.0f}")
    N.set_option("cbf_lookup_shadow", 1)
    del cbf
for est in (56_000_000, 112_000_000):  # Bloom 2^29 / 2^30 bits: 512 / 1024 slices, return trip
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=0.01)
    blm.add_many(keys)
    blm.set_engine_option("bloom_lookup", 1)
    out.append(f"bloom m={blm.number_bits:.3g} return trip: {bench.timed_loop(torch, lambda: blm.check_many(keys), 10, warm=3) * 1e3:.0f}")
    del blm
print("  ".join(out))
