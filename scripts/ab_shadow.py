#!/usr/bin/env python3
"""round 3: repeated CountingBloomFilter lookups of an unchanged table with and without the kept 4-bit images (psk_sketch::shadow)"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, timed_loop  # noqa: E402

import pyprobables_amd as pa
from pyprobables_amd import _native as N

n = 10_000_000
keys = gen_keys(n, 0, 0)
for est, label in ((28005615, "2^28 counters (1 GiB)"), (10_000_000, "9.6e7 counters"), (3_600_000, "3.45e7 counters"), (1_800_000, "1.7e7 counters")):
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01)
    cbf.add_many(keys[: min(n, est // 2)])
    line = f"{label:24s} m={cbf.number_bits:10d}"
    for opt in (0, 1):
        N.set_option("cbf_lookup_shadow", opt)
        ms = timed_loop(lambda: cbf.check_many(keys), 10, warm=3)
        line += f" | shadow {opt}: {ms*1e3:7.1f} us ({n/ms/1e6:5.1f} G keys/s)"
    for nn in (1_000_000, 3_000_000):
        for opt in (0, 1):
            N.set_option("cbf_lookup_shadow", opt)
            ms = timed_loop(lambda: cbf.check_many(keys[:nn]), 10, warm=3)
            line += f" | {nn//1000000}M keys shadow {opt}: {ms*1e3:6.1f} us"
    N.set_option("cbf_lookup_shadow", 1)
    print(line, flush=True)
    del cbf
