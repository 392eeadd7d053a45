#!/usr/bin/env python3
"""Generate tests/golden/golden_cdemo.json by running the REAL reference (pyprobables v0.7.0) over the workloads of the plain-C
programs under examples/ (psk_demo.c, psk_threads_demo.c), so that tests/test_gpu_c_abi.py can pin the C side of the boundary to
digests the reference produced, not only to properties.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_cdemo.py [/root/reference]      (build container only; ~2 minutes)

Data only: the recipe of each key stream and the sha256 of what the reference held afterwards.
"""

import hashlib
import json
import struct
import sys
from pathlib import Path

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from probables import BloomFilter, CountingBloomFilter  # noqa: E402

M64 = 2**64 - 1


def sm(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def keys_of(seed, n):
    """the C programs' gen_keys(): 64-bit word i of the buffer = splitmix64(seed + i); key j = words 2j, 2j + 1"""
    return [struct.pack("<QQ", sm(seed + 2 * j), sm(seed + 2 * j + 1)) for j in range(n)]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


out = {"_generator": "tests/golden/gen_golden_cdemo.py (pyprobables reference, imported)"}

# examples/psk_demo.c: BloomFilter(100000, 0.01), insert keys 0..49999 of the SEED = 0x5EED stream, look all 100000 up
blm = BloomFilter(est_elements=100000, false_positive_rate=0.01)
ks = keys_of(0x5EED, 100000)
for k in ks[:50000]:
    blm.add(k)
hits = bytes(1 if blm.check(k) else 0 for k in ks)
out["psk_demo"] = {"est_elements": 100000, "fpr": 0.01, "m": blm.number_bits, "k": blm.number_hashes, "seed": 0x5EED, "inserted": 50000, "looked_up": 100000,
                   "sha256_table": sha(blm.bloom), "positives": sum(hits), "false_positives": sum(hits[50000:]), "sha256_membership_bytes": sha(hits),
                   "elements_added": blm.elements_added}

# examples/psk_threads_demo.c, thread A: BloomFilter(28005615, 0.01) (m = 2^28, k = 7), 5 rounds of 300000 keys, seed 0x5EED + r * 10 * 300000
ROUNDS, NKEYS = 5, 300000
blm = BloomFilter(est_elements=28005615, false_positive_rate=0.01)
for r in range(ROUNDS):
    for k in keys_of(0x5EED + r * 10 * NKEYS, NKEYS):
        blm.add(k)
out["threads_bloom"] = {"est_elements": 28005615, "fpr": 0.01, "m": blm.number_bits, "k": blm.number_hashes, "rounds": ROUNDS, "keys_per_round": NKEYS,
                        "seed": 0x5EED, "sha256_table": sha(blm.bloom), "elements_added": blm.elements_added}

# thread B: CountingBloomFilter(4600000, 0.03) (m = 33572829, k = 5), per round add 300000 keys then remove the first half, seed 0xC0FFEE + ...
cbf = CountingBloomFilter(est_elements=4600000, false_positive_rate=0.03)
mins_last = None
for r in range(ROUNDS):
    ks = keys_of(0xC0FFEE + r * 10 * NKEYS, NKEYS)
    for k in ks:
        cbf.add(k)
    for k in ks[: NKEYS // 2]:
        cbf.remove(k)
    if r == ROUNDS - 1:
        mins_last = struct.pack(f"<{NKEYS}I", *[cbf.check(k) for k in ks])
out["threads_cbf"] = {"est_elements": 4600000, "fpr": 0.03, "m": cbf.number_bits, "k": cbf.number_hashes, "rounds": ROUNDS, "keys_per_round": NKEYS,
                      "seed": 0xC0FFEE, "sha256_table": sha(cbf.bloom), "elements_added": cbf.elements_added, "sha256_last_round_mins_u32": sha(mins_last)}

Path(__file__).with_name("golden_cdemo.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
