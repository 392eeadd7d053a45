#!/usr/bin/env python3
"""Generate tests/golden/golden.json by running the REAL reference (pyprobables v0.7.0).

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [/root/reference]

The output is data only: inputs (or the recipe of the synthetic key stream) and the
outputs the reference produced for them.  Synthetic keys follow SURVEY.md section 8(d):
    key(i) = LE64(sm(SEED+2i)) || LE64(sm(SEED+2i+1)),  w(i) = 1 + sm((SEED^0xC0FFEE)+i) % 7
with sm = splitmix64 and SEED = 0x5EED.
"""

import hashlib
import json
import struct
import sys
from pathlib import Path

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import probables  # noqa: E402
from probables import BloomFilter, CountingBloomFilter, CountMinSketch  # noqa: E402
from probables.exceptions import InitializationError  # noqa: E402
from probables.hashes import default_fnv_1a, fnv_1a  # noqa: E402

M64 = 2**64 - 1
SEED = 0x5EED


def sm(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def key(i):
    return struct.pack("<QQ", sm(SEED + 2 * i), sm(SEED + 2 * i + 1))


def w(i):
    return 1 + sm((SEED ^ 0xC0FFEE) + i) % 7


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def packbits(bools):
    """LSB-first bit packing -> hex"""
    out = bytearray((len(bools) + 7) // 8)
    for i, b in enumerate(bools):
        if b:
            out[i >> 3] |= 1 << (i & 7)
    return out.hex()


G = {"reference_version": probables.__version__, "seed": SEED}

# ------------------------------------------------------------------ generator anchors
G["keygen"] = {
    "key0": key(0).hex(),
    "key1": key(1).hex(),
    "key9999": key(9999).hex(),
    "w0_7": [w(i) for i in range(8)],
    "sha256_keys_0_999": sha(b"".join(key(i) for i in range(1000))),
}

# ------------------------------------------------------------------------- hashing
hash_cases = []
str_keys = [
    "this is a test",
    "this is also a test",
    "gMPflVXtwGDXbIhP73TX",
    "LtHf1prlU1bCeYZEdqWf",
    "",
    "a",
    "test",
    "é",  # code point 233 (single element, unlike utf-8)
    "€",  # code point 8364 (> 255, XORed whole)
    "naïve café €5 \U0001f600",
    "x" * 257,
]
for s in str_keys:
    hash_cases.append({"type": "str", "key": s, "depth": 7, "hashes": default_fnv_1a(s, 7)})
byte_keys = [
    b"this is a test",
    b"",
    bytes(range(16)),
    bytes(range(256)),
    key(0),
    key(12345),
    "é".encode("utf-8"),
    "é".encode("latin-1"),
    b"\x00" * 5,
    b"\xff" * 33,
]
for b in byte_keys:
    hash_cases.append({"type": "bytes", "key": b.hex(), "depth": 7, "hashes": default_fnv_1a(b, 7)})
hash_cases.append({"type": "bytes", "key": key(7).hex(), "depth": 40, "hashes": default_fnv_1a(key(7), 40)})
G["hashes"] = hash_cases
G["fnv_1a_seeded"] = [{"key": "seed test", "seed": s, "hash": fnv_1a("seed test", s)} for s in (0, 1, 2, 31, 1000, 2**40)]

# -------------------------------------------------------------------------- sizing
sizing = []
for n, p in [
    (10, 0.05), (1000, 0.05), (100000, 0.01), (16000000, 0.001), (28005615, 0.01), (224044920, 0.01),
    (1, 0.9), (5, 0.5), (20000, 0.01), (5000, 0.01), (1000, 0.001), (123457, 0.0371), (10, 0.0),
    (2, 0.75), (50, 0.99),
]:
    try:
        fpr, k, m = BloomFilter._get_optimized_params(n, p)
        sizing.append({"n": n, "p": p, "fpr": fpr, "k": k, "m": m})
    except InitializationError as ex:
        sizing.append({"n": n, "p": p, "error": ex.message})
    except (ValueError, ZeroDivisionError, OverflowError) as ex:
        sizing.append({"n": n, "p": p, "raises": type(ex).__name__})
for n, p in [(0, 0.05), (-1, 0.05), (10, 1.0), (10, -0.1), (10, 1.5), ("a", 0.1), (10, "b"), (1, 0.999999)]:
    try:
        fpr, k, m = BloomFilter._get_optimized_params(n, p)
        sizing.append({"n": n, "p": p, "fpr": fpr, "k": k, "m": m})
    except InitializationError as ex:
        sizing.append({"n": n, "p": p, "error": ex.message})
G["sizing"] = sizing
cms_sizing = []
for conf, err in [(0.96875, 0.002), (0.99, 0.001), (0.5, 0.5), (0.999, 0.0001)]:
    c = CountMinSketch(confidence=conf, error_rate=err)
    cms_sizing.append({"confidence": conf, "error_rate": err, "width": c.width, "depth": c.depth})
G["cms_sizing"] = cms_sizing

# ---------------------------------------------------------------- Bloom, small (str keys)
blm = BloomFilter(est_elements=10, false_positive_rate=0.05)
for i in range(10):
    blm.add(f"this is a test {i}")
G["bloom_small"] = {
    "est_elements": 10, "fpr": 0.05, "keys": [f"this is a test {i}" for i in range(10)],
    "export_hex": blm.export_hex(),
    "bytes_hex": bytes(blm).hex(),
    "str": str(blm),
    "check_keys": [f"this is a test {i}" for i in range(20)],
    "check": [bool(blm.check(f"this is a test {i}")) for i in range(20)],
    "estimate_elements": blm.estimate_elements(),
    "current_fpr": blm.current_false_positive_rate(),
}
b1 = BloomFilter(est_elements=10, false_positive_rate=0.05)
b1.add("this is a test")
G["bloom_one"] = {"md5_bytes": hashlib.md5(bytes(b1)).hexdigest(), "bytes_hex": bytes(b1).hex()}

# ------------------------------------------------------------------- Bloom cfg 1 (BASELINE configs[0])
blm = BloomFilter(est_elements=1000, false_positive_rate=0.05)
for i in range(10000):
    blm.add(key(i))
G["bloom_cfg1"] = {
    "est_elements": 1000, "fpr": 0.05, "k": blm.number_hashes, "m": blm.number_bits, "n_keys": 10000,
    "table_hex": bytes(blm.bloom).hex(),
    "sha256_table": sha(bytes(blm.bloom)),
    "sha256_bytes": sha(bytes(blm)),
    "bits_set": blm._cnt_number_bits_set(),
    "elements_added": blm.elements_added,
    "all_checks_true": all(blm.check(key(i)) for i in range(10000)),
    "check_fresh_10000_10999": packbits([blm.check(key(i)) for i in range(10000, 11000)]),
}

# ------------------------------------------------- Bloom, non-power-of-two m (exact 64-bit modulo)
blm = BloomFilter(est_elements=100000, false_positive_rate=0.01)
for i in range(50000):
    blm.add(key(i))
res = [bool(blm.check(key(i))) for i in range(40000, 60000)]
G["bloom_np2"] = {
    "est_elements": 100000, "fpr": 0.01, "k": blm.number_hashes, "m": blm.number_bits, "n_keys": 50000,
    "sha256_table": sha(bytes(blm.bloom)),
    "bits_set": blm._cnt_number_bits_set(),
    "check_range": [40000, 60000],
    "positives": sum(res),
    "membership_bits": packbits(res),
    "sha256_membership_bytes": sha(bytes(int(x) for x in res)),
    "estimate_elements": blm.estimate_elements(),
}

# ------------------------------------------------------------ Bloom, m slightly above 2^32 bits? too big
# for pure python; instead a m > 2^32 MODULO check is pinned through add_alt/check_alt on a tiny filter:
blm = BloomFilter(est_elements=10, false_positive_rate=0.05)
big = [2**64 - 1, 2**63, 2**32, 2**32 + 62, 63, 0, 126, 12345678901234567890]
blm.add_alt(big)
G["bloom_alt"] = {"hashes": big, "table_hex": bytes(blm.bloom).hex(), "check_same": bool(blm.check_alt(big)),
                  "check_other": bool(blm.check_alt([1, 2, 3, 4]))}

# -------------------------------------------------- Bloom, variable-length keys (str, bytes, non-ASCII)
var_keys = []
for i in range(3000):
    r = sm(0xABCDEF + i)
    ln = r % 41  # 0..40, includes empty keys
    var_keys.append(bytes((sm(r + j) & 0xFF) for j in range(ln)))
blm = BloomFilter(est_elements=5000, false_positive_rate=0.01)
for kx in var_keys[:2000]:
    blm.add(kx)
G["bloom_varlen"] = {
    "est_elements": 5000, "fpr": 0.01, "keys_hex": [kx.hex() for kx in var_keys],
    "n_added": 2000,
    "table_hex": bytes(blm.bloom).hex(),
    "membership_bits": packbits([blm.check(kx) for kx in var_keys]),
}
uni_keys = ["café %d" % i for i in range(50)] + ["€%d 中文" % i for i in range(50)] + ["plain %d" % i for i in range(50)]
blm = BloomFilter(est_elements=200, false_positive_rate=0.01)
for s in uni_keys[::2]:
    blm.add(s)
G["bloom_unicode"] = {
    "est_elements": 200, "fpr": 0.01, "keys": uni_keys,
    "table_hex": bytes(blm.bloom).hex(),
    "membership_bits": packbits([blm.check(s) for s in uni_keys]),
}

# ----------------------------------------------------------------------------- CMS
cms = CountMinSketch(width=1000, depth=5)
r_add = cms.add("this is a test", 100)
G["cms_one"] = {"md5_bytes": hashlib.md5(bytes(cms)).hexdigest(), "add_return": r_add, "elements_added": cms.elements_added}

cms = CountMinSketch(width=4096, depth=5)
for i in range(50000):
    cms.add(key(i % 5000), w(i))
G["cms_stream"] = {
    "width": 4096, "depth": 5, "n_updates": 50000, "n_distinct": 5000,
    "sha256_bins": sha(bytes(cms._bins)),
    "elements_added": cms.elements_added,
    "check_0_199": [cms.check(key(i)) for i in range(200)],
    "check_fresh_5000_5099": [cms.check(key(i)) for i in range(5000, 5100)],
}
cms.query_type = "mean"
G["cms_stream"]["mean_0_199"] = [cms.check(key(i)) for i in range(200)]
cms.query_type = "mean-min"
G["cms_stream"]["meanmin_0_199"] = [cms.check(key(i)) for i in range(200)]
cms.query_type = "min"
for i in range(20000):
    cms.remove(key(i % 5000), w(i))
G["cms_stream"]["after_remove_20000"] = {
    "sha256_bins": sha(bytes(cms._bins)), "elements_added": cms.elements_added,
    "check_0_49": [cms.check(key(i)) for i in range(50)],
}

# small non-power-of-two width, with per-op returns (sequential semantics incl. all three queries)
ops = []
for qt in ("min", "mean", "mean-min"):
    cms = CountMinSketch(width=37, depth=4)
    cms.query_type = qt
    rets = []
    for i in range(300):
        kx = key(i % 23)
        if i % 5 == 4:
            rets.append(cms.remove(kx, w(i)))
        else:
            rets.append(cms.add(kx, w(i)))
    ops.append({"query": qt, "width": 37, "depth": 4, "returns": rets, "bins": list(cms._bins),
                "elements_added": cms.elements_added, "checks": [cms.check(key(i)) for i in range(30)]})
G["cms_ordered"] = ops

# saturation at the int32 rails (countminsketch_test.py:262-278 style)
cms = CountMinSketch(width=8, depth=3)
sat = {"width": 8, "depth": 3, "steps": []}
h = [1, 10, 19]
for op, n in [("add", 2**31 - 5), ("add", 3), ("add", 10), ("remove", 2**31 - 1), ("remove", 2**31 - 1), ("remove", 100)]:
    r = cms.add_alt(h, n) if op == "add" else cms.remove_alt(h, n)
    sat["steps"].append({"op": op, "n": n, "ret": r, "bins": list(cms._bins), "elements_added": cms.elements_added})
G["cms_saturation"] = sat

# join
c1 = CountMinSketch(width=64, depth=3)
c2 = CountMinSketch(width=64, depth=3)
for i in range(200):
    c1.add(key(i), w(i))
    c2.add(key(i + 100), w(i + 7))
c1.join(c2)
G["cms_join"] = {"width": 64, "depth": 3, "bins": list(c1._bins), "elements_added": c1.elements_added}

# ------------------------------------------------------------------------------ CBF
cbf = CountingBloomFilter(est_elements=10, false_positive_rate=0.05)
for i in range(10):
    cbf.add(f"this is a test {i}")
G["cbf_small"] = {"export_hex": cbf.export_hex(), "str": str(cbf), "bytes_hex": bytes(cbf).hex()}

cbf = CountingBloomFilter(est_elements=20000, false_positive_rate=0.01)
B = 5000
for bt in range(4):
    for i in range(bt * B, (bt + 1) * B):
        cbf.add(key(i))
    if bt >= 1:
        for i in range((bt - 1) * B, (bt - 1) * B + B // 2):
            cbf.remove(key(i))
G["cbf_stream"] = {
    "est_elements": 20000, "fpr": 0.01, "k": cbf.number_hashes, "m": cbf.number_bits, "B": B,
    "sha256_table": sha(bytes(cbf.bloom)),
    "elements_added": cbf.elements_added,
    "sum": sum(cbf.bloom), "max": max(cbf.bloom),
    "check_0_99": [cbf.check(key(i)) for i in range(100)],
    "check_19950_20049": [cbf.check(key(i)) for i in range(19950, 20050)],
}

# weighted well-formed stream
cbf = CountingBloomFilter(est_elements=2000, false_positive_rate=0.01)
for i in range(3000):
    cbf.add(key(i % 1000), w(i))
for i in range(1000):
    cbf.remove(key(i), w(i))
G["cbf_weighted"] = {
    "est_elements": 2000, "fpr": 0.01, "m": cbf.number_bits, "k": cbf.number_hashes,
    "sha256_table": sha(bytes(cbf.bloom)), "elements_added": cbf.elements_added,
    "check_0_99": [cbf.check(key(i)) for i in range(100)],
}

# ordered (ill-formed) stream on a tiny table: removes of absent keys, partial removes, duplicates
# (keys whose k indices collide inside one key are excluded: the reference's remove_alt decrements a
#  duplicated index twice and raises OverflowError when that goes below zero -- countingbloom.py:204-206)
cbf = CountingBloomFilter(est_elements=10, false_positive_rate=0.05)
pool = [i for i in range(200) if len({h % cbf.number_bits for h in cbf.hashes(key(i))}) == cbf.number_hashes][:31]
rets, opl = [], []
for i in range(400):
    kx = pool[i * 7 % 31]
    r = sm(i + 99)
    if r % 3 == 0:
        n = 1 + (r >> 8) % 4
        rets.append(cbf.remove(key(kx), n))
        opl.append([kx, -n])
    else:
        n = 1 + (r >> 8) % 3
        rets.append(cbf.add(key(kx), n))
        opl.append([kx, n])
G["cbf_ordered"] = {"est_elements": 10, "fpr": 0.05, "ops": opl, "returns": rets, "table": list(cbf.bloom),
                    "elements_added": cbf.elements_added}

# saturation + frozen-on-remove (countingbloom_test.py:435-459 style) and duplicate indices
cbf = CountingBloomFilter(est_elements=10, false_positive_rate=0.05)
steps = []
h = [5, 5, 6, 7]
for op, n in [("add", 3), ("remove", 1), ("add", 2**32 - 4), ("add", 5), ("remove", 2), ("remove", 2**32 - 1)]:
    r = cbf.add_alt(h, n) if op == "add" else cbf.remove_alt(h, n)
    steps.append({"op": op, "n": n, "ret": r, "c5": cbf.bloom[5], "c6": cbf.bloom[6], "c7": cbf.bloom[7],
                  "elements_added": cbf.elements_added})
G["cbf_saturation"] = {"hashes": h, "steps": steps}

out = Path(__file__).resolve().parent / "golden.json"
out.write_text(json.dumps(G, indent=0, ensure_ascii=True))
print("wrote", out, out.stat().st_size, "bytes")
