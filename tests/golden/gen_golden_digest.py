#!/usr/bin/env python3
"""Generate tests/golden/golden_digest.json with the REAL reference: sketches driven by the digest hash families
(probables/hashes.py:125-150 default_md5 / default_sha256).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_digest.py [/root/reference]
"""

import json
import sys
from pathlib import Path

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import probables  # noqa: E402
from probables import BloomFilter, CountingBloomFilter, CountMinSketch  # noqa: E402
from probables.hashes import default_md5, default_sha256  # noqa: E402

G = {"reference_version": probables.__version__}
keys = [f"key-{i}" for i in range(300)] + ["", "é", "日本語のキー", "x" * 55, "y" * 56, "z" * 64, "w" * 130]
G["keys"] = keys
G["md5_depth5"] = {k: default_md5(k, 5) for k in ["this is a test", "this is also a test", "", "é"]}
G["sha256_depth5"] = {k: default_sha256(k, 5) for k in ["this is a test", "this is also a test", "", "é"]}
for name, fn in (("md5", default_md5), ("sha256", default_sha256)):
    blm = BloomFilter(est_elements=1000, false_positive_rate=0.01, hash_function=fn)
    for k in keys:
        blm.add(k)
    probes = [f"key-{i}" for i in range(250, 400)]
    G[f"bloom_{name}"] = {"hex": blm.export_hex(), "probes": probes, "membership": [int(blm.check(p)) for p in probes]}
    cms = CountMinSketch(width=500, depth=4, hash_function=fn)
    for j, k in enumerate(keys):
        cms.add(k, 1 + j % 5)
    G[f"cms_{name}"] = {"bins": list(cms._bins), "elements_added": cms.elements_added,
                        "check": [cms.check(k) for k in keys[:50]]}
    cbf = CountingBloomFilter(est_elements=500, false_positive_rate=0.05, hash_function=fn)
    for j, k in enumerate(keys):
        cbf.add(k, 1 + j % 3)
    G[f"cbf_{name}"] = {"table": list(cbf.bloom), "elements_added": cbf.elements_added}

out = Path(__file__).resolve().parent / "golden_digest.json"
out.write_text(json.dumps(G, indent=0, ensure_ascii=True))
print("wrote", out, out.stat().st_size, "bytes")
