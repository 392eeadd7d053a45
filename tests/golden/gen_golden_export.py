#!/usr/bin/env python3
"""Generate tests/golden/golden_export.json by running the REAL reference: the text of export_c_header, export_size, the hex
and byte exports and the statistics string of small filters (bloom.py:274-338, countingbloom.py:80-123, countminsketch.py:147-166).

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_export.py [/root/reference]

The output is data only: the inputs (key strings, parameters) and what the reference wrote for them.
"""

import hashlib
import json
import sys
import tempfile
from pathlib import Path

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import probables  # noqa: E402
from probables import BloomFilter, CountingBloomFilter, CountMinSketch  # noqa: E402


def header_text(f):
    with tempfile.TemporaryDirectory() as d:
        p = Path(d) / "f.h"
        f.export_c_header(str(p))
        return p.read_text(encoding="utf-8")


def file_bytes(f):
    with tempfile.TemporaryDirectory() as d:
        p = Path(d) / "f.bin"
        f.export(str(p))
        return p.read_bytes().hex()


def text(t, est):
    """the text itself for the smallest filters, its sha256 otherwise (the fixture stays small)"""
    return t if est <= 10 else {"sha256": hashlib.sha256(t.encode("utf-8")).hexdigest(), "len": len(t)}


out = {"reference_version": getattr(probables, "__version__", "?"), "cases": []}
for cls, name in ((BloomFilter, "bloom"), (CountingBloomFilter, "cbf")):
    for est, fpr, nkeys in ((10, 0.05, 7), (100, 0.01, 60)):
        f = cls(est_elements=est, false_positive_rate=fpr)
        keys = [f"this is a test {i}" for i in range(nkeys)]
        for k in keys:
            f.add(k)
        if cls is CountingBloomFilter:
            f.add(keys[0], 5)
            f.remove(keys[1])
        out["cases"].append({
            "kind": name, "est_elements": est, "false_positive_rate": fpr, "keys": keys,
            "extra": "add(keys[0], 5); remove(keys[1])" if cls is CountingBloomFilter else "",
            "c_header": text(header_text(f), est), "export_size": f.export_size(), "export_hex": text(f.export_hex(), est), "bytes_hex": text(bytes(f).hex(), est),
            "file_hex": text(file_bytes(f), est), "str": str(f), "estimate_elements": f.estimate_elements(),
            "current_false_positive_rate": f.current_false_positive_rate(), "elements_added": f.elements_added,
        })
for width, depth, nkeys in ((8, 3, 5), (100, 5, 40)):
    c = CountMinSketch(width=width, depth=depth)
    keys = [f"this is a test {i}" for i in range(nkeys)]
    for i, k in enumerate(keys):
        c.add(k, 1 + i % 4)
    out["cases"].append({
        "kind": "cms", "width": width, "depth": depth, "keys": keys, "weights": [1 + i % 4 for i in range(nkeys)],
        "bytes_hex": bytes(c).hex(), "file_hex": file_bytes(c), "str": str(c), "elements_added": c.elements_added,
        "checks": [c.check(k) for k in keys],
    })
dst = Path(__file__).resolve().parent / "golden_export.json"
dst.write_text(json.dumps(out, indent=1) + "\n")
print("wrote", dst, len(out["cases"]), "cases")
