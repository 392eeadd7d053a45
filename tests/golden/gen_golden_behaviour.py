#!/usr/bin/env python3
"""Generate tests/golden/golden_behaviour.json by running the REAL reference: corner behaviours its own tests pin and the other
fixtures do not -- filters with every bit set, set algebra and joins between mismatched operands (None / exception type and text),
a CountMinSketch loaded under another hash family, query-type changes, invalid constructor arguments.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_behaviour.py [/root/reference]

The output is data only: the scenario (a recipe both sides can run) and what the reference answered.
"""

import json
import sys
from pathlib import Path

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import probables  # noqa: E402
from probables import BloomFilter, CountingBloomFilter, CountMeanMinSketch, CountMeanSketch, CountMinSketch  # noqa: E402
from probables.hashes import default_md5, default_sha256  # noqa: E402

HASHES = {"fnv": None, "md5": default_md5, "sha256": default_sha256}


def outcome(fn):
    try:
        r = fn()
    except Exception as e:  # noqa: BLE001  (the point is to record whatever the reference raises)
        return {"raises": type(e).__name__, "msg": str(e), "bases": [b.__name__ for b in type(e).__mro__[1:-1]]}
    if isinstance(r, float):
        return {"result": r, "repr": repr(r)}
    if hasattr(r, "export_hex"):  # a filter came back: what it holds
        return {"result": {"class": type(r).__name__, "elements_added": r.elements_added, "export_hex_len": len(r.export_hex()),
                           "checks": [int(r.check(k)) for k in keys(40)]}}
    return {"result": r}


def keys(n, tag="k"):
    return [f"{tag}-{i}" for i in range(n)]


out = {"reference_version": getattr(probables, "__version__", "?"), "full": [], "setops": [], "cms_join": [], "cms_misc": [], "ctor": []}

# ---- every bit set (bloom_test.py test_bf_all_bits_set, countingbloom_test.py test_cbf_all_bits_set)
for cls, name in ((BloomFilter, "bloom"), (CountingBloomFilter, "cbf")):
    f = cls(est_elements=10, false_positive_rate=0.05)
    for k in keys(300):
        f.add(k)
    out["full"].append({"kind": name, "est_elements": 10, "false_positive_rate": 0.05, "nkeys": 300,
                        "estimate_elements": outcome(f.estimate_elements), "current_false_positive_rate": outcome(f.current_false_positive_rate),
                        "str": str(f), "elements_added": f.elements_added})

# ---- set algebra between mismatched operands
def make(cls, est, fpr, hname, n):
    f = cls(est_elements=est, false_positive_rate=fpr, hash_function=HASHES[hname])
    for k in keys(n):
        f.add(k)
    return f


for cls, name in ((BloomFilter, "bloom"), (CountingBloomFilter, "cbf")):
    for a_args, b_args, label in (((100, 0.01, "fnv", 20), (100, 0.01, "fnv", 30), "compatible"),
                                  ((100, 0.01, "fnv", 20), (200, 0.01, "fnv", 20), "different est_elements"),
                                  ((100, 0.01, "fnv", 20), (100, 0.05, "fnv", 20), "different fpr"),
                                  ((100, 0.01, "fnv", 20), (100, 0.01, "md5", 20), "different hash family"),
                                  ((100, 0.01, "fnv", 0), (100, 0.01, "fnv", 0), "both empty"),
                                  ((100, 0.01, "fnv", 20), (100, 0.01, "fnv", 0), "second empty")):
        a, b = make(cls, *a_args), make(cls, *b_args)
        rec = {"kind": name, "label": label, "a": list(a_args), "b": list(b_args)}
        for opname in ("union", "intersection", "jaccard_index"):
            rec[opname] = outcome(lambda op=opname: getattr(a, op)(b))
        out["setops"].append(rec)
    a = make(cls, 100, 0.01, "fnv", 5)
    for opname in ("union", "intersection", "jaccard_index"):
        out["setops"].append({"kind": name, "label": "operand is not a filter", "op": opname, "operand": "int 1", "outcome": outcome(lambda op=opname: getattr(a, op)(1))})
    other = CountingBloomFilter(est_elements=100, false_positive_rate=0.01) if cls is BloomFilter else BloomFilter(est_elements=100, false_positive_rate=0.01)
    for opname in ("union", "intersection", "jaccard_index"):
        out["setops"].append({"kind": name, "label": "operand is the other filter class", "op": opname, "operand": "other", "outcome": outcome(lambda op=opname: getattr(a, op)(other))})

# ---- CountMinSketch.join (countminsketch.py:356-399)
def cms(cls, w, d, hname, n, weight=1):
    c = cls(width=w, depth=d, hash_function=HASHES[hname])
    for k in keys(n):
        c.add(k, weight)
    return c


CLS = {"min": CountMinSketch, "mean": CountMeanSketch, "meanmin": CountMeanMinSketch}
for a_args, b_args, label in ((("min", 50, 4, "fnv", 10), ("min", 50, 4, "fnv", 12), "compatible"),
                              (("min", 50, 4, "fnv", 10), ("min", 60, 4, "fnv", 10), "different width"),
                              (("min", 50, 4, "fnv", 10), ("min", 50, 5, "fnv", 10), "different depth"),
                              (("min", 50, 4, "fnv", 10), ("min", 50, 4, "md5", 10), "different hash family"),
                              (("min", 50, 4, "fnv", 10), ("mean", 50, 4, "fnv", 10), "mixed classes: min <- mean"),
                              (("meanmin", 50, 4, "fnv", 10), ("min", 50, 4, "fnv", 10), "mixed classes: meanmin <- min")):
    a = cms(CLS[a_args[0]], *a_args[1:])
    b = cms(CLS[b_args[0]], *b_args[1:])

    def run():
        a.join(b)
        return {"elements_added": a.elements_added, "checks": [a.check(k) for k in keys(14)], "bytes_hex_len": len(bytes(a).hex())}
    out["cms_join"].append({"label": label, "a": list(a_args), "b": list(b_args), "outcome": outcome(run)})
a = cms(CountMinSketch, 50, 4, "fnv", 3)
out["cms_join"].append({"label": "operand is not a sketch", "operand": "int 1", "outcome": outcome(lambda: a.join(1))})

# ---- query types, loading under another hash family
c = cms(CountMinSketch, 50, 4, "fnv", 10, 3)
rec = {"label": "query types on one sketch", "width": 50, "depth": 4, "nkeys": 10, "weight": 3, "steps": []}
for q in ("mean", "mean-min", "min", "bogus", None, "MEAN"):
    def setq(q=q):
        c.query_type = q
        return {"query_type": c.query_type, "checks": [c.check(k) for k in keys(12)]}
    rec["steps"].append({"set": q, "outcome": outcome(setq)})
out["cms_misc"].append(rec)
import tempfile  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    p = str(Path(d) / "c.cms")
    c2 = cms(CountMinSketch, 50, 4, "fnv", 10, 2)
    c2.export(p)
    back = CountMinSketch(filepath=p, hash_function=default_md5)
    out["cms_misc"].append({"label": "exported under fnv, loaded under md5", "width": 50, "depth": 4, "nkeys": 10, "weight": 2,
                            "checks_original": [c2.check(k) for k in keys(12)], "checks_loaded": [back.check(k) for k in keys(12)],
                            "elements_added": back.elements_added})

# ---- constructor arguments (countminsketch_test.py test_cms_invalid_*, bloom_test.py test_invalid_*)
for kw in ({"width": 0, "depth": 5}, {"width": 5, "depth": 0}, {"width": -1, "depth": 5}, {"width": 5, "depth": -1}, {"confidence": 0.0, "error_rate": 0.1},
           {"confidence": 0.9, "error_rate": 0.0}, {"confidence": -1, "error_rate": 0.1}, {"confidence": 0.9, "error_rate": -0.1}, {},
           {"confidence": 0.96875, "error_rate": 0.002}, {"width": 10}, {"depth": 10}):
    def mk(kw=kw):
        s = CountMinSketch(**kw)
        return {"width": s.width, "depth": s.depth, "confidence": s.confidence, "error_rate": s.error_rate}
    out["ctor"].append({"cls": "CountMinSketch", "kwargs": kw, "outcome": outcome(mk)})
for cls, name in ((BloomFilter, "BloomFilter"), (CountingBloomFilter, "CountingBloomFilter")):
    for kw in ({"est_elements": 0, "false_positive_rate": 0.1}, {"est_elements": -5, "false_positive_rate": 0.1}, {"est_elements": 10, "false_positive_rate": 0.0},
               {"est_elements": 10, "false_positive_rate": 1.0}, {"est_elements": 10, "false_positive_rate": -0.1}, {"est_elements": 10, "false_positive_rate": 1.5},
               {"est_elements": 1, "false_positive_rate": 0.999}, {"est_elements": 10}, {"false_positive_rate": 0.1}, {}, {"est_elements": 10.5, "false_positive_rate": 0.1},
               {"est_elements": "10", "false_positive_rate": 0.1}):
        def mkf(kw=kw, cls=cls):
            f = cls(**kw)
            return {"number_bits": f.number_bits, "number_hashes": f.number_hashes, "bloom_length": f.bloom_length}
        out["ctor"].append({"cls": name, "kwargs": kw, "outcome": outcome(mkf)})

dst = Path(__file__).resolve().parent / "golden_behaviour.json"
dst.write_text(json.dumps(out, indent=1) + "\n")
print("wrote", dst, {k: len(v) for k, v in out.items() if isinstance(v, list)})
