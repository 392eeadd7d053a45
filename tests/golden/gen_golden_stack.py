#!/usr/bin/env python3
"""Generate tests/golden/golden_stack.json by running the REAL reference's ExpandingBloomFilter /
RotatingBloomFilter (pyprobables v0.7.0, probables/blooms/expandingbloom.py).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_stack.py [/root/reference]

Data only: the recipe of every key stream (indices into `k<i>` strings or into the synthetic 16-byte key
generator of SURVEY.md 8(d)) and what the reference produced for it.
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
from probables import ExpandingBloomFilter, RotatingBloomFilter  # noqa: E402

M64 = 2**64 - 1
SEED = 0x5EED


def sm(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def key16(i):
    return struct.pack("<QQ", sm(SEED + 2 * i), sm(SEED + 2 * i + 1))


def packbits(bools):
    out = bytearray((len(bools) + 7) // 8)
    for i, b in enumerate(bools):
        if b:
            out[i >> 3] |= 1 << (i & 7)
    return out.hex()


def state(blm):
    raw = bytes(blm)
    return {
        "filters": len(blm._blooms),
        "counts": [b.elements_added for b in blm._blooms],
        "elements_added": blm.elements_added,
        "sha256": hashlib.sha256(raw).hexdigest(),
        "nbytes": len(raw),
    }


G = {"reference_version": probables.__version__, "seed": SEED}

# ---- KATs of the reference's own tests (tests/expandingbloom_test.py)
blm = ExpandingBloomFilter(est_elements=25, false_positive_rate=0.05)
G["kat_empty_md5"] = hashlib.md5(bytes(blm)).hexdigest()  # :104 eb5769ae9babdf7b37d6ce64d58812bc
blm = ExpandingBloomFilter(est_elements=10, false_positive_rate=0.05)
for i in range(120):
    blm.add(f"{i}")
G["kat_without_force"] = {"expansions": blm.expansions, "elements_added": blm.elements_added}  # :47-54 -> 8, 120
blm = ExpandingBloomFilter(est_elements=25, false_positive_rate=0.05)
for i in range(105):
    blm.add(str(i))
G["kat_frombytes"] = {"expansions": blm.expansions, "hex": bytes(blm).hex()}  # :111-126 -> 3


def stream(n, pool, salt):
    return [int(sm(salt * 1000003 + j) % pool) for j in range(n)]


# ---- string-key streams with repeats, recorded in full (small filters)
def run_strings(cls, name, est, fpr, n, pool, salt, force=False, **kw):
    blm = cls(est_elements=est, false_positive_rate=fpr, **kw)
    seq = stream(n, pool, salt)
    snaps = {}
    for j, i in enumerate(seq):
        blm.add(f"k{i}", force)
        if j + 1 in (n // 3, n):
            snaps[str(j + 1)] = state(blm)
    probes = list(range(0, pool + 40, 3))
    G[name] = {
        "est_elements": est, "fpr": fpr, "n": n, "pool": pool, "salt": salt, "force": force, "kw": kw,
        "snapshots": snaps, "hex": bytes(blm).hex(),
        "probes": probes, "membership_bits": packbits([blm.check(f"k{i}") for i in probes]),
    }


run_strings(ExpandingBloomFilter, "ebf_small", 25, 0.05, 300, 180, 1)
run_strings(ExpandingBloomFilter, "ebf_force", 10, 0.05, 100, 60, 2, force=True)
run_strings(ExpandingBloomFilter, "ebf_highfpr", 300, 0.3, 4000, 3000, 3)
run_strings(RotatingBloomFilter, "rbf_small", 20, 0.05, 400, 250, 4, max_queue_size=3)
run_strings(RotatingBloomFilter, "rbf_highfpr", 200, 0.25, 3000, 2500, 5, max_queue_size=4)

# push / pop bookkeeping (expandingbloom_test.py:202-236 style)
rbf = RotatingBloomFilter(est_elements=10, false_positive_rate=0.05, max_queue_size=3)
log = []
for step, op in enumerate(["add", "push", "add", "push", "push", "add", "pop", "add", "push"]):
    if op == "add":
        for i in range(7):
            rbf.add(f"s{step}-{i}")
    elif op == "push":
        rbf.push()
    else:
        rbf.pop()
    log.append({"op": op, "queue": rbf.current_queue_size, "counts": [b.elements_added for b in rbf._blooms],
                "elements_added": rbf.elements_added})
G["rbf_push_pop"] = {"log": log, "hex": bytes(rbf).hex()}


# ---- synthetic 16-byte keys (device-resident in the engine tests): hashes only
def run_synth(cls, name, est, fpr, n, pool, salt, **kw):
    blm = cls(est_elements=est, false_positive_rate=fpr, **kw)
    seq = stream(n, pool, salt)
    for i in seq:
        blm.add(key16(i))
    probes = list(range(0, pool + 2000, 7))
    G[name] = {"est_elements": est, "fpr": fpr, "n": n, "pool": pool, "salt": salt, "kw": kw, "final": state(blm),
               "probe_step": 7, "probe_stop": pool + 2000,
               "membership_bits": packbits([blm.check(key16(i)) for i in probes])}


run_synth(ExpandingBloomFilter, "ebf_synth16", 2000, 0.1, 30000, 20000, 6)
run_synth(RotatingBloomFilter, "rbf_synth16", 1500, 0.1, 30000, 25000, 7, max_queue_size=4)

out = Path(__file__).resolve().parent / "golden_stack.json"
out.write_text(json.dumps(G, indent=0, ensure_ascii=True))
print("wrote", out, out.stat().st_size, "bytes")
