"""Pin the C oracle's ExpandingBloomFilter / RotatingBloomFilter restatement (oracle/psk_oracle.c
psk_o_stack_*) against fixtures produced by the real reference (tests/golden/gen_golden_stack.py)."""

import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from _util import unpackbits

GS = json.loads((Path(__file__).parent / "golden" / "golden_stack.json").read_text())


def _stream(oracle, n, pool, salt):
    return [int(oracle.splitmix64(salt * 1000003 + j) % pool) for j in range(n)]


def _state(st):
    raw = st.export_bytes()
    return {"filters": st.nfilters, "counts": [int(c) for c in st.counts[: st.nfilters]], "elements_added": st.els_added,
            "sha256": hashlib.sha256(raw).hexdigest(), "nbytes": len(raw)}


def test_reference_kats(oracle):
    st = oracle.OracleStack(25, 0.05)
    assert hashlib.md5(st.export_bytes()).hexdigest() == GS["kat_empty_md5"] == "eb5769ae9babdf7b37d6ce64d58812bc"
    st = oracle.OracleStack(10, 0.05)
    st.add_keys([str(i).encode() for i in range(120)])
    assert (st.nfilters - 1, st.els_added) == (8, 120) == (GS["kat_without_force"]["expansions"], GS["kat_without_force"]["elements_added"])
    st = oracle.OracleStack(25, 0.05)
    st.add_keys([str(i).encode() for i in range(105)])
    assert st.nfilters - 1 == 3 == GS["kat_frombytes"]["expansions"]
    assert st.export_bytes().hex() == GS["kat_frombytes"]["hex"]


@pytest.mark.parametrize("name", ["ebf_small", "ebf_force", "ebf_highfpr", "rbf_small", "rbf_highfpr"])
def test_string_streams(oracle, name):
    g = GS[name]
    st = oracle.OracleStack(g["est_elements"], g["fpr"], queue=g["kw"].get("max_queue_size", 0))
    seq = _stream(oracle, g["n"], g["pool"], g["salt"])
    done = 0
    for upto in sorted(int(x) for x in g["snapshots"]):
        st.add_keys([f"k{i}".encode() for i in seq[done:upto]], force=g["force"])
        done = upto
        assert _state(st) == g["snapshots"][str(upto)]
    assert st.export_bytes().hex() == g["hex"]
    got = st.check_keys([f"k{i}".encode() for i in g["probes"]])
    assert np.array_equal(got, unpackbits(g["membership_bits"], len(g["probes"])))


@pytest.mark.parametrize("name", ["ebf_synth16", "rbf_synth16"])
def test_synthetic_streams(oracle, name):
    g = GS[name]
    st = oracle.OracleStack(g["est_elements"], g["fpr"], queue=g["kw"].get("max_queue_size", 0))
    seq = _stream(oracle, g["n"], g["pool"], g["salt"])
    pool = oracle.gen_keys16(0, g["pool"] + 2000)
    st.add_keys([bytes(pool[i]) for i in seq])
    assert _state(st) == g["final"]
    probes = list(range(0, g["probe_stop"], g["probe_step"]))
    got = st.check_keys([bytes(pool[i]) for i in probes])
    assert np.array_equal(got, unpackbits(g["membership_bits"], len(probes)))
