"""The oracle's set algebra (oracle/psk_oracle.c: union / intersection / jaccard_index / estimate_elements) against
fixtures produced by the REAL reference (tests/golden/gen_golden_setops.py): bloom.py:340-352,371-460 and
countingbloom.py:210-304.  CPU only."""

import json
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def setops():
    return json.loads((ROOT / "tests" / "golden" / "golden_setops.json").read_text())


def _bloom_from_hex(oracle, case, which):
    ob = oracle.OracleBloom(case["m"], case["k"])
    ob.bloom[:] = np.frombuffer(bytes.fromhex(case[f"{which}_hex"]), dtype=np.uint8)
    return ob


def test_fixture_tables_are_what_the_oracle_builds(oracle, setops):
    """the inputs of the set operations themselves: oracle inserts == the reference's tables"""
    for case in setops["bloom"]:
        for which in ("a", "b"):
            lo, hi = case[f"{which}_keys"]
            ob = oracle.OracleBloom(case["m"], case["k"])
            if hi > lo:
                ob.add_keys(oracle.gen_keys16(lo, hi - lo))
            assert ob.bloom.tobytes().hex() == case[f"{which}_hex"]


def test_bloom_union_intersection_jaccard(oracle, setops):
    for case in setops["bloom"]:
        a, b = _bloom_from_hex(oracle, case, "a"), _bloom_from_hex(oracle, case, "b")
        u, x = a.union(b), a.intersection(b)
        assert u.bloom.tobytes().hex() == case["union_hex"]
        assert x.bloom.tobytes().hex() == case["intersection_hex"]
        assert u.els_added == case["union_elements_added"] == case["union_estimate_elements"]
        assert x.els_added == case["intersection_elements_added"] == case["intersection_estimate_elements"]
        assert a.jaccard_index(b) == case["jaccard"]          # same two integers divided in binary64: exact
        assert b.jaccard_index(a) == case["jaccard_ba"]
        assert a.jaccard_index(a) == case["jaccard_self"]
    assert setops["bloom"][2]["jaccard"] == 1.0               # empty vs empty
    assert setops["bloom"][4]["union_elements_added"] == -1   # every bit set


def _cbf_from_list(oracle, case, which):
    oc = oracle.OracleCBF(case["m"], case["k"])
    oc.bloom[:] = np.array(case[f"{which}_table"], dtype=np.uint32)
    return oc


def test_cbf_inputs(oracle, setops):
    for case in setops["cbf"]:
        for which in ("a", "b"):
            oc = oracle.OracleCBF(case["m"], case["k"])
            for lo, hi, weighted in case[f"{which}_ops"]:
                w = oracle.gen_weights(lo, hi - lo).astype(np.int64) if weighted else None
                oc.update_keys(oracle.gen_keys16(lo, hi - lo), w)
            assert oc.bloom.tolist() == case[f"{which}_table"]
            assert oc.els_added == case[f"{which}_elements_added"]
            assert oc.bits_set() == case[f"{which}_bits_set"]


def test_cbf_union_intersection_jaccard(oracle, setops):
    for case in setops["cbf"]:
        a, b = _cbf_from_list(oracle, case, "a"), _cbf_from_list(oracle, case, "b")
        u, x = a.union(b), a.intersection(b)
        assert u.bloom.tolist() == case["union_table"]
        assert x.bloom.tolist() == case["intersection_table"]
        assert u.els_estimate == case["union_elements_added"] == case["union_estimate_elements"]
        assert x.els_estimate == case["intersection_elements_added"] == case["intersection_estimate_elements"]
        assert a.jaccard_index(b) == case["jaccard"]
        assert b.jaccard_index(a) == case["jaccard_ba"]
        assert a.jaccard_index(a) == case["jaccard_self"]
    one_sided = setops["cbf"][1]
    assert not any(one_sided["intersection_table"]) and one_sided["jaccard"] == 0.0
    assert setops["cbf"][2]["jaccard"] == 1.0


def test_all_cores_leg_equals_single_thread(oracle):
    """the multi-threaded baseline leg (per-thread replica + OR merge) builds the single-stream table"""
    m, k, n = 958506, 7, 40_000
    one = oracle.OracleBloom(m, k)
    one.add_keys(oracle.gen_keys16(5, n))
    for threads in (1, 3, 8):
        mt = oracle.OracleBloom(m, k)
        assert mt.insert_check_mt(5, n, threads) == n
        assert np.array_equal(mt.bloom, one.bloom)
