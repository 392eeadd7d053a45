"""GPU parity tests: the HIP path (through the C ABI, via the drop-in classes) against
  * the golden fixtures produced by the real reference (tests/golden/golden.json),
  * the reference's own KATs (literal constants, cited),
  * the plain-C oracle on the same seeded inputs.
Bit-exact everywhere (integer / byte work).  Run with `-m gpu` on an MI355X.
"""

import hashlib
import struct

import numpy as np
import pytest

from _util import as_key, sha, unpackbits

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def pa():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pyprobables_amd

    return pyprobables_amd


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _table(f, dtype=np.uint8):
    return np.frombuffer(bytes(f.bloom), dtype=dtype)


# ------------------------------------------------------------------ hashing kernel
def test_fnv_hash_kernel_golden(pa, golden):
    import ctypes as C

    from pyprobables_amd import _native as N
    from pyprobables_amd.keys import pack_keys

    for case in golden["hashes"]:
        b = pack_keys([as_key(case)])
        out = np.zeros(case["depth"], dtype=np.uint64)
        N.check(N.lib().psk_fnv1a_hash(*b.args(), case["depth"], b.where, out.ctypes.data, 0, None))
        assert [int(x) for x in out] == case["hashes"], case["key"][:40]
    # a ragged batch in one launch
    cases = [c for c in golden["hashes"] if c["type"] == "bytes" and c["depth"] == 7]
    b = pack_keys([as_key(c) for c in cases])
    out = np.zeros((len(cases), 7), dtype=np.uint64)
    N.check(N.lib().psk_fnv1a_hash(*b.args(), 7, b.where, out.ctypes.data, 0, None))
    assert out.tolist() == [c["hashes"] for c in cases]
    # str batch with code points > 255 (VARLEN32 layout)
    cases = [c for c in golden["hashes"] if c["type"] == "str"]
    b = pack_keys([as_key(c) for c in cases])
    assert b.layout == N.KEYS_VARLEN32
    out = np.zeros((len(cases), 7), dtype=np.uint64)
    N.check(N.lib().psk_fnv1a_hash(*b.args(), 7, b.where, out.ctypes.data, 0, None))
    assert out.tolist() == [c["hashes"] for c in cases]
    _ = C


# ------------------------------------------------------------------ Bloom
def test_bloom_kat_export_hex_and_membership(pa, golden):
    # reference tests/bloom_test.py:233-286
    g = golden["bloom_small"]
    blm = pa.BloomFilter(est_elements=10, false_positive_rate=0.05)
    for k in g["keys"]:
        blm.add(k)
    assert blm.export_hex() == "6da491461a6bba4d000000000000000a000000000000000a3d4ccccd" == g["export_hex"]
    assert bytes(blm).hex() == g["bytes_hex"]
    assert str(blm) == g["str"]
    assert [blm.check(k) for k in g["check_keys"]] == g["check"]
    assert [bool(x) for x in blm.check_many(g["check_keys"])] == g["check"]
    assert ("this is a test 3" in blm) is True and ("this is a test 15" in blm) is False
    assert blm.estimate_elements() == g["estimate_elements"]
    assert blm.current_false_positive_rate() == g["current_fpr"]
    assert blm.elements_added == 10
    # load hex / bytes round trips (bloom_test.py:267-286, 343-360)
    b2 = pa.BloomFilter(hex_string=g["export_hex"])
    assert [b2.check(k) for k in g["check_keys"]] == g["check"]
    assert b2.elements_added == 10 and b2.number_bits == 63 and b2.number_hashes == 4
    b3 = pa.BloomFilter.frombytes(bytes(blm))
    assert bytes(b3) == bytes(blm)


def test_bloom_kat_md5(pa, golden):
    # reference tests/bloom_test.py:323-341
    blm = pa.BloomFilter(est_elements=10, false_positive_rate=0.05)
    blm.add("this is a test")
    assert hashlib.md5(bytes(blm)).hexdigest() == "8d27e30e1c5875b0edcf7413c7bdb221" == golden["bloom_one"]["md5_bytes"]


def test_bloom_cfg1_golden(pa, golden, oracle):
    # BASELINE configs[0]: BloomFilter(1000, 0.05), 10k synthetic 16-byte keys
    g = golden["bloom_cfg1"]
    keys = oracle.gen_keys16(0, g["n_keys"])
    for src in (keys, _dev(keys)):  # host-staged and device-resident batches
        blm = pa.BloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"])
        assert (blm.number_hashes, blm.number_bits) == (g["k"], g["m"])
        blm.add_many(src)
        assert bytes(blm.bloom).hex() == g["table_hex"]
        assert sha(bytes(blm)) == g["sha256_bytes"]
        assert blm._cnt_number_bits_set() == g["bits_set"]
        assert blm.elements_added == g["elements_added"]
        res = blm.check_many(src)
        res = res.cpu().numpy() if hasattr(res, "cpu") else res
        assert bool(res.all()) == g["all_checks_true"]
        fresh = blm.check_many(oracle.gen_keys16(10000, 1000))
        assert np.array_equal(fresh.astype(np.uint8), unpackbits(g["check_fresh_10000_10999"], 1000))


def test_bloom_non_pow2_modulo_golden(pa, golden, oracle):
    g = golden["bloom_np2"]
    blm = pa.BloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"])
    assert blm.number_bits == g["m"] == 958506
    blm.add_many(_dev(oracle.gen_keys16(0, g["n_keys"])))
    assert sha(bytes(blm.bloom)) == g["sha256_table"]
    assert blm._cnt_number_bits_set() == g["bits_set"]
    lo, hi = g["check_range"]
    dk = _dev(oracle.gen_keys16(lo, hi - lo))
    res = blm.check_many(dk).cpu().numpy().astype(np.uint8)
    assert int(res.sum()) == g["positives"]
    assert np.array_equal(res, unpackbits(g["membership_bits"], hi - lo))
    assert sha(res.tobytes()) == g["sha256_membership_bytes"]
    assert blm.estimate_elements() == g["estimate_elements"]
    # ballot bitmap + hit count variant
    bits, hits = blm.check_many_bits(dk)
    assert int(hits.item()) == g["positives"]
    got = np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder="little")[: hi - lo]
    assert np.array_equal(got, res)
    bits_h, hits_h = blm.check_many_bits(oracle.gen_keys16(lo, hi - lo))
    assert hits_h == g["positives"] and np.array_equal(bits_h, bits.cpu().numpy().view(np.uint64))


def test_bloom_alt_hashes_golden(pa, golden):
    g = golden["bloom_alt"]
    blm = pa.BloomFilter(est_elements=10, false_positive_rate=0.05)
    blm.add_alt(g["hashes"])  # more hashes than k: only the first k are used (bloom.py:246)
    assert bytes(blm.bloom).hex() == g["table_hex"]
    assert blm.check_alt(g["hashes"]) == g["check_same"]
    assert blm.check_alt([1, 2, 3, 4]) == g["check_other"]
    with pytest.raises(ValueError):
        blm.add_alt([1, 2])  # fewer than k


def test_bloom_varlen_and_unicode_golden(pa, golden):
    g = golden["bloom_varlen"]
    keys = [bytes.fromhex(x) for x in g["keys_hex"]]  # ragged, includes empty keys
    blm = pa.BloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"])
    blm.add_many(keys[: g["n_added"]])
    assert bytes(blm.bloom).hex() == g["table_hex"]
    assert np.array_equal(blm.check_many(keys).astype(np.uint8), unpackbits(g["membership_bits"], len(keys)))
    g = golden["bloom_unicode"]
    blm = pa.BloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"])
    blm.add_many(g["keys"][::2])  # str keys with code points > 255
    assert bytes(blm.bloom).hex() == g["table_hex"]
    assert np.array_equal(blm.check_many(g["keys"]).astype(np.uint8), unpackbits(g["membership_bits"], len(g["keys"])))


@pytest.mark.parametrize("key_len", [1, 3, 4, 7, 8, 12, 16, 20, 33, 64])
def test_bloom_fixed_lengths_vs_oracle(pa, oracle, key_len):
    rng = np.random.default_rng(key_len)
    keys = rng.integers(0, 256, size=(5000, key_len), dtype=np.uint8)
    blm = pa.BloomFilter(est_elements=4000, false_positive_rate=0.02)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(keys[:3000]))
    ob.add_keys(keys[:3000])
    assert np.array_equal(_table(blm), ob.bloom)
    assert np.array_equal(blm.check_many(keys).astype(np.uint8), ob.check_keys(keys))


@pytest.mark.parametrize("est,fpr", [(1, 0.9), (5, 0.5), (1000, 0.001), (3000, 1e-9), (200, 1e-30)])
def test_bloom_k_sweep_vs_oracle(pa, oracle, est, fpr):
    # k = 1, 1, 10, 30, 100 (several 8-wide hash groups + remainders); m = 1 included
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
    keys = oracle.gen_keys16(0, 2 * est + 10)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(keys[:est])
    ob.add_keys(keys[:est])
    assert np.array_equal(_table(blm), ob.bloom)
    assert np.array_equal(blm.check_many(_dev(keys)).cpu().numpy().astype(np.uint8), ob.check_keys(keys))


def test_bloom_empty_batch_and_clear(pa, oracle):
    blm = pa.BloomFilter(est_elements=100, false_positive_rate=0.01)
    blm.add_many([])
    assert blm.elements_added == 0 and blm._cnt_number_bits_set() == 0
    assert blm.check_many([]).shape == (0,)
    blm.add_many(np.zeros((0, 16), dtype=np.uint8))
    blm.add_many([b"", b""])  # empty keys are legal: hash = seeded offset basis
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    ob.add_varlen([b""])
    assert np.array_equal(_table(blm), ob.bloom)
    assert blm.check(b"") is True and blm.elements_added == 2
    blm.clear()
    assert blm.elements_added == 0 and blm._cnt_number_bits_set() == 0


def test_bloom_custom_hash_function_plugin(pa):
    # reference tests/bloom_test.py:489-574: decorator based and raw callables
    from pyprobables_amd import default_md5, default_sha256, hash_with_depth_int

    @hash_with_depth_int
    def my_hash(key, depth=1, encoding="utf-8"):
        return int(hashlib.sha512(key.encode(encoding)).hexdigest(), 16) & (2**64 - 1)

    for hf in (my_hash, default_md5, default_sha256):
        blm = pa.BloomFilter(est_elements=10, false_positive_rate=0.05, hash_function=hf)
        assert blm.hash_function is hf
        words = [f"this is a test {i}" for i in range(10)]
        blm.add_many(words[:5])
        blm.add(words[5])
        # expected table from the same hashes applied by hand (bloom.py:246-249)
        exp = np.zeros(8, dtype=np.uint8)
        for w in words[:6]:
            for h in hf(w, 4):
                exp[(h % 63) // 8] |= 1 << ((h % 63) % 8)
        assert np.array_equal(_table(blm), exp)
        assert all(blm.check_many(words[:6])) and blm.check(words[0])
        assert blm.hashes("abc") == hf("abc", 4)


def test_bloom_union_intersection_jaccard_vs_host(pa, oracle):
    a = pa.BloomFilter(est_elements=2000, false_positive_rate=0.01)
    b = pa.BloomFilter(est_elements=2000, false_positive_rate=0.01)
    a.add_many(oracle.gen_keys16(0, 1500))
    b.add_many(oracle.gen_keys16(1000, 1500))
    ta, tb = _table(a), _table(b)
    u, i = a.union(b), a.intersection(b)
    assert np.array_equal(_table(u), ta | tb)
    assert np.array_equal(_table(i), ta & tb)
    cu, ci = int(np.unpackbits(ta | tb).sum()), int(np.unpackbits(ta & tb).sum())
    assert a.jaccard_index(b) == ci / cu
    assert u.elements_added == u.estimate_elements()
    c = pa.BloomFilter(est_elements=2001, false_positive_rate=0.01)
    with pytest.raises(pa.SimilarityError):
        a.union(c)
    with pytest.raises(TypeError):
        a.union("nope")
    assert pa.BloomFilter(10, 0.05).jaccard_index(pa.BloomFilter(10, 0.05)) == 1.0  # bloom_test.py:225-231


def test_bloom_idempotence_and_roundtrip_property(pa, oracle):
    # size-independent properties: re-inserting changes nothing; everything inserted is found
    blm = pa.BloomFilter(est_elements=50000, false_positive_rate=0.01)
    dk = _dev(oracle.gen_keys16(0, 40000))
    blm.add_many(dk)
    t1 = bytes(blm.bloom)
    blm.add_many(dk)
    assert bytes(blm.bloom) == t1
    assert bool(blm.check_many(dk).all())


# ------------------------------------------------------------------ CountMinSketch
def test_cms_kat_md5_and_returns(pa, golden):
    # reference tests/countminsketch_test.py:187-203 and :76-109
    cms = pa.CountMinSketch(width=1000, depth=5)
    assert cms.add("this is a test", 100) == 100 == golden["cms_one"]["add_return"]
    assert hashlib.md5(bytes(cms)).hexdigest() == "fb1c39dd1a73f1ef0d7fc79f60fc028e" == golden["cms_one"]["md5_bytes"]
    cms = pa.CountMinSketch(width=1000, depth=5)
    assert [cms.add("this is a test") for _ in range(4)] == [1, 2, 3, 4]
    assert cms.elements_added == 4
    assert cms.remove("this is a test") == 3 and cms.elements_added == 3
    assert cms.remove("this is a test", 2) == 1
    assert cms.check("this is a test") == 1 and ("this is a test" in cms)
    assert cms.check("this is not a test") == 0 and ("this is not a test" not in cms)


def test_cms_stream_golden(pa, golden, oracle):
    g = golden["cms_stream"]
    n, d = g["n_updates"], g["n_distinct"]
    keys = oracle.gen_keys16(0, d)
    stream_keys = keys[np.arange(n) % d]
    w = oracle.gen_weights(0, n)
    for on_device in (False, True):
        cms = pa.CountMinSketch(width=g["width"], depth=g["depth"])
        if on_device:
            cms.add_many(_dev(stream_keys), _dev(w))
        else:
            cms.add_many(stream_keys, w)
        assert sha(bytes(cms._bins)) == g["sha256_bins"]
        assert cms.elements_added == g["elements_added"]
        assert cms.check_many(keys[:200]).tolist() == g["check_0_199"]
        assert cms.check_many(_dev(oracle.gen_keys16(5000, 100))).cpu().tolist() == g["check_fresh_5000_5099"]
        cms.query_type = "mean"
        assert cms.check_many(keys[:200]).tolist() == g["mean_0_199"]
        cms.query_type = "mean-min"
        assert cms.check_many(keys[:200]).tolist() == g["meanmin_0_199"]
        cms.query_type = "min"
        cms.remove_many(keys[np.arange(20000) % d], oracle.gen_weights(0, 20000))
        a = g["after_remove_20000"]
        assert sha(bytes(cms._bins)) == a["sha256_bins"]
        assert cms.elements_added == a["elements_added"]
        assert cms.check_many(keys[:50]).tolist() == a["check_0_49"]
        assert cms.batch_diagnostics()["saturated"] == 0


def test_cms_ordered_returns_golden(pa, golden, oracle):
    keys = oracle.gen_keys16(0, 30)
    w = oracle.gen_weights(0, 300).astype(np.int64)
    idx = np.arange(300) % 23
    signed = np.where(np.arange(300) % 5 == 4, -w, w)
    for g in golden["cms_ordered"]:
        cms = pa.CountMinSketch(width=g["width"], depth=g["depth"])
        cms.query_type = g["query"]
        rets = cms.update_ordered(keys[idx], signed)
        assert rets.tolist() == g["returns"], g["query"]
        assert list(cms._bins) == g["bins"]
        assert cms.elements_added == g["elements_added"]
        assert cms.check_many(keys).tolist() == g["checks"]
        assert [cms.check(bytes(k)) for k in keys[:5]] == g["checks"][:5]


def test_cms_saturation_golden(pa, golden):
    # int32 rails: reference tests/countminsketch_test.py:262-278 style, through add_alt / remove_alt
    g = golden["cms_saturation"]
    cms = pa.CountMinSketch(width=g["width"], depth=g["depth"])
    for st in g["steps"]:
        r = cms.add_alt([1, 10, 19], st["n"]) if st["op"] == "add" else cms.remove_alt([1, 10, 19], st["n"])
        assert r == st["ret"]
        assert list(cms._bins) == st["bins"]
        assert cms.elements_added == st["elements_added"]


def test_cms_batch_saturating_path_vs_oracle(pa, oracle):
    # weights big enough that the wrap-free bound fails: the CAS path must clamp exactly like the reference
    keys = oracle.gen_keys16(0, 64)
    w = np.full(64, 2**30, dtype=np.int32)
    cms = pa.CountMinSketch(width=16, depth=3)
    oc = oracle.OracleCMS(16, 3)
    cms.add_many(_dev(keys), _dev(w))
    oc.add_keys(keys, w)
    assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
    assert int(oc.bins.max()) == 2**31 - 1
    assert cms.batch_diagnostics()["saturated"] > 0
    assert cms.elements_added == oc.els_added
    cms.remove_many(keys, w)  # all-negative batch: clamps at INT32_MIN the same way for any order
    cms.remove_many(keys, w)
    cms.remove_many(keys, w)
    for _ in range(3):
        oc.remove_keys(keys, w)
    assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
    assert int(oc.bins.min()) == -(2**31)


def test_cms_non_pow2_width_and_join_golden(pa, golden, oracle):
    g = golden["cms_join"]
    c1 = pa.CountMinSketch(width=g["width"], depth=g["depth"])
    c2 = pa.CountMinSketch(width=g["width"], depth=g["depth"])
    c1.add_many(oracle.gen_keys16(0, 200), oracle.gen_weights(0, 200))
    c2.add_many(oracle.gen_keys16(100, 200), oracle.gen_weights(7, 200))
    c1.join(c2)
    assert list(c1._bins) == g["bins"] and c1.elements_added == g["elements_added"]
    # confidence / error-rate constructor + export round trip
    for s in golden["cms_sizing"]:
        c = pa.CountMinSketch(confidence=s["confidence"], error_rate=s["error_rate"])
        assert (c.width, c.depth) == (s["width"], s["depth"])
    c3 = pa.CountMinSketch.frombytes(bytes(c1))
    assert bytes(c3) == bytes(c1) and c3.elements_added == c1.elements_added
    with pytest.raises(pa.CountMinSketchError):
        c1.join(pa.CountMinSketch(width=65, depth=3))


def test_cms_subclasses_and_str(pa):
    assert pa.CountMeanSketch(width=100, depth=3).query_type == "mean"
    assert pa.CountMeanMinSketch(width=100, depth=3).query_type == "mean-min"
    cms = pa.CountMinSketch(width=1000, depth=5)
    cms.add("this is a test", 100)
    assert str(cms) == "Count-Min Sketch:\n\tWidth: 1000\n\tDepth: 5\n\tConfidence: 0.96875\n\tError Rate: 0.002\n\tElements Added: 100"


# ------------------------------------------------------------------ CountingBloomFilter
CBF_HEX_KAT = (  # reference tests/countingbloom_test.py:200-221
    "01000000000000000100000002000000000000000100000001000000"
    "00000000000000000000000001000000000000000000000002000000"
    "00000000010000000200000000000000000000000000000001000000"
    "00000000000000000200000000000000010000000200000000000000"
    "00000000000000000100000000000000000000000100000000000000"
    "01000000020000000000000000000000000000000100000001000000"
    "00000000010000000000000001000000020000000000000000000000"
    "01000000000000000100000001000000010000000000000001000000"
    "03000000000000000100000001000000000000000000000001000000"
    "000000000000000a000000000000000a3d4ccccd"
)


def test_cbf_kat_hex_md5_str(pa, golden):
    cbf = pa.CountingBloomFilter(est_elements=10, false_positive_rate=0.05)
    for i in range(10):
        assert cbf.add(f"this is a test {i}") >= 1
    assert cbf.export_hex() == CBF_HEX_KAT == golden["cbf_small"]["export_hex"]
    assert str(cbf) == golden["cbf_small"]["str"]
    assert bytes(cbf).hex() == golden["cbf_small"]["bytes_hex"]
    # batch insert gives the same table
    cbf2 = pa.CountingBloomFilter(est_elements=10, false_positive_rate=0.05)
    cbf2.add_many([f"this is a test {i}" for i in range(10)])
    assert cbf2.export_hex() == CBF_HEX_KAT and cbf2.elements_added == 10
    # reference tests/countingbloom_test.py:106-144
    cbf = pa.CountingBloomFilter(est_elements=10, false_positive_rate=0.01)
    for word in ["test", "out", "the", "counting", "bloom", "filter", "test", "Test", "out", "test"]:
        cbf.add(word)
    assert hashlib.md5(bytes(cbf)).hexdigest() == "0b83c837da30e25f768f0527c039d341"
    assert cbf.check("test") == 3 and cbf.check("out") == 2 and cbf.check("nope") == 0
    c2 = pa.CountingBloomFilter.frombytes(bytes(cbf))
    assert bytes(c2) == bytes(cbf)
    c3 = pa.CountingBloomFilter(hex_string=cbf.export_hex())
    assert bytes(c3) == bytes(cbf)


def test_cbf_stream_golden(pa, golden, oracle):
    # the cfg-4 shaped well-formed add/remove stream, unordered batch kernels
    g = golden["cbf_stream"]
    B = g["B"]
    for on_device in (False, True):
        cbf = pa.CountingBloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"])
        assert (cbf.number_bits, cbf.number_hashes) == (g["m"], g["k"])
        for bt in range(4):
            ka = oracle.gen_keys16(bt * B, B)
            cbf.add_many(_dev(ka) if on_device else ka)
            if bt >= 1:
                kr = oracle.gen_keys16((bt - 1) * B, B // 2)
                cbf.remove_many(_dev(kr) if on_device else kr)
        assert sha(bytes(cbf.bloom)) == g["sha256_table"]
        assert cbf.elements_added == g["elements_added"]
        tab = _table(cbf, np.uint32)
        assert int(tab.sum()) == g["sum"] and int(tab.max()) == g["max"]
        assert cbf.check_many(oracle.gen_keys16(0, 100)).tolist() == g["check_0_99"]
        assert cbf.check_many(_dev(oracle.gen_keys16(19950, 100))).cpu().tolist() == g["check_19950_20049"]
        assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}


def test_cbf_weighted_golden(pa, golden, oracle):
    g = golden["cbf_weighted"]
    keys = oracle.gen_keys16(0, 1000)
    cbf = pa.CountingBloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"])
    cbf.add_many(keys[np.arange(3000) % 1000], oracle.gen_weights(0, 3000))
    cbf.remove_many(keys, oracle.gen_weights(0, 1000))
    assert sha(bytes(cbf.bloom)) == g["sha256_table"]
    assert cbf.elements_added == g["elements_added"]
    assert cbf.check_many(keys[:100]).tolist() == g["check_0_99"]


def test_cbf_ordered_ill_formed_stream_golden(pa, golden, oracle):
    # removes of absent keys, partial removes: exact only in order -> the ordered device kernel
    g = golden["cbf_ordered"]
    ops = np.array(g["ops"], dtype=np.int64)
    keys = np.concatenate([oracle.gen_keys16(int(i), 1) for i in ops[:, 0]])
    cbf = pa.CountingBloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"])
    rets = cbf.update_ordered(keys, ops[:, 1])
    assert rets.tolist() == g["returns"]
    assert list(cbf.bloom) == g["table"]
    assert cbf.elements_added == g["elements_added"]


def test_cbf_saturation_golden(pa, golden):
    # reference tests/countingbloom_test.py:407-459: duplicate indices, 2^32-1 saturation, frozen on remove
    g = golden["cbf_saturation"]
    cbf = pa.CountingBloomFilter(est_elements=10, false_positive_rate=0.05)
    for st in g["steps"]:
        r = cbf.add_alt(g["hashes"], st["n"]) if st["op"] == "add" else cbf.remove_alt(g["hashes"], st["n"])
        assert r == st["ret"], st
        tab = cbf.bloom
        assert (tab[5], tab[6], tab[7]) == (st["c5"], st["c6"], st["c7"])
        assert cbf.elements_added == st["elements_added"]
    assert cbf.check_alt(g["hashes"]) == 2**32 - 1


def test_cbf_batch_saturating_path_vs_oracle(pa, oracle):
    keys = oracle.gen_keys16(0, 200)
    w = np.full(200, 2**31, dtype=np.uint32)
    cbf = pa.CountingBloomFilter(est_elements=10, false_positive_rate=0.05)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    cbf.add_many(_dev(keys), w)
    oc.update_keys(keys, w.astype(np.int64))
    assert np.array_equal(_table(cbf, np.uint32), oc.bloom)
    assert int(oc.bloom.max()) == 2**32 - 1 and cbf.batch_diagnostics()["saturated"] > 0
    assert cbf.elements_added == oc.els_added


def test_cbf_union_intersection_jaccard(pa, oracle):
    a = pa.CountingBloomFilter(est_elements=500, false_positive_rate=0.01)
    b = pa.CountingBloomFilter(est_elements=500, false_positive_rate=0.01)
    a.add_many(oracle.gen_keys16(0, 300))
    b.add_many(oracle.gen_keys16(200, 300))
    ta, tb = _table(a, np.uint32), _table(b, np.uint32)
    assert np.array_equal(_table(a.union(b), np.uint32), ta + tb)
    assert np.array_equal(_table(a.intersection(b), np.uint32), np.where((ta > 0) & (tb > 0), ta + tb, 0))
    assert a.jaccard_index(b) == int(((ta > 0) & (tb > 0)).sum()) / int(((ta > 0) | (tb > 0)).sum())


# ------------------------------------------------------------------ larger randomized cross-checks
@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 257, 100003])
def test_ragged_batch_sizes_all_structures(pa, oracle, n):
    keys = oracle.gen_keys16(7, n)
    w = oracle.gen_weights(3, n)
    dk, dw = _dev(keys), _dev(w)
    blm = pa.BloomFilter(est_elements=200000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(dk)
    ob.add_keys(keys)
    assert np.array_equal(_table(blm), ob.bloom)
    probe = oracle.gen_keys16(0, n + 50)
    assert np.array_equal(blm.check_many(_dev(probe)).cpu().numpy().astype(np.uint8), ob.check_keys(probe))
    cms = pa.CountMinSketch(width=5003, depth=4)
    oc = oracle.OracleCMS(5003, 4)
    cms.add_many(dk, dw)
    oc.add_keys(keys, w)
    assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
    assert np.array_equal(cms.check_many(_dev(probe)).cpu().numpy(), oc.check_keys(probe).astype(np.int32))
    cbf = pa.CountingBloomFilter(est_elements=50000, false_positive_rate=0.01)
    ocb = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    cbf.add_many(dk, w.astype(np.uint32))
    ocb.update_keys(keys, w.astype(np.int64))
    assert np.array_equal(_table(cbf, np.uint32), ocb.bloom)
    assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().view(np.uint32), ocb.check_keys(probe))
    assert cbf.elements_added == ocb.els_added and cms.elements_added == oc.els_added


# ------------------------------------------------------------------ merge kernels (local half of the multi-GPU merge)
def test_or_reduce_slices_kernel_and_single_rank_merge(pa):
    import ctypes as C

    from pyprobables_amd import _native as N
    from pyprobables_amd import parallel

    rng = np.random.default_rng(1)
    for nslices, words in [(2, 4), (8, 1 << 16), (3, 1000), (5, 12)]:
        src = rng.integers(-2**31, 2**31 - 1, size=(nslices, words), dtype=np.int64).astype(np.int32)
        d_src = _dev(src.reshape(-1))
        d_dst = torch.zeros(words, dtype=torch.int32, device="cuda")
        parallel.hip_or_reduce(d_dst, d_src, nslices, words)
        assert np.array_equal(d_dst.cpu().numpy(), np.bitwise_or.reduce(src, axis=0))
    with pytest.raises(ValueError):  # slices must be 16-byte multiples
        N.check(N.lib().psk_or_reduce_slices(d_dst.data_ptr(), d_src.data_ptr(), 2, 3, 0, None))
    _ = C


def test_merge_path_on_rccl_single_rank(pa, oracle, monkeypatch):
    """drive the real collective composition (all_to_all_single / OR kernel / all_gather_into_tensor / all_reduce)
    through RCCL on one GPU: catches dtype / API problems the gloo tests cannot see"""
    import os
    import socket

    import torch.distributed as dist

    from pyprobables_amd import parallel

    if dist.is_initialized():
        pytest.skip("a process group already exists")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        monkeypatch.setenv("PSK_FORCE_MERGE_PATH", "1")
        keys = oracle.gen_keys16(0, 50_000)
        blm = pa.BloomFilter(est_elements=100000, false_positive_rate=0.01)  # 958506 bits: padded, not a multiple of R*16
        blm.add_many(_dev(keys))
        before = bytes(blm.bloom)
        parallel.merge_bloom(blm)
        assert bytes(blm.bloom) == before and blm.elements_added == 50_000
        cms = pa.CountMinSketch(width=1009, depth=4)
        cms.add_many(_dev(keys), _dev(oracle.gen_weights(0, 50_000)))
        bins, els = bytes(cms._bins), cms.elements_added
        parallel.merge_counters(cms)
        assert bytes(cms._bins) == bins and cms.elements_added == els
        cbf = pa.CountingBloomFilter(est_elements=20000, false_positive_rate=0.01)
        cbf.add_many(_dev(keys))
        tab = bytes(cbf.bloom)
        parallel.merge_counters(cbf)
        assert bytes(cbf.bloom) == tab and cbf.elements_added == 50_000
        cbf.add_many(_dev(keys))  # the wrap-free bound was rescanned: adds still work
        assert cbf.elements_added == 100_000
    finally:
        dist.destroy_process_group()


def test_bloom_per_key_add_loop_is_write_combined(pa, oracle):
    """the reference's usage pattern -- `for key in keys: blm.add(key)` -- reaches the GPU as batches; every read
    (check / export / in / stats) sees all prior adds"""
    keys = [bytes(k) for k in oracle.gen_keys16(0, 70_000)] + ["str key %d" % i for i in range(100)] + ["ключ-€"]
    blm = pa.BloomFilter(est_elements=100_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    for i, k in enumerate(keys):
        blm.add(k)
        if i == 10:
            assert blm.check(keys[3]) and (keys[10] in blm)       # a read in the middle flushes
            assert blm.elements_added == 11
    hs = np.array([oracle.default_fnv_1a(k, blm.number_hashes) for k in keys], dtype=np.uint64)
    ob.add_hashes(hs)
    assert blm.elements_added == len(keys)
    assert np.array_equal(_table(blm), ob.bloom)                  # export path flushes the tail
    assert blm.check(keys[-1]) and blm.check("str key 7")
    blm.add("late key")
    assert blm.estimate_elements() > 0 and "late key" in blm
    blm.add("dropped by clear")
    blm.clear()
    assert blm.elements_added == 0 and blm._cnt_number_bits_set() == 0
    with pytest.raises(TypeError):
        blm.add(123)


def test_work_follows_torchs_current_stream(pa, oracle):
    """device batches are enqueued on torch's CURRENT stream: two filters driven from two side streams at the same time
    (independent handles), then joined -- results as if run one after the other"""
    n = 400_000
    keys = oracle.gen_keys16(123, n)
    dk = torch.from_numpy(keys).cuda()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    filters = [pa.BloomFilter(est_elements=2_000_000, false_positive_rate=0.01) for _ in streams]
    outs = []
    for st, blm, part in zip(streams, filters, (dk[: n // 2], dk[n // 2:])):
        with torch.cuda.stream(st):
            blm.add_many(part)
            outs.append(blm.check_many(dk))
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    for blm, res, (lo, hi) in zip(filters, outs, ((0, n // 2), (n // 2, n))):
        ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
        ob.add_keys(keys[lo:hi])
        assert np.array_equal(np.frombuffer(bytes(blm.bloom), dtype=np.uint8), ob.bloom)
        assert np.array_equal(res.cpu().numpy().astype(np.uint8), ob.check_keys(keys))


def test_exports_match_the_reference_text_for_text(pa, tmp_path):
    """export_c_header / export / export_hex / bytes / export_size / the statistics string of small filters against what the REAL
    reference wrote for the same keys (tests/golden/golden_export.json, generated by tests/golden/gen_golden_export.py:
    bloom.py:274-338, countingbloom.py:80-123, countminsketch.py:147-166), and the files load back (frombytes / filepath)"""
    import hashlib
    import json
    from pathlib import Path

    cases = json.loads((Path(__file__).resolve().parent / "golden" / "golden_export.json").read_text())["cases"]

    def same(got: str, want):
        if isinstance(want, dict):
            return len(got) == want["len"] and hashlib.sha256(got.encode("utf-8")).hexdigest() == want["sha256"]
        return got == want

    for c in cases:
        if c["kind"] == "cms":
            s = pa.CountMinSketch(width=c["width"], depth=c["depth"])
            for kx, w in zip(c["keys"], c["weights"]):
                s.add(kx, w)
            assert bytes(s).hex() == c["bytes_hex"] and str(s) == c["str"] and s.elements_added == c["elements_added"]
            assert [s.check(kx) for kx in c["keys"]] == c["checks"]
            f = tmp_path / "s.cms"
            s.export(str(f))
            assert f.read_bytes().hex() == c["file_hex"]
            back = pa.CountMinSketch(filepath=str(f))
            assert bytes(back).hex() == c["bytes_hex"] and [back.check(kx) for kx in c["keys"]] == c["checks"]
            continue
        cls = pa.BloomFilter if c["kind"] == "bloom" else pa.CountingBloomFilter
        flt = cls(est_elements=c["est_elements"], false_positive_rate=c["false_positive_rate"])
        for kx in c["keys"]:
            flt.add(kx)
        if c["kind"] == "cbf":
            flt.add(c["keys"][0], 5)
            flt.remove(c["keys"][1])
        h = tmp_path / "f.h"
        flt.export_c_header(str(h))
        assert same(h.read_text(encoding="utf-8"), c["c_header"]), c["kind"]
        assert flt.export_size() == c["export_size"]
        assert same(flt.export_hex(), c["export_hex"]) and same(bytes(flt).hex(), c["bytes_hex"])
        f = tmp_path / "f.blm"
        flt.export(str(f))
        assert same(f.read_bytes().hex(), c["file_hex"])
        assert str(flt) == c["str"]
        assert flt.estimate_elements() == c["estimate_elements"] and flt.elements_added == c["elements_added"]
        assert flt.current_false_positive_rate() == c["current_false_positive_rate"]
        back = cls(filepath=str(f))
        assert same(bytes(back).hex(), c["bytes_hex"]) and all(back.check(kx) for kx in c["keys"] if kx != c["keys"][1] or c["kind"] == "bloom")
        again = cls.frombytes(f.read_bytes())
        assert same(again.export_hex(), c["export_hex"])


def _plain(k):
    return bytes(k) if isinstance(k, (bytearray, memoryview)) else k


@pytest.mark.parametrize("poll_us", [200, 0, 1])  # (1 us: every poll gives up -- the stream wait takes over, and after eight in a row the handle stops polling)
def test_single_key_calls_equal_the_batch_calls_for_every_kind_of_key(pa, poll_us):
    """`key in blm`, `cms.add(key)`, `cbf.remove(key)` ... go through preallocated argument / result words (`_base.OneKey`) when the engine
    hashes the key itself; keys that need the general packer (wide code points, bytearray, memoryview) take it.  On the device a one-op
    call spreads its probes over the lanes of a wave (k_cbf_ordered / k_cms_ordered, psk_device.hpp) and posts a completion mailbox the
    host polls (`host_poll_us`; 0 = wait for the stream).  Either way the answers, the tables and elements_added are those of the SAME
    ops run as one ordered batch -- the literal one-lane loop -- for every kind of key the reference accepts (hashes.py:98: a str by code
    point, anything else by byte value), for probes that collide (tiny tables) and at the counters' rails."""
    from pyprobables_amd import _native as N

    kinds = ["plain ascii", "café ÿ", "wide € \U0001f600", b"raw \x00 bytes \xff", bytearray(b"byte array"),
             memoryview(b"memory view"), "", b"", "x" * 300]
    N.set_option("host_poll_us", poll_us)
    try:
        blm, ref = pa.BloomFilter(est_elements=1000, false_positive_rate=0.01), pa.BloomFilter(est_elements=1000, false_positive_rate=0.01)
        for k in kinds[::2]:
            blm.add(k)
        ref.add_many([_plain(k) for k in kinds[::2]])
        assert blm.export_hex() == ref.export_hex()
        for k in kinds:
            assert blm.check(k) is bool(ref.check_many([_plain(k), "other"])[0]) and (k in blm) is blm.check(k)
        assert blm.check("never added") is False

        for query in ("min", "mean", "mean-min"):
            for width in (1000, 2):
                cms, ref = pa.CountMinSketch(width=width, depth=5), pa.CountMinSketch(width=width, depth=5)
                cms.query_type = ref.query_type = query
                ops, got = [], []
                for i, k in enumerate(kinds):
                    for w in (i + 1, 1, -2):
                        got.append(cms.add(k, w) if w > 0 else cms.remove(k, -w))
                        ops.append((_plain(k), w))
                for w in ((1 << 31) - 5, 7, -(1 << 40), 1 << 40):  # (the int32 rails: countminsketch.py:276-283)
                    got.append(cms.add("rail", w) if w > 0 else cms.remove("rail", -w))
                    ops.append(("rail", w))
                want = ref.update_ordered([k for k, _ in ops], [w for _, w in ops])
                assert got == [int(v) for v in want]
                assert bytes(cms._tab.read()) == bytes(ref._tab.read()) and cms.elements_added == ref.elements_added
                assert [cms.check(k) for k in kinds] == [int(v) for v in ref.check_many([_plain(k) for k in kinds])]

        for est in (1000, 2):  # (est_elements = 2: a handful of counters, so the k probes of a key collide)
            cbf, ref = (pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.05) for _ in range(2))
            ops, got = [], []
            for i, k in enumerate(kinds):
                for w in (i + 1, 1, -1, -3):
                    got.append(cbf.add(k, w) if w > 0 else cbf.remove(k, -w))
                    ops.append((_plain(k), w))
            for w in ((1 << 32) - 3, 5, -1, 2):  # (saturation: countingbloom.py:148-152, and the frozen counters of :198)
                got.append(cbf.add("rail", w) if w > 0 else cbf.remove("rail", -w))
                ops.append(("rail", w))
            want = ref.update_ordered([k for k, _ in ops], [w for _, w in ops])
            assert got == [int(v) for v in want]
            assert cbf.export_hex() == ref.export_hex() and cbf.elements_added == ref.elements_added
            assert [cbf.check(k) for k in kinds] == [int(v) for v in ref.check_many([_plain(k) for k in kinds])]
        with pytest.raises(TypeError):
            blm.check(17)
        with pytest.raises(TypeError):
            cms.add(17)
    finally:
        N.set_option("host_poll_us", 200)
