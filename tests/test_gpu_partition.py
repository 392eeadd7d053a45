"""GPU parity of the partitioned (large-batch) path: radix-bin probes by table slice, apply in LDS.
The path is forced on for small batches here (partition_min_keys = 1) and compared bit-for-bit with
the oracle and the golden fixtures; the direct path is the cross-check."""

import numpy as np
import pytest

from _util import sha, unpackbits

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def pa():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pyprobables_amd

    return pyprobables_amd


@pytest.fixture()
def force_partition():
    from pyprobables_amd import _native as N

    names = ("partition", "partition_min_keys", "partition_max_keys", "partition_cache_bytes", "partition_two_level_slices")
    old = [N.get_option(k) for k in names]
    N.set_option("partition", 1)
    N.set_option("partition_min_keys", 1)
    yield N
    for k, v in zip(names, old):
        N.set_option(k, v)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _table(f, dtype=np.uint8):
    return np.frombuffer(bytes(f.bloom), dtype=dtype)


def test_bloom_partitioned_golden_np2(pa, golden, oracle, force_partition):
    g = golden["bloom_np2"]  # m = 958506: non power of two, partial last slice
    blm = pa.BloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"])
    blm.add_many(_dev(oracle.gen_keys16(0, g["n_keys"])))
    assert sha(bytes(blm.bloom)) == g["sha256_table"]
    assert blm._cnt_number_bits_set() == g["bits_set"]
    lo, hi = g["check_range"]
    res = blm.check_many(_dev(oracle.gen_keys16(lo, hi - lo))).cpu().numpy().astype(np.uint8)
    assert np.array_equal(res, unpackbits(g["membership_bits"], hi - lo))


@pytest.mark.parametrize("est,fpr,n", [
    (28005615, 0.01, 1_000_000),   # headline geometry: m = 2^28, k = 7, 256 slices of 128 KiB
    (5_000_000, 0.05, 300_000),    # k = 4
    (2_000_000, 0.001, 200_000),   # k = 10 (exact instantiation)
    (1_000_000, 0.1, 200_000),     # k = 3
    (1_000_000, 0.02, 200_000),    # k = 6
    (1_000_000, 0.0001, 150_000),  # k = 13 -> 16 chains in groups of four
    (1_000_000, 0.00001, 150_000), # k = 17 -> KT = 32, five groups run
    (500_000, 0.000001, 100_000),  # k = 20
    # power-of-two m (32-bit hash chains) for the same k values
    (1160981, 0.0009653916676755292, 150_000),   # k = 10, m = 2^24
    (892544, 0.00011962950342528697, 150_000),   # k = 13, m = 2^24
    (682063, 7.370214733371715e-06, 120_000),    # k = 17, m = 2^24
    (967126, 0.12447325804747715, 150_000),      # k = 3,  m = 2^22
    (967089, 0.015491121938168474, 150_000),     # k = 6,  m = 2^23
    (300_000, 0.03, 150_000),      # k = 5, m ~ 2.2 Mbit: small slices
    (1000, 0.001, 5000),           # m < 2^16: not eligible, must fall through to the direct kernels
])
def test_bloom_partitioned_vs_oracle(pa, oracle, force_partition, est, fpr, n):
    keys = oracle.gen_keys16(11, n)
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
    assert (blm.number_hashes, blm.number_bits) == oracle.bloom_params(est, fpr)[1:]
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    half = n // 2
    blm.add_many(_dev(keys[:half]))
    blm.add_many(_dev(keys[:half // 3]))  # second batch ORs into a non-empty table
    ob.add_keys(keys[:half])
    assert np.array_equal(_table(blm), ob.bloom)
    assert blm.elements_added == half + half // 3
    assert np.array_equal(blm.check_many(_dev(keys)).cpu().numpy().astype(np.uint8), ob.check_keys(keys))


def test_bloom_partitioned_layouts_vs_oracle(pa, oracle, force_partition):
    rng = np.random.default_rng(5)
    blm_args = dict(est_elements=400_000, false_positive_rate=0.01)
    # 12-byte keys (dword source), ragged byte keys, pre-hashed keys, host-staged 16-byte keys
    k12 = rng.integers(0, 256, size=(60_000, 12), dtype=np.uint8)
    blm = pa.BloomFilter(**blm_args)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(k12))
    ob.add_keys(k12)
    assert np.array_equal(_table(blm), ob.bloom)
    ragged = [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in rng.integers(0, 40, size=30_000)]
    blm.add_many(ragged)
    ob.add_varlen(ragged)
    assert np.array_equal(_table(blm), ob.bloom)
    hs = rng.integers(0, 2**63, size=(50_000, 7), dtype=np.uint64) * 2 + 1
    blm.add_alt_many(hs)
    ob.add_hashes(hs)
    assert np.array_equal(_table(blm), ob.bloom)
    k16 = oracle.gen_keys16(0, 70_000)
    blm.add_many(k16)  # host buffer -> staged
    ob.add_keys(k16)
    assert np.array_equal(_table(blm), ob.bloom)
    assert np.array_equal(blm.check_many(k16).astype(np.uint8), ob.check_keys(k16))


def test_bloom_partitioned_bucket_overflow_is_exact(pa, oracle, force_partition):
    # every key identical: all probes land in <= k slices, far beyond the bucket capacity -> spill path
    key = oracle.gen_keys16(3, 1)
    keys = np.repeat(key, 200_000, axis=0)
    keys[::1000] = oracle.gen_keys16(100, 200)  # a few distinct ones in between
    blm = pa.BloomFilter(est_elements=2_000_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(keys))
    ob.add_keys(keys)
    assert np.array_equal(_table(blm), ob.bloom)


def test_bloom_partitioned_rounds(pa, oracle, force_partition):
    force_partition.set_option("partition_max_keys", 4096)  # many partition rounds per batch
    keys = oracle.gen_keys16(0, 50_001)
    blm = pa.BloomFilter(est_elements=500_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(keys))
    ob.add_keys(keys)
    assert np.array_equal(_table(blm), ob.bloom)


def test_cache_sized_rounds_are_exact(pa, oracle, force_partition):
    """partition_cache_bytes cuts a batch into equal rounds (Infinity Cache budget): insert, lookup and weighted
    counter adds must not depend on where the cuts fall"""
    n = 2_500_000
    keys = oracle.gen_keys16(5, n)
    dk = _dev(keys)
    blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    ob.add_keys(keys[: n // 2])
    want = ob.check_keys(keys)
    for budget in (240 << 20, 24 << 20, 0):  # 1 round, ~4 insert / ~7 lookup rounds (1 M key floor), feature off
        force_partition.set_option("partition_cache_bytes", budget)
        blm.clear()
        blm.add_many(dk[: n // 2])
        assert np.array_equal(_table(blm), ob.bloom), budget
        assert np.array_equal(blm.check_many(dk).cpu().numpy().astype(np.uint8), want), budget
    w = oracle.gen_weights(0, n)
    oc = oracle.OracleCMS(2**20, 5)
    oc.add_keys(keys, w)
    for budget in (240 << 20, 16 << 20):
        force_partition.set_option("partition_cache_bytes", budget)
        cms = pa.CountMinSketch(width=2**20, depth=5)
        cms.add_many(dk, _dev(w))
        assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins), budget
        assert cms.elements_added == oc.els_added


def test_partitioned_equals_direct(pa, oracle, force_partition):
    keys = _dev(oracle.gen_keys16(0, 400_000))
    a = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    a.add_many(keys)
    force_partition.set_option("partition", 0)
    b = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    b.add_many(keys)
    assert torch.equal(a.table_tensor, b.table_tensor)
    assert a._cnt_number_bits_set() == b._cnt_number_bits_set() > 0


# ------------------------------------------------------------------ lookups through the partitioned path
def test_bloom_check_partitioned_vs_oracle(pa, oracle, force_partition):
    keys = oracle.gen_keys16(0, 300_000)
    blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(keys[:150_000]))
    ob.add_keys(keys[:150_000])
    got = blm.check_many(_dev(keys)).cpu().numpy().astype(np.uint8)
    assert np.array_equal(got, ob.check_keys(keys))
    assert got[:150_000].all() and not got[150_000:].all()
    # a dense small filter: many false positives, non power-of-two m, k = 5
    blm = pa.BloomFilter(est_elements=300_000, false_positive_rate=0.03)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(keys[:250_000])
    ob.add_keys(keys[:250_000])
    got = blm.check_many(keys).astype(np.uint8)  # host-staged
    exp = ob.check_keys(keys)
    assert np.array_equal(got, exp) and 0 < int(exp[250_000:].sum()) < 50_000
    # bitmap variant (large batches: the partitioned lookup's bytes packed into ballot words by k_pack_answer_bits) must agree, word for word
    bits, hits = blm.check_many_bits(keys)
    assert hits == int(exp.sum())
    assert np.array_equal(np.unpackbits(np.asarray(bits).view(np.uint8), bitorder="little")[: len(keys)], exp)
    odd = keys[: len(keys) - 37]                                   # a last word with 27 live bits; device batch, hits accumulate on the device
    dbits, dhits = blm.check_many_bits(_dev(odd))
    assert int(dhits.item()) == int(exp[: len(odd)].sum())
    got = np.unpackbits(dbits.cpu().numpy().view(np.uint8), bitorder="little")
    assert np.array_equal(got[: len(odd)], exp[: len(odd)]) and not got[len(odd):].any()


def test_bloom_check_partitioned_overflow_and_rounds(pa, oracle, force_partition):
    key = oracle.gen_keys16(3, 1)
    keys = np.repeat(key, 100_000, axis=0)
    keys[::7] = oracle.gen_keys16(100, len(keys[::7]))
    blm = pa.BloomFilter(est_elements=2_000_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(keys[:50_000]))
    ob.add_keys(keys[:50_000])
    assert np.array_equal(blm.check_many(_dev(keys)).cpu().numpy().astype(np.uint8), ob.check_keys(keys))
    force_partition.set_option("partition_max_keys", 8192)
    assert np.array_equal(blm.check_many(_dev(keys)).cpu().numpy().astype(np.uint8), ob.check_keys(keys))


# ------------------------------------------------------------------ counters through the partitioned path
@pytest.mark.parametrize("width,depth", [(2**20, 5), (100_003, 4), (2**16, 9)])
def test_cms_add_partitioned_vs_oracle(pa, oracle, force_partition, width, depth):
    n = 300_000
    keys = oracle.gen_keys16(0, n // 3)
    stream = keys[np.arange(n) % (n // 3)]
    w = oracle.gen_weights(0, n)
    cms = pa.CountMinSketch(width=width, depth=depth)
    oc = oracle.OracleCMS(width, depth)
    cms.add_many(_dev(stream), _dev(w))        # weighted probes (8 B)
    cms.add_many(_dev(stream[:100_000]))        # unit weights (4 B probes)
    oc.add_keys(stream, w)
    oc.add_keys(stream[:100_000])
    assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
    assert cms.elements_added == oc.els_added
    cms.remove_many(_dev(stream[:50_000]), _dev(w[:50_000]))
    cms.remove_many(stream[50_000:60_000])
    oc.remove_keys(stream[:50_000], w[:50_000])
    oc.remove_keys(stream[50_000:60_000])
    assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
    assert cms.elements_added == oc.els_added
    assert np.array_equal(cms.check_many(_dev(keys)).cpu().numpy(), oc.check_keys(keys).astype(np.int32))


def test_cms_partitioned_saturation_paths(pa, oracle, force_partition):
    keys = oracle.gen_keys16(0, 4096)
    # (a) partial sums fit 32 bits, table clamps at INT32_MAX in the fold step
    w = np.full(4096, 2**18, dtype=np.int32)
    cms = pa.CountMinSketch(width=2**16, depth=2)
    oc = oracle.OracleCMS(2**16, 2)
    for _ in range(17):  # 2^30 per batch (< 2^31: LDS partial sums are wrap-free), 2^27 per key per batch
        cms.add_many(_dev(np.repeat(keys[:8], 512, axis=0)), _dev(w))
        oc.add_keys(np.repeat(keys[:8], 512, axis=0), w)
    assert int(oc.bins.max()) == 2**31 - 1
    assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
    # (b) batch sum|w| >= 2^31: pass 2 must take its CAS fallback and still be exact
    w = np.full(4096, 2**30, dtype=np.int32)
    cms = pa.CountMinSketch(width=2**16, depth=2)
    oc = oracle.OracleCMS(2**16, 2)
    cms.add_many(_dev(keys), _dev(w))
    oc.add_keys(keys, w)
    assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
    assert cms.elements_added == oc.els_added


def test_cbf_add_partitioned_vs_oracle(pa, oracle, force_partition, golden):
    g = golden["cbf_stream"]  # the well-formed add/remove stream: adds partitioned, removes direct
    B = g["B"]
    cbf = pa.CountingBloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"])
    for bt in range(4):
        cbf.add_many(_dev(oracle.gen_keys16(bt * B, B)))
        if bt >= 1:
            cbf.remove_many(_dev(oracle.gen_keys16((bt - 1) * B, B // 2)))
    assert sha(bytes(cbf.bloom)) == g["sha256_table"]
    assert cbf.elements_added == g["elements_added"]
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    # weighted, larger table (2^22 counters), duplicates inside the batch
    n = 200_000
    keys = oracle.gen_keys16(5, n // 2)
    stream = keys[np.arange(n) % (n // 2)]
    w = oracle.gen_weights(1, n).astype(np.uint32)
    cbf = pa.CountingBloomFilter(est_elements=437_000, false_positive_rate=0.01)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    cbf.add_many(_dev(stream), w)
    oc.update_keys(stream, w.astype(np.int64))
    assert np.array_equal(np.frombuffer(bytes(cbf.bloom), dtype=np.uint32), oc.bloom)
    assert cbf.elements_added == oc.els_added


def test_partitioned_more_layouts_and_big_k(pa, oracle, force_partition):
    rng = np.random.default_rng(9)
    # 13-byte keys (byte-granular source), device resident
    k13 = rng.integers(0, 256, size=(80_000, 13), dtype=np.uint8)
    blm = pa.BloomFilter(est_elements=500_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(k13[:50_000]))
    ob.add_keys(k13[:50_000])
    assert np.array_equal(_table(blm), ob.bloom)
    assert np.array_equal(blm.check_many(_dev(k13)).cpu().numpy().astype(np.uint8), ob.check_keys(k13))
    # k = 17 and k = 27: 32 hash chains per key
    for fpr in (1e-5, 1e-8):
        blm = pa.BloomFilter(est_elements=100_000, false_positive_rate=fpr)
        assert 16 < blm.number_hashes <= 32
        keys = oracle.gen_keys16(0, 60_000)
        ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
        blm.add_many(_dev(keys[:40_000]))
        ob.add_keys(keys[:40_000])
        assert np.array_equal(_table(blm), ob.bloom)
        assert np.array_equal(blm.check_many(_dev(keys)).cpu().numpy().astype(np.uint8), ob.check_keys(keys))
    # str keys with code points > 255 (uint32 code-point layout) through the partitioned kernels
    words = [("ключ-%d-€" % i) * (1 + i % 3) for i in range(30_000)]
    blm = pa.BloomFilter(est_elements=300_000, false_positive_rate=0.01)
    blm.add_many(words[:20_000])
    hs = np.array([oracle.default_fnv_1a(w, blm.number_hashes) for w in words], dtype=np.uint64)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    ob.add_hashes(hs[:20_000])
    assert np.array_equal(_table(blm), ob.bloom)
    assert np.array_equal(blm.check_many(words).astype(np.uint8), ob.check_hashes(hs))
    # counters with the byte-granular source
    cms = pa.CountMinSketch(width=2**18, depth=6)
    oc = oracle.OracleCMS(2**18, 6)
    w = rng.integers(1, 70_000, size=80_000).astype(np.int32)  # some weights too big to ride inline -> exact spill
    cms.add_many(_dev(k13), _dev(w))
    oc.add_keys(k13, w)
    assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
    assert cms.elements_added == oc.els_added


@pytest.mark.parametrize("est,fpr", [(62_000_000, 0.12447325804747715), (45_000_000, 0.05), (28005615, 0.01)])
def test_lookup_key_ids_fit_the_probe_word(pa, oracle, force_partition, est, fpr):
    """k = 3 / 4 / 7 on tables with 2^20-bit slices: the keyed probe packs (key index in tile << 20 | bit in slice) into
    32 bits, so the tile size has to respect the slice size (1024-thread tiles of k = 3 would need 33 bits)"""
    n = 200_000
    keys = oracle.gen_keys16(4, n)
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
    assert blm.number_bits >= 2**28
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(keys[: n // 2]))
    ob.add_keys(keys[: n // 2])
    assert np.array_equal(_table(blm), ob.bloom)
    assert np.array_equal(blm.check_many(_dev(keys)).cpu().numpy().astype(np.uint8), ob.check_keys(keys))


# ------------------------------------------------------------------ split lookup (pass 1 before the table is final)
@pytest.fixture(params=[0, 3], ids=["keyed", "tile-flags"])
def split_scheme(request, force_partition):
    """the split lookups under both schemes that have a split form: keyed probes and tile flags (round 5)"""
    force_partition.set_option("bloom_lookup", request.param)
    yield request.param
    force_partition.set_option("bloom_lookup", 2)


def test_split_lookup_sees_the_table_at_finish_time(pa, oracle, force_partition, split_scheme):
    n = 600_000
    keys = oracle.gen_keys16(21, n)
    dk = _dev(keys)
    for est, fpr in ((28005615, 0.01), (3_000_000, 0.02)):  # power-of-two and general m
        blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
        ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
        blm.add_many(dk[: n // 4])
        ob.add_keys(keys[: n // 4])
        blm.check_many_begin(dk)                 # hashes + partitions; must not look at the table
        blm.add_many(dk[n // 4: n // 2])          # the table changes while the lookup is pending
        ob.add_keys(keys[n // 4: n // 2])
        got = blm.check_many_finish().cpu().numpy().astype(np.uint8)
        assert np.array_equal(got, ob.check_keys(keys))
        assert np.array_equal(_table(blm), ob.bloom)
        with pytest.raises(Exception):
            blm.check_many_finish()                # nothing pending any more


def test_split_lookup_rounds_and_small_batches(pa, oracle, force_partition, split_scheme):
    keys = oracle.gen_keys16(2, 50_000)
    dk = _dev(keys)
    blm = pa.BloomFilter(est_elements=400_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(dk[:20_000])
    ob.add_keys(keys[:20_000])
    want = ob.check_keys(keys)
    force_partition.set_option("partition_max_keys", 7000)  # 8 rounds: only the first is partitioned ahead
    blm.check_many_begin(dk)
    assert np.array_equal(blm.check_many_finish().cpu().numpy().astype(np.uint8), want)
    force_partition.set_option("partition", 0)               # not eligible: finish falls back to the direct kernel
    blm.check_many_begin(dk)
    assert np.array_equal(blm.check_many_finish().cpu().numpy().astype(np.uint8), want)
    blm.check_many_begin(keys)                                # host batch: looked up at finish time
    assert np.array_equal(np.asarray(blm.check_many_finish(), dtype=np.uint8), want)


def test_split_lookup_segment_overflow_is_redone_exactly(pa, oracle, force_partition, split_scheme):
    # all keys identical: their probes overflow <= k segments during begin; finish must re-check the round on the device
    key = oracle.gen_keys16(9, 1)
    keys = np.repeat(key, 150_000, axis=0)
    keys[::500] = oracle.gen_keys16(500, 300)
    dk = _dev(keys)
    blm = pa.BloomFilter(est_elements=2_000_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.check_many_begin(dk)
    blm.add_many(dk[1:2])      # the repeated key goes in only now
    ob.add_keys(keys[1:2])
    got = blm.check_many_finish().cpu().numpy().astype(np.uint8)
    assert np.array_equal(got, ob.check_keys(keys))
    assert got[1] == 1 and got.sum() >= 150_000 - 300


def test_split_lookup_overflow_in_later_rounds(pa, oracle, force_partition, split_scheme):
    # six rounds; the duplicate-heavy stretch sits in rounds 1 and 2 (begin scatters round 0 only: the later rounds run in full
    # at finish, their overflowing probes take the exact direct test), and the table changes while the lookup is pending
    n = 260_000
    keys = oracle.gen_keys16(77, n)
    keys[60_000:140_000] = oracle.gen_keys16(9, 1)
    dk = _dev(keys)
    blm = pa.BloomFilter(est_elements=2_000_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(dk[:30_000])
    ob.add_keys(keys[:30_000])
    force_partition.set_option("partition_max_keys", 50_000)  # 6 rounds
    blm.check_many_begin(dk)
    blm.add_many(dk[60_000:60_001])     # the repeated key and a stretch of the last rounds go in while the lookup is pending
    blm.add_many(dk[230_000:240_000])
    ob.add_keys(keys[60_000:60_001])
    ob.add_keys(keys[230_000:240_000])
    got = blm.check_many_finish().cpu().numpy().astype(np.uint8)
    assert np.array_equal(got, ob.check_keys(keys))
    assert got[60_000:140_000].all() and got[230_000:240_000].all()


# ------------------------------------------------------------------ pass 2: chunked and end-to-end segment walks
@pytest.mark.parametrize("dense_groups", [0, 1 << 30])
def test_pass2_segment_walks_agree_with_the_oracle(pa, oracle, force_partition, dense_groups):
    """option dense_walk_groups: 0 = every segment cut into its own 64-group chunks, huge = a wave walks its segments end to
    end (the default picks by the mean segment length).  Inserts, both lookup schemes and the counter kernels under either."""
    old = force_partition.get_option("dense_walk_groups")
    force_partition.set_option("dense_walk_groups", dense_groups)
    try:
        n = 700_000
        keys = oracle.gen_keys16(123, n)
        fresh = oracle.gen_keys16(90_000_000, n // 2)
        dk, df = _dev(keys), _dev(fresh)
        for est, fpr in ((28005615, 0.01), (3_000_000, 0.02), (224044920 // 4, 0.01)):  # 256 slices, general m, 512 slices
            blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
            ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
            blm.add_many(dk[: n // 2])
            blm.add_many(dk[n // 3:])
            ob.add_keys(keys)
            assert np.array_equal(_table(blm), ob.bloom)
            for scheme in (0, 1, 3):
                force_partition.set_option("bloom_lookup", scheme)
                assert np.array_equal(blm.check_many(dk).cpu().numpy().astype(np.uint8), ob.check_keys(keys))
                assert np.array_equal(blm.check_many(df).cpu().numpy().astype(np.uint8), ob.check_keys(fresh))
            force_partition.set_option("bloom_lookup", 2)
        w = (np.arange(n, dtype=np.int32) % 7) + 1
        cms = pa.CountMinSketch(width=2**20, depth=5)
        oc = oracle.OracleCMS(2**20, 5)
        cms.add_many(dk, torch.from_numpy(w).cuda())
        cms.add_many(dk[: n // 2])
        oc.add_keys(keys, w)
        oc.add_keys(keys[: n // 2])
        assert np.array_equal(cms.table_tensor.cpu().numpy()[: oc.bins.size], oc.bins)
        assert np.array_equal(cms.check_many(dk).cpu().numpy(), oc.check_keys(keys))
        cbf = pa.CountingBloomFilter(est_elements=3_000_000, false_positive_rate=0.01)
        ocb = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
        cbf.add_many(dk)
        ocb.update_keys(keys)
        assert np.array_equal(cbf.check_many(dk).cpu().numpy().astype(np.uint32), ocb.check_keys(keys))
        cbf.remove_many(dk[: n // 2])
        ocb.update_keys(keys[: n // 2], -np.ones(n // 2, dtype=np.int64))
        assert np.array_equal(_table(cbf, np.uint32), ocb.bloom)
    finally:
        force_partition.set_option("dense_walk_groups", old)
        force_partition.set_option("bloom_lookup", 2)


# ------------------------------------------------------------------ two-level path (coarse buckets, then k_part_split)
@pytest.mark.parametrize("est,fpr,n", [
    (28005615, 0.01, 700_000),    # 256 slices -> 128 coarse buckets x 2
    (3_000_000, 0.01, 400_000),   # general m, 28 slices
    (60_000_000, 0.02, 900_000),  # ~ 490 Mbit: 468 slices, last one partial
    (967126, 0.12447325804747715, 300_000),  # k = 3, m = 2^22: 4 slices -> not two-level (<= threshold), still exact
])
def test_two_level_bloom_insert_vs_oracle(pa, oracle, force_partition, est, fpr, n):
    force_partition.set_option("partition_two_level_slices", 4)
    keys = oracle.gen_keys16(31, n)
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(keys[: n // 2]))
    blm.add_many(_dev(keys[n // 3:]))       # overlapping second batch into a non-empty table
    ob.add_keys(keys)
    assert np.array_equal(_table(blm), ob.bloom)
    assert np.array_equal(blm.check_many(_dev(keys[:50_000])).cpu().numpy().astype(np.uint8), ob.check_keys(keys[:50_000]))


def test_two_level_bloom_overflow_and_layouts(pa, oracle, force_partition):
    force_partition.set_option("partition_two_level_slices", 4)
    key = oracle.gen_keys16(3, 1)
    keys = np.repeat(key, 120_000, axis=0)   # every probe in <= k slices: level-1 AND level-2 segments overflow -> spills
    keys[::700] = oracle.gen_keys16(100, 172)
    blm = pa.BloomFilter(est_elements=4_000_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(keys))
    ob.add_keys(keys)
    assert np.array_equal(_table(blm), ob.bloom)
    rng = np.random.default_rng(8)
    ragged = [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in rng.integers(0, 40, size=40_000)]
    blm.add_many(ragged)
    ob.add_varlen(ragged)
    assert np.array_equal(_table(blm), ob.bloom)


@pytest.mark.parametrize("width,depth,weighted", [(2**20, 5, True), (2**20, 5, False), (300_007, 4, True), (2**18, 7, False)])
def test_two_level_counter_adds_vs_oracle(pa, oracle, force_partition, width, depth, weighted):
    force_partition.set_option("partition_two_level_slices", 4)
    n = 500_000
    keys = oracle.gen_keys16(77, n // 2)
    keys = np.concatenate([keys, keys])      # every key twice
    w = oracle.gen_weights(0, n)
    w[::997] = 5000                          # too big for the level-1 inline field (2^11): exact spill
    cms = pa.CountMinSketch(width=width, depth=depth)
    oc = oracle.OracleCMS(width, depth)
    if weighted:
        cms.add_many(_dev(keys), _dev(w))
        oc.add_keys(keys, w)
        cms.remove_many(_dev(keys[:1000]), _dev(w[:1000]))
        oc.remove_keys(keys[:1000], w[:1000])
    else:
        cms.add_many(_dev(keys))
        oc.add_keys(keys)
    assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
    assert cms.elements_added == oc.els_added


def test_two_level_cbf_add_vs_oracle(pa, oracle, force_partition):
    force_partition.set_option("partition_two_level_slices", 4)
    n = 400_000
    keys = oracle.gen_keys16(5, n)
    w = (1 + (np.arange(n) % 4)).astype(np.uint32)
    cbf = pa.CountingBloomFilter(est_elements=2_000_000, false_positive_rate=0.01)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    cbf.add_many(_dev(keys), w)
    oc.update_keys(keys, w.astype(np.int64))
    cbf.add_many(_dev(keys[:100_000]))
    oc.update_keys(keys[:100_000], np.ones(100_000, dtype=np.int64))
    assert np.array_equal(np.frombuffer(bytes(cbf.bloom), dtype=np.uint32), oc.bloom)
    assert cbf.elements_added == oc.els_added


# ------------------------------------------------------------------ write-combined CBF updates
def test_cbf_combined_updates_vs_oracle(pa, oracle, force_partition):
    """psk_cbf_update_combined: batches wait on the device and are applied list by list (adds, then removes as plain
    decrements); small lists here so that several flushes, the weighted / unit switch, host and device batches, a key
    length change and the read-triggered flush are all exercised"""
    from pyprobables_amd import _native as N

    old = N.get_option("combine_keys")
    N.set_option("combine_keys", 150_000)
    try:
        B = 40_000
        cbf = pa.CountingBloomFilter(est_elements=2_000_000, false_positive_rate=0.01, combine_updates=True)
        oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
        for b in range(12):
            keys = oracle.gen_keys16(b * B, B)
            if b % 3 == 2:      # weighted batch (after unit ones: the earlier entries get their 1s)
                w = oracle.gen_weights(b * B, B).astype(np.uint32)
                cbf.add_many(_dev(keys), w)
                oc.update_keys(keys, w.astype(np.int64))
            elif b % 3 == 1:    # host batch
                cbf.add_many(keys)
                oc.update_keys(keys)
            else:
                cbf.add_many(_dev(keys))
                oc.update_keys(keys)
            if b >= 1 and b % 3 != 0:   # unit removes of keys that were added with weight >= 1
                prev = oracle.gen_keys16((b - 1) * B, B // 2)
                cbf.remove_many(_dev(prev))
                oc.update_keys(prev, -np.ones(B // 2, dtype=np.int64))
            if b in (4, 9):     # a read in the middle sees everything handed over so far
                probe = oracle.gen_keys16(b * B, 1000)
                assert np.array_equal(np.asarray(cbf.check_many(probe)).view(np.uint32), oc.check_keys(probe))
        assert np.array_equal(np.frombuffer(bytes(cbf.bloom), dtype=np.uint32), oc.bloom)
        assert cbf.elements_added == oc.els_added
        assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
        # other key lengths flush what waits and start a new list
        k8 = np.ascontiguousarray(oracle.gen_keys16(7, 30_000)[:, :8])
        cbf.add_many(_dev(k8))
        cbf.add_many(_dev(oracle.gen_keys16(900_000, 10)))
        oc.update_keys(k8)
        oc.update_keys(oracle.gen_keys16(900_000, 10))
        assert torch.equal(cbf.table_tensor[: cbf.number_bits].cpu(), torch.from_numpy(oc.bloom.view(np.int32)))
        # a remove of an absent key is a contract violation in this mode (tallied), not a no-op
        cbf.remove_many(_dev(oracle.gen_keys16(5_000_000, 100)))
        assert cbf.batch_diagnostics()["violations"] > 0
        # clear drops what waits
        cbf.clear()
        cbf.add_many(_dev(oracle.gen_keys16(0, 1000)))
        cbf.clear()
        assert cbf.elements_added == 0 and cbf._cnt_number_bits_set() == 0
    finally:
        N.set_option("combine_keys", old)


def test_cbf_unchecked_remove_direct_and_partitioned_agree(pa, oracle, force_partition):
    """the decrement kernels behind the combined path: direct (CbfSub) and partitioned (k_counter_apply fold) forms, incl. a
    frozen counter (2^32-1 stays) and weights too large for the inline probe (exact spill)"""
    from pyprobables_amd import _native as N

    n = 120_000
    keys = oracle.gen_keys16(3, n)
    w = (1 + (np.arange(n) % 5)).astype(np.uint32)
    w[::1000] = 70_000
    results = []
    for part in (1, 0):
        N.set_option("partition", part)
        cbf = pa.CountingBloomFilter(est_elements=900_000, false_positive_rate=0.01, combine_updates=True)
        cbf.add_many(_dev(keys), w)
        cbf.add_many(_dev(keys[:10]), np.full(10, 2**32 - 1, dtype=np.uint32))   # saturate a few counters: they freeze
        cbf.remove_many(_dev(keys[: n // 2]), w[: n // 2])
        results.append((cbf.table_tensor.clone(), cbf.elements_added, cbf.batch_diagnostics()["violations"]))
    N.set_option("partition", 1)
    assert torch.equal(results[0][0], results[1][0]) and results[0][1:] == results[1][1:]
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    oc.update_keys(keys, w.astype(np.int64))
    oc.update_keys(keys[:10], np.full(10, 2**32 - 1, dtype=np.int64))
    oc.update_keys(keys[: n // 2], -w[: n // 2].astype(np.int64))
    assert np.array_equal(results[0][0].cpu().numpy().view(np.uint32)[: oc.m], oc.bloom)


# ------------------------------------------------------------------ Bloom lookups: return trip vs keyed probes
@pytest.mark.parametrize("mode", [1, 0, 3, 4])
def test_bloom_lookup_modes_vs_oracle(pa, oracle, force_partition, mode):
    """option bloom_lookup: 1 = pass 1 with perm / runinfo + k_bloom_gather + k_bloom_collect, 0 = keyed probes + k_bloom_test, 3 = the
    insert's compact probes + a flag per tile that met a clear bit + k_bloom_flag_resolve (round 5), 4 = lazy gathers (one key per lane, the next
    probe only while every earlier bit was set: k_bloom_check_lazy); all against the oracle over hits, misses,
    k classes, table sizes, layouts, rounds and segment overflow"""
    force_partition.set_option("bloom_lookup", mode)
    try:
        n = 160_000
        keys = oracle.gen_keys16(21, n)
        for est, fpr in [(28005615, 0.01), (5_000_000, 0.05), (2_000_000, 0.001), (1_000_000, 0.1), (300_000, 0.03), (1_000_000, 0.00001),
                         (1160981, 0.0009653916676755292), (224044920, 0.01)]:
            blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
            ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
            blm.add_many(_dev(keys[: n // 2]))
            ob.add_keys(keys[: n // 2])
            want = ob.check_keys(keys)
            assert np.array_equal(blm.check_many(_dev(keys)).cpu().numpy().astype(np.uint8), want), (est, fpr)
            assert np.array_equal(np.asarray(blm.check_many(keys[50_000:90_000])).astype(np.uint8), want[50_000:90_000])
        # several rounds, a partial last tile
        blm = pa.BloomFilter(est_elements=3_000_000, false_positive_rate=0.01)
        ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
        blm.add_many(_dev(keys[::3]))
        ob.add_keys(keys[::3])
        for max_keys in (1 << 25, 50_000, 4096 * 7 + 1):
            force_partition.set_option("partition_max_keys", max_keys)
            assert np.array_equal(blm.check_many(_dev(keys[:-3])).cpu().numpy().astype(np.uint8), ob.check_keys(keys[:-3]))
        force_partition.set_option("partition_max_keys", 1 << 25)
        # duplicate-heavy batch: segments overflow, the flagged redo keeps the answers exact
        same = np.repeat(keys[5:6], 150_000, axis=0)
        mix = np.concatenate([same, keys[:50_000]])
        assert np.array_equal(blm.check_many(_dev(mix)).cpu().numpy().astype(np.uint8), ob.check_keys(mix))
        fresh = oracle.gen_keys16(9_000_000, 120_000)   # (almost) all misses
        assert np.array_equal(blm.check_many(_dev(fresh)).cpu().numpy().astype(np.uint8), ob.check_keys(fresh))
        # ragged / wide-character keys and pre-hashed batches
        words = [("ключ-%d-€" % i) * (1 + i % 3) for i in range(30_000)]
        blm.add_many(words[:20_000])
        hs = np.array([oracle.default_fnv_1a(w, blm.number_hashes) for w in words], dtype=np.uint64)
        ob.add_hashes(hs[:20_000])
        assert np.array_equal(np.asarray(blm.check_many(words)).astype(np.uint8), ob.check_hashes(hs))
        assert np.array_equal(np.asarray(blm.check_alt_many(hs)).astype(np.uint8), ob.check_hashes(hs))
        if mode == 3:
            # what the scheme is for: every key present (no tile flagged), then ONE absent key in the middle (one tile re-checked), several
            # rounds of 16 tiles per workgroup, and the flags left clean for the next call
            blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
            ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
            blm.add_many(_dev(keys))
            ob.add_keys(keys)
            assert blm.check_many(_dev(keys)).cpu().numpy().all()
            one = keys.copy()
            one[77_777] = fresh[0]
            want = ob.check_keys(one)
            assert not want[77_777]
            assert np.array_equal(blm.check_many(_dev(one)).cpu().numpy().astype(np.uint8), want)
            assert blm.check_many(_dev(keys)).cpu().numpy().all()
            force_partition.set_option("partition_max_keys", 40_000)
            assert np.array_equal(blm.check_many(_dev(one)).cpu().numpy().astype(np.uint8), want)
            force_partition.set_option("partition_max_keys", 1 << 25)
            assert np.array_equal(blm.check_many(_dev(one[:-5])).cpu().numpy().astype(np.uint8), want[:-5])
    finally:
        force_partition.set_option("bloom_lookup", 2)


def test_bloom_lookup_auto_mode_follows_the_miss_rate_and_stays_exact(pa, oracle, force_partition):
    """default option bloom_lookup = 2: the scheme of the next large lookup follows the tally of the previous ones (pinned page,
    no synchronisation) -- whatever it picks, every answer equals the oracle's across alternating all-hit / all-miss batches"""
    assert force_partition.get_option("partition") == 1
    n = 300_000
    keys = oracle.gen_keys16(0, n)
    fresh = oracle.gen_keys16(50_000_000, n)
    blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(_dev(keys))
    ob.add_keys(keys)
    want_hit, want_miss = ob.check_keys(keys), ob.check_keys(fresh)
    mixed = np.concatenate([keys[: n // 2], fresh[: n // 2]])
    want_mixed = ob.check_keys(mixed)
    dk, df, dm = _dev(keys), _dev(fresh), _dev(mixed)
    for batch, want in [(dk, want_hit), (dk, want_hit), (df, want_miss), (df, want_miss), (df, want_miss), (dm, want_mixed), (dk, want_hit),
                        (dk, want_hit), (dk, want_hit), (df, want_miss), (dm, want_mixed),
                        # a run of all-miss batches: return trip, then lazy gathers (scheme 4: enough of them for it to be entered and kept),
                        # a mixed batch through it (too many gathers per key: back to the return trip) and misses again (held off for a while)
                        (df, want_miss), (df, want_miss), (df, want_miss), (df, want_miss), (df, want_miss), (dm, want_mixed), (dm, want_mixed),
                        (df, want_miss), (df, want_miss), (dk, want_hit)]:
        got = blm.check_many(batch)
        torch.cuda.synchronize()   # the tally of this call is on the pinned page before the next call chooses
        assert np.array_equal(got.cpu().numpy().astype(np.uint8), want)
    # and without synchronisation in between (the choice may lag by a call)
    outs = [blm.check_many(b) for b in (df, dk, df, dk, dm)]
    for got, want in zip(outs, (want_miss, want_hit, want_miss, want_hit, want_mixed)):
        assert np.array_equal(got.cpu().numpy().astype(np.uint8), want)


def test_cms_weighted_adds_compact_probe_format(pa, oracle, force_partition):
    """weighted CountMinSketch adds whose weights fit four bits travel as 20-bit fields (PayWeightSmall, countminsketch.py:267-288): forced on,
    forced off and chosen by the hint of the previous batches -- every bin, elements_added and every answer equal the oracle's, whatever the
    weights turn out to be (0, 15, 16, large, negative: the ones outside 0 .. 15 reach the table directly)"""
    N = force_partition
    old = N.get_option("cms_small_weights")
    rng = np.random.default_rng(31)
    n = 400_000
    keys = oracle.gen_keys16(77_000, n)
    dk = _dev(keys)
    try:
        for width, depth in ((2**20, 5), (1_000_003, 4), (2**16, 3)):
            for mode in (2, 0, 1):
                N.set_option("cms_small_weights", mode)
                cms = pa.CountMinSketch(width=width, depth=depth)
                oc = oracle.OracleCMS(width, depth)
                batches = [
                    rng.integers(1, 8, size=n),                                   # cfg 3's weights
                    rng.integers(0, 16, size=n),                                  # the whole small range, zeros included
                    np.where(rng.random(n) < 0.001, 16 + rng.integers(0, 5000, size=n), rng.integers(1, 16, size=n)),   # a few big ones
                    rng.integers(1, 8, size=n),
                    np.where(rng.random(n) < 0.5, -rng.integers(1, 4, size=n), rng.integers(1, 16, size=n)),            # negative weights
                    rng.integers(100, 100_000, size=n),                           # nothing small at all
                    rng.integers(1, 8, size=n),
                ]
                for i, w in enumerate(batches):
                    w32 = w.astype(np.int32)
                    cms.add_many(dk, _dev(w32))
                    oc.add_keys(keys, w32)
                    if i in (1, 4, 6):
                        assert np.array_equal(cms.table_tensor.cpu().numpy()[: oc.bins.size], oc.bins), (width, depth, mode, i)
                        assert cms.elements_added == oc.els_added
                assert np.array_equal(cms.check_many(dk[:50_000]).cpu().numpy().astype(np.int64), oc.check_keys(keys[:50_000]))
    finally:
        N.set_option("cms_small_weights", old)


def test_cms_compact_format_follows_the_hint_and_backs_off(pa, oracle, force_partition):
    """option cms_small_weights = 1 (default): the first weighted batch of a sketch travels in the wide format; once pass 1 has reported a
    batch without any weight outside 0 .. 15 the compact one is taken; ONE such weight keeps the wide format for the next 64 weighted
    batches (psk_sketch::wt) -- and every bin equals the oracle's throughout"""
    N = force_partition
    assert N.get_option("cms_small_weights") == 1
    used = lambda: N.get_option("cms_small_weights_used")
    n = 200_000
    keys = oracle.gen_keys16(4242, n)
    dk = _dev(keys)
    small = _dev(np.full(n, 3, dtype=np.int32))
    big = np.full(n, 3, dtype=np.int32)
    big[7] = 1000
    dbig = _dev(big)
    cms = pa.CountMinSketch(width=2**18, depth=5)
    oc = oracle.OracleCMS(2**18, 5)

    def add(w_dev, w_host, expect_compact):
        u0 = used()
        cms.add_many(dk, w_dev)
        torch.cuda.synchronize()               # the hint of this batch is on the pinned page before the next call reads it
        oc.add_keys(keys, w_host)
        assert (used() - u0 == 1) == expect_compact, (used() - u0, expect_compact)

    w3 = np.full(n, 3, dtype=np.int32)
    add(small, w3, False)                      # nothing known yet: wide
    add(small, w3, True)
    add(small, w3, True)
    add(dbig, big, True)                       # (the big weight itself goes to the table directly: exact, and it is reported)
    for _ in range(3):
        add(small, w3, False)                  # backing off
    assert np.array_equal(cms.table_tensor.cpu().numpy()[: oc.bins.size], oc.bins) and cms.elements_added == oc.els_added
    for _ in range(61):
        add(small, w3, False)
    add(small, w3, True)                       # 64 weighted batches after the report: compact again
    assert np.array_equal(cms.table_tensor.cpu().numpy()[: oc.bins.size], oc.bins) and cms.elements_added == oc.els_added
