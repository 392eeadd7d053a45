"""Partitioned LOOKUPS of the counter structures (psk_lookup.hpp: pass 1 with perm / runinfo, k_counter_gather,
k_lookup_collect) against the oracle, forced on for small batches: CountMinSketch.check under all three queries
(countminsketch.py:332-340, 429-453) and CountingBloomFilter.check (countingbloom.py:166-174)."""

import numpy as np
import pytest

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


def _direct(N, fn):
    """the same call through the direct kernels (the cross-check)"""
    N.set_option("partition", 0)
    try:
        return fn()
    finally:
        N.set_option("partition", 1)


@pytest.mark.parametrize("width,depth", [(2**20, 5), (100_003, 4), (2**16, 9), (2**17, 1), (70_001, 13), (2**18, 20), (2**16 + 1, 7), (2**22, 3)])
def test_cms_check_all_queries_vs_oracle(pa, oracle, force_partition, width, depth):
    n = 240_000
    keys = oracle.gen_keys16(3, n // 2)
    stream = keys[np.arange(n) % (n // 2)]
    w = oracle.gen_weights(0, n)
    cms = pa.CountMinSketch(width=width, depth=depth)
    cms.add_many(_dev(stream), _dev(w))
    cms.remove_many(_dev(stream[:30_000]), _dev(w[:30_000]) * 3)   # some bins go negative: signed min / floor-division mean
    probe = oracle.gen_keys16(3, n)                                # half of them never added
    dprobe = _dev(probe)
    for query in ("min", "mean", "mean-min"):
        oc = oracle.OracleCMS(width, depth, query)
        oc.add_keys(stream, w)
        oc.remove_keys(stream[:30_000], w[:30_000] * 3)
        assert cms.elements_added == oc.els_added
        cms.query_type = query
        want = oc.check_keys(probe)
        got = cms.check_many(dprobe).cpu().numpy().astype(np.int64)
        assert np.array_equal(got, want), query
        assert np.array_equal(_direct(force_partition, lambda: cms.check_many(dprobe).cpu().numpy().astype(np.int64)), want)
        # host buffers take the same path (staged)
        assert np.array_equal(np.asarray(cms.check_many(probe[:70_000])).astype(np.int64), want[:70_000])


def test_cms_check_rounds_tiles_and_tail(pa, oracle, force_partition):
    """several rounds (partition_max_keys), a last partial tile, and a key count that is not a multiple of anything"""
    n = 333_337
    keys = oracle.gen_keys16(0, n)
    cms = pa.CountMinSketch(width=2**19, depth=5)
    oc = oracle.OracleCMS(2**19, 5)
    w = oracle.gen_weights(5, n)
    cms.add_many(_dev(keys), _dev(w))
    oc.add_keys(keys, w)
    want = oc.check_keys(keys).astype(np.int32)
    for max_keys in (1 << 25, 100_000, 4096 * 13):
        force_partition.set_option("partition_max_keys", max_keys)
        assert np.array_equal(cms.check_many(_dev(keys)).cpu().numpy(), want)
    force_partition.set_option("partition_cache_bytes", 8 << 20)   # cache-sized rounds
    force_partition.set_option("partition_max_keys", 1 << 25)
    assert np.array_equal(cms.check_many(_dev(keys)).cpu().numpy(), want)


def test_cms_check_segment_overflow_is_redone_exactly(pa, oracle, force_partition):
    """200 k identical keys land in depth slices only: their segments overflow, the device flag triggers the direct redo"""
    n = 200_000
    keys = oracle.gen_keys16(0, n)
    same = np.repeat(keys[7:8], n, axis=0)
    mix = np.concatenate([same[: n // 2], keys[: n // 2]])
    cms = pa.CountMinSketch(width=2**20, depth=5)
    oc = oracle.OracleCMS(2**20, 5)
    cms.add_many(_dev(keys))
    oc.add_keys(keys)
    for batch in (same, mix):
        assert np.array_equal(cms.check_many(_dev(batch)).cpu().numpy(), oc.check_keys(batch).astype(np.int32))
    # the flag does not stick: a normal batch afterwards is partitioned again and exact
    assert np.array_equal(cms.check_many(_dev(keys)).cpu().numpy(), oc.check_keys(keys).astype(np.int32))


def test_cms_check_layouts(pa, oracle, force_partition):
    rng = np.random.default_rng(4)
    k13 = rng.integers(0, 256, size=(90_000, 13), dtype=np.uint8)    # byte-granular source
    k24 = rng.integers(0, 256, size=(90_000, 24), dtype=np.uint8)    # dword source
    h13 = np.array([oracle.default_fnv_1a(bytes(k), 6) for k in k13[:20_000]], dtype=np.uint64)
    h24 = np.array([oracle.default_fnv_1a(bytes(k), 6) for k in k24[:20_000]], dtype=np.uint64)
    cms = pa.CountMinSketch(width=2**18, depth=6)
    cms.add_many(_dev(k13[:20_000]))
    cms.add_many(_dev(k24[:20_000]))
    ref = pa.CountMinSketch(width=2**18, depth=6)                     # the same table through the pre-hashed layout
    ref.add_alt_many(h13)
    ref.add_alt_many(h24)
    assert torch.equal(ref.table_tensor, cms.table_tensor)
    want13 = _direct(force_partition, lambda: cms.check_many(_dev(k13)).cpu().numpy())
    assert int(want13[:20_000].min()) >= 1 and int((want13[20_000:] == 0).sum()) > 0
    assert np.array_equal(cms.check_many(_dev(k13)).cpu().numpy(), want13)            # 13-byte keys, partitioned == direct
    assert np.array_equal(cms.check_many(_dev(k24)).cpu().numpy(), _direct(force_partition, lambda: cms.check_many(_dev(k24)).cpu().numpy()))
    assert np.array_equal(np.asarray(cms.check_alt_many(h13)), want13[:20_000])       # pre-hashed batch (host)
    assert np.array_equal(cms.check_alt_many(_dev(h13.view(np.int64))).cpu().numpy(), want13[:20_000])
    words = [("ключ-%d-€" % i) * (1 + i % 3) for i in range(40_000)]                  # code points > 255, ragged
    cms.add_many(words[:25_000])
    got = np.asarray(cms.check_many(words))
    assert np.array_equal(got, _direct(force_partition, lambda: np.asarray(cms.check_many(words))))
    assert int(got[:25_000].min()) >= 1


@pytest.mark.parametrize("est,fpr", [(437_000, 0.01), (2_000_000, 0.001), (300_000, 0.1), (150_000, 0.0001), (3_500_000, 0.02)])
def test_cbf_check_vs_oracle(pa, oracle, force_partition, est, fpr):
    n = 200_000
    keys = oracle.gen_keys16(9, n)
    w = oracle.gen_weights(2, n).astype(np.uint32)
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=fpr)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    cbf.add_many(_dev(keys[: n // 2]), w[: n // 2])
    oc.update_keys(keys[: n // 2], w[: n // 2].astype(np.int64))
    cbf.remove_many(_dev(keys[: n // 8]))
    oc.update_keys(keys[: n // 8], -np.ones(n // 8, dtype=np.int64))
    want = oc.check_keys(keys)
    got = cbf.check_many(_dev(keys)).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
    assert np.array_equal(np.asarray(cbf.check_many(keys[:50_000])).view(np.uint32), want[:50_000])
    assert int((want == 0).sum()) > 0 and int(want.max()) >= 7


def test_cbf_check_alt_takes_the_min_over_all_supplied_hashes(pa, oracle, force_partition):
    """countingbloom.py:174: check_alt looks at EVERY supplied hash, not only the first k"""
    cbf = pa.CountingBloomFilter(est_elements=437_000, false_positive_rate=0.01)
    keys = oracle.gen_keys16(0, 50_000)
    cbf.add_many(_dev(keys))
    h = np.array([oracle.default_fnv_1a(bytes(k), 11) for k in keys[:20_000]], dtype=np.uint64)  # 11 > k = 7
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    oc.update_keys(keys)
    want = np.array([oc.check_alt(row) for row in h[:3000]], dtype=np.uint32)
    got = np.asarray(cbf.check_alt_many(h)).view(np.uint32)
    assert np.array_equal(got[:3000], want)
    assert np.array_equal(got, _direct(force_partition, lambda: np.asarray(cbf.check_alt_many(h)).view(np.uint32)))


def test_lookup_scratch_release_and_regrow(pa, oracle, force_partition):
    cms = pa.CountMinSketch(width=2**20, depth=5)
    keys = oracle.gen_keys16(0, 150_000)
    cms.add_many(_dev(keys))
    a = cms.check_many(_dev(keys)).cpu().numpy()
    cms._tab.release_scratch()
    assert np.array_equal(cms.check_many(_dev(keys)).cpu().numpy(), a)


def test_lookup_value_formats_and_shared_slices(pa, oracle, force_partition):
    """pass 2 writes a slice's values as uint16 when every counter of the slice is below 2^16 and as uint32 otherwise; both
    formats in one table (big weights on a few keys, negative CMS bins), with and without two workgroups per slice"""
    n = 200_000
    keys = oracle.gen_keys16(77, n)
    w = np.ones(n, dtype=np.uint32)
    w[::997] = 3_000_000_000 // 7          # a few huge counters: their slices go wide
    w[5::1013] = 70_000
    cbf = pa.CountingBloomFilter(est_elements=437_000, false_positive_rate=0.01)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    cbf.add_many(_dev(keys), w)
    oc.update_keys(keys, w.astype(np.int64))
    want = oc.check_keys(keys)
    assert int(want.max()) > 2**16 and int((want < 2**16).sum()) > n // 2
    cms = pa.CountMinSketch(width=100_003, depth=3)      # 10 slices of 2^15 (not a multiple of the CU count: shared slices)
    ocm = oracle.OracleCMS(100_003, 3)
    wi = w.astype(np.int64).clip(max=2**31 - 1).astype(np.int32)
    cms.add_many(_dev(keys), _dev(wi))
    ocm.add_keys(keys, wi)
    cms.remove_many(_dev(keys[:5000]), _dev(np.full(5000, 100_000, dtype=np.int32)))   # negative bins
    ocm.remove_keys(keys[:5000], np.full(5000, 100_000, dtype=np.int32))
    wantc = ocm.check_keys(keys).astype(np.int32)
    assert int(wantc.min()) < 0
    assert np.array_equal(cbf.check_many(_dev(keys)).cpu().numpy().view(np.uint32), want)
    assert np.array_equal(cms.check_many(_dev(keys)).cpu().numpy(), wantc)


# ------------------------------------------------------------------ 2^26 .. 2^27 counters: slices of 2^16 counters held as 16-bit values
def test_cbf_lookups_between_2p26_and_2p27_counters_take_half_slices(pa, oracle, force_partition):
    """a CountingBloomFilter for 10 M elements at 1 % has 9.6e7 counters = 2925 slices of 2^15: beyond the one-level lookup, which
    used to mean the direct gathers.  k_counter_gather<true> keeps 2^16 counters per slice as 16-bit values; check / remove
    (validated: lookup + decrement) against the oracle, the direct kernels, and with the option off"""
    n = 400_000
    keys = oracle.gen_keys16(11, n)
    probe = np.concatenate([keys[: n // 2], oracle.gen_keys16(99_000_000, n // 2)])
    dk, dp = _dev(keys), _dev(probe)
    cbf = pa.CountingBloomFilter(est_elements=10_000_000, false_positive_rate=0.01)
    assert 2**26 < cbf.number_bits <= 2**27
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    w = (np.arange(n, dtype=np.int64) % 5) + 1
    cbf.add_many(dk, w.astype(np.uint32))
    oc.update_keys(keys, w)
    want = oc.check_keys(probe)
    got = cbf.check_many(dp).cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, want)
    assert np.array_equal(_direct(force_partition, lambda: cbf.check_many(dp).cpu().numpy().astype(np.uint32)), want)
    force_partition.set_option("lookup_half_slices", 0)
    try:
        assert np.array_equal(cbf.check_many(dp).cpu().numpy().astype(np.uint32), want)
    finally:
        force_partition.set_option("lookup_half_slices", 1)
    # validated remove = the same lookup + the partitioned decrement
    cbf.remove_many(dk[: n // 4])
    oc.update_keys(keys[: n // 4], -np.ones(n // 4, dtype=np.int64))
    assert np.array_equal(np.frombuffer(bytes(cbf.bloom), dtype=np.uint32), oc.bloom)
    assert cbf.elements_added == oc.els_added
    # a counter at or above 2^16 cannot live in a 16-bit slice image: the device flag sends the batch through the direct kernel
    big = oracle.gen_keys16(5_000_000, 1)
    cbf.add_many(_dev(big), 70_000)
    oc.update_keys(big, np.array([70_000], dtype=np.int64))
    probe2 = np.concatenate([big, probe])
    assert np.array_equal(cbf.check_many(_dev(probe2)).cpu().numpy().astype(np.uint32), oc.check_keys(probe2))


def test_cms_2p27_bins_half_slices_and_negative_bins(pa, oracle, force_partition):
    """CountMinSketch 2^24 x 8 = 2^27 bins (2048 slices of 2^16): min / mean lookups; a negative bin (remove) has bits above 2^16
    as a uint32, so its slice raises the redo flag and the answers still equal the oracle's"""
    n = 300_000
    keys = oracle.gen_keys16(21, n)
    w = oracle.gen_weights(3, n)
    cms = pa.CountMinSketch(width=2**24, depth=8)
    oc = oracle.OracleCMS(2**24, 8)
    cms.add_many(_dev(keys), _dev(w))
    oc.add_keys(keys, w)
    probe = np.concatenate([keys[: n // 2], oracle.gen_keys16(77_000_000, n // 2)])
    for query in ("min", "mean"):
        cms.query_type = query
        oq = oracle.OracleCMS(2**24, 8, query)
        oq.add_keys(keys, w)
        assert np.array_equal(cms.check_many(_dev(probe)).cpu().numpy().astype(np.int64), oq.check_keys(probe)), query
    cms.query_type = "min"
    cms.remove_many(_dev(keys[:1000]), _dev(w[:1000]) * 2)   # those bins go negative
    oc.remove_keys(keys[:1000], w[:1000] * 2)
    assert np.array_equal(cms.check_many(_dev(probe)).cpu().numpy().astype(np.int64), oc.check_keys(probe))


def test_wide_device_weights_are_range_checked(pa, oracle):
    """a 64-bit device tensor of counts is narrowed to the engine's 32-bit weights: values outside the range raise like the host
    path does (they used to wrap silently); in-range 64-bit counts are accepted"""
    keys = oracle.gen_keys16(0, 1000)
    dk = _dev(keys)
    cms = pa.CountMinSketch(width=4096, depth=4)
    with pytest.raises(OverflowError):
        cms.add_many(dk, torch.full((1000,), 2**40, dtype=torch.int64, device="cuda"))
    with pytest.raises(TypeError):
        cms.add_many(dk, torch.ones(1000, dtype=torch.float32, device="cuda"))
    cms.add_many(dk, torch.full((1000,), 3, dtype=torch.int64, device="cuda"))
    oc = oracle.OracleCMS(4096, 4)
    oc.add_keys(keys, np.full(1000, 3, dtype=np.int32))
    assert np.array_equal(cms.table_tensor.cpu().numpy()[: oc.bins.size], oc.bins)
    assert cms.elements_added == 3000


# ------------------------------------------------------------------ return-trip lookups into tables of ~900 slices and more: 4096-key pass-1 tiles
@pytest.mark.parametrize("est", [112_000_000, 224_044_920])
def test_bloom_return_trip_with_4096_key_tiles(pa, oracle, force_partition, est):
    """m = 2^30 / 2^31 bits (1024 / 2048 slices): pass 1 of the return trip runs 4096-key tiles there (PayBloomLookup::fat1024 -- cut to what the
    LDS stage holds at 2048 slices), perm[] positions reach past 2^15, pass 3 prefetches four perm records per thread.  A batch that ends inside a
    tile, keys that repeat, half of the keys absent.  bloom.py:261-272."""
    n = 2_600_003
    keys = oracle.gen_keys16(61, n)
    keys[5000:5200] = keys[4]
    probe = np.concatenate([keys[: n // 2], oracle.gen_keys16(3_000_000_000, n // 2 + 1)])
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=0.01)
    blm.add_many(_dev(keys))
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    ob.add_keys(keys)
    blm.set_engine_option("bloom_lookup", 1)
    assert np.array_equal(blm.check_many(_dev(probe)).cpu().numpy().astype(np.uint8), ob.check_keys(probe).astype(np.uint8))
    assert bool(blm.check_many(_dev(keys)).all())
