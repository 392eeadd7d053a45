"""Pass 1 through fixed-capacity LDS bins (psk_part_bins.hpp, round 6) against the counting-sort pass 1 it replaces and against the oracle:
same segments, same groups, same tables, same answers -- bit for bit -- for every geometry it is eligible for, and where its exact fallbacks
fire: a bin that fills up (hundreds of copies of ONE key in a tile), a segment that fills up, tiles with a short tail, the tile-flag lookups
with their flagged tiles.  bloom.py:241-272, hashes.py:71-103."""

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
def N():
    from pyprobables_amd import _native as N

    names = ("partition", "partition_min_keys", "partition_max_keys", "pass1_bins", "bloom_lookup")
    old = [N.get_option(k) for k in names]
    N.set_option("partition", 1)
    N.set_option("partition_min_keys", 1)
    yield N
    for k, v in zip(names, old):
        N.set_option(k, v)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _table(f):
    return np.frombuffer(bytes(f.bloom), dtype=np.uint8)


@pytest.mark.parametrize("est,fpr,n", [
    (28005615, 0.01, 2_500_003),    # the headline geometry: m = 2^28, k = 7, 256 slices (32-bit chains, bin = a bit field of the hash)
    (28005615, 0.05, 1_000_000),    # k = 4
    (10_000_000, 0.01, 1_200_000),  # m ~ 9.6e7: not a power of two (64-bit chains, Barrett), the table ends inside the last slice
    (6_000_000, 0.002, 700_001),    # k = 9 -> not eligible (k > 8): the same kernel either way
    (40_000_000, 0.03, 900_000),    # k = 5, 2^28 < m: 512 slices do not fit the bins twice per CU -> the counting sort either way
    (1_500_000, 0.01, 400_000),     # a small table: few slices, several lanes per slice in the write-out
])
def test_bins_and_counting_sort_build_the_same_filter(pa, oracle, N, est, fpr, n):
    keys = oracle.gen_keys16(3, n)
    probe = oracle.gen_keys16(3 + n // 2, n)   # half present
    tabs, answers = [], []
    for bins in (1, 0):
        N.set_option("pass1_bins", bins)
        blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
        blm.add_many(_dev(keys[: n // 3]))
        blm.add_many(_dev(keys[n // 3:]))       # a second batch: other tile sizes, cursors start again
        tabs.append(_table(blm).copy())
        got = []
        for scheme in (3, 0):                   # tile flags (pass 1 = the insert's), keyed probes
            blm.set_engine_option("bloom_lookup", scheme)
            got.append(blm.check_many(_dev(probe)).cpu().numpy().astype(np.uint8))
            assert bool(blm.check_many(_dev(keys)).all())
        answers.append(got)
        del blm
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    ob.add_keys(keys)
    want = ob.check_keys(probe).astype(np.uint8)
    assert np.array_equal(tabs[0], ob.bloom) and np.array_equal(tabs[1], ob.bloom)
    for got in answers:
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


def test_a_bin_that_fills_up_takes_the_exact_fallback(pa, oracle, N):
    """400 copies of one key inside one tile put 400 probes into each of its k bins (capacity ~66): everything past the capacity goes to the
    table directly (inserts: atomicOr; tile-flag lookups: the probe is tested at once and flags its tile)"""
    n = 300_000
    keys = oracle.gen_keys16(77, n)
    keys[1000:1400] = keys[5]
    keys[200_000:200_300] = keys[7]
    fresh = oracle.gen_keys16(900_000_000, n)
    fresh[5000:5400] = fresh[3]                     # an absent key, 400 times in one tile
    N.set_option("pass1_bins", 1)
    blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    blm.add_many(_dev(keys))
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    ob.add_keys(keys)
    assert np.array_equal(_table(blm), ob.bloom)
    blm.set_engine_option("bloom_lookup", 3)
    for probe in (keys, fresh, np.concatenate([keys[:150_000], fresh[:150_000]])):
        assert np.array_equal(blm.check_many(_dev(probe)).cpu().numpy().astype(np.uint8), ob.check_keys(probe).astype(np.uint8))


def test_a_segment_that_fills_up_takes_the_exact_fallback(pa, oracle, N):
    """every key the same: all probes of the batch land in k of the 256 slices -- their (slice, workgroup) segments overflow and the groups
    that do not fit are applied probe by probe"""
    n = 200_000
    keys = np.repeat(oracle.gen_keys16(11, 1), n, axis=0)
    keys[::1000] = oracle.gen_keys16(12, n)[::1000]
    N.set_option("pass1_bins", 1)
    blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    blm.add_many(_dev(keys))
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    ob.add_keys(keys)
    assert np.array_equal(_table(blm), ob.bloom)
    blm.set_engine_option("bloom_lookup", 3)
    probe = np.concatenate([keys[:5000], oracle.gen_keys16(555_000_000, 5000)])
    assert np.array_equal(blm.check_many(_dev(probe)).cpu().numpy().astype(np.uint8), ob.check_keys(probe).astype(np.uint8))


@pytest.mark.parametrize("n", [1, 63, 64, 1535, 1536, 1537, 3073, 100_001])
def test_short_batches_and_tile_tails(pa, oracle, N, n):
    keys = oracle.gen_keys16(1234, n)
    N.set_option("pass1_bins", 1)
    blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    blm.add_many(_dev(keys))
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    ob.add_keys(keys)
    assert np.array_equal(_table(blm), ob.bloom)
    blm.set_engine_option("bloom_lookup", 3)
    probe = np.concatenate([keys, oracle.gen_keys16(99_000_000, 777)])
    assert np.array_equal(blm.check_many(_dev(probe)).cpu().numpy().astype(np.uint8), ob.check_keys(probe).astype(np.uint8))


def test_eight_byte_keys_and_rounds(pa, oracle, N):
    """the 8-byte layout goes through the bins as well; partition_max_keys cuts the batch into rounds (every round starts its segments again)"""
    rng = np.random.default_rng(5)
    n = 600_000
    k8 = rng.integers(0, 256, size=(n, 8), dtype=np.uint8)
    N.set_option("pass1_bins", 1)
    N.set_option("partition_max_keys", 150_000)
    blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    blm.add_many(_dev(k8))
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    ob.add_varlen([bytes(r) for r in k8[:50_000]])
    got = blm.check_many(_dev(k8)).cpu().numpy()
    assert bool(got.all())
    N.set_option("pass1_bins", 0)
    ref = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    ref.add_many(_dev(k8))
    assert np.array_equal(_table(blm), _table(ref))
    sub = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    N.set_option("pass1_bins", 1)
    sub.add_many(_dev(k8[:50_000]))
    assert np.array_equal(_table(sub), ob.bloom)


@pytest.mark.parametrize("width,depth", [(2**20, 5), (2**18, 7), (100_003, 4)])
def test_weighted_cms_adds_through_the_bins(pa, oracle, N, width, depth):
    """weighted CountMinSketch adds in the compact probe format (weights 0 .. 15 as 20-bit fields) go through the bins as well:
    countminsketch.py:267-288.  Weights of 0, of 15, of 16 and more (straight to the table: exact saturating add) and one key 300 times in a
    tile (its bins fill up: the rest takes the exact fallback WITH its weight); elements_added sums every weight."""
    n = 700_003
    keys = oracle.gen_keys16(21, n)
    keys[4000:4300] = keys[17]
    w = oracle.gen_weights(21, n).astype(np.int64)          # 1 .. 7
    w[::97] = 0
    w[5::1013] = 15
    old = N.get_option("cms_small_weights")
    try:
        N.set_option("cms_small_weights", 2)                 # the compact format whatever the hint says
        tabs = []
        for bins in (1, 0):
            N.set_option("pass1_bins", bins)
            cms = pa.CountMinSketch(width=width, depth=depth)
            cms.add_many(_dev(keys), _dev(w.astype(np.int32)))
            w2 = w.copy()
            w2[3::5000] = 16 + (np.arange(w2[3::5000].shape[0]) % 1000)   # big weights inside a compact-format batch
            cms.add_many(_dev(keys), _dev(w2.astype(np.int32)))
            tabs.append((np.frombuffer(bytes(cms._bins), dtype=np.int32).copy(), cms.elements_added))
            del cms
        oc = oracle.OracleCMS(width, depth)
        oc.add_keys(keys, w.astype(np.int32))
        oc.add_keys(keys, w2.astype(np.int32))
        for tab, els in tabs:
            assert np.array_equal(tab, oc.bins) and els == oc.els_added
    finally:
        N.set_option("cms_small_weights", old)


@pytest.mark.parametrize("width,depth", [(2**20, 5), (2**18, 7), (100_003, 4), (2**16, 8)])
def test_cms_lookups_with_bin_table_positions(pa, oracle, N, width, depth):
    """return-trip lookups (countminsketch.py:332-340, 429-453) of tables the bins FILLED: pass 1 of the lookups themselves keeps the counting
    sort (the bins measured slower there, profiles/r06_ab_lookup_bins.txt), so the option must not change an answer -- min, mean and mean-min
    queries against the oracle, with keys that repeat 400 times inside a tile"""
    n = 500_009
    keys = oracle.gen_keys16(31, n)
    w = oracle.gen_weights(31, n).astype(np.int32)
    probe = np.concatenate([keys[: n // 2], oracle.gen_keys16(800_000_000, n // 2)])
    hot = probe.copy()
    hot[1000:1400] = hot[3]                                    # 400 copies of one key in a tile
    for query in ("min", "mean", "mean-min"):
        cls = {"min": pa.CountMinSketch, "mean": pa.CountMeanSketch, "mean-min": pa.CountMeanMinSketch}[query]
        oc = oracle.OracleCMS(width, depth, query=query)
        oc.add_keys(keys, w)
        for bins in (1, 0):
            N.set_option("pass1_bins", bins)
            cms = cls(width=width, depth=depth)
            cms.add_many(_dev(keys), _dev(w))
            for p in (probe, hot):
                got = cms.check_many(_dev(p)).cpu().numpy()
                assert np.array_equal(got.astype(np.int64), oc.check_keys(p).astype(np.int64)), (query, bins)
            del cms


@pytest.mark.parametrize("est,fpr", [(28005615, 0.01), (10_000_000, 0.01), (28005615, 0.05)])
def test_bloom_return_trip_lookups_with_bin_table_positions(pa, oracle, N, est, fpr):
    """bloom_lookup = 1 (the return trip: one byte per group of six probes comes back, bloom.py:261-272) on filters the bins / the counting
    sort filled"""
    n = 900_001
    keys = oracle.gen_keys16(41, n)
    probe = np.concatenate([keys[: n // 2], oracle.gen_keys16(600_000_000, n // 2)])
    probe[2000:2300] = probe[n - 5]                            # an absent key 300 times in one tile
    for bins in (1, 0):
        N.set_option("pass1_bins", bins)
        blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
        blm.add_many(_dev(keys))
        if bins:
            ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
            ob.add_keys(keys)
            want = ob.check_keys(probe).astype(np.uint8)
        blm.set_engine_option("bloom_lookup", 1)
        assert np.array_equal(blm.check_many(_dev(probe)).cpu().numpy().astype(np.uint8), want)
        assert bool(blm.check_many(_dev(keys)).all())
        del blm


@pytest.mark.parametrize("est", [2_000_000, 7_000_000])
def test_cbf_lookups_with_bin_table_positions(pa, oracle, N, est):
    """CountingBloomFilter.check (countingbloom.py:166-174) through 32-bit slices (1.9e7 counters) and 4-bit slice images (6.7e7) of tables
    whose unit adds went through the bins (k <= 8, slices that fit) and through the counting sort"""
    n = 600_000
    keys = oracle.gen_keys16(51, n)
    probe = np.concatenate([keys[: n // 2], oracle.gen_keys16(700_000_000, n // 2)])
    old = N.get_option("lookup_nibble_slices")
    try:
        N.set_option("lookup_nibble_slices", 2)                # (the 4-bit images whatever the batch size, where the table is big enough)
        want = None
        for bins in (1, 0):
            N.set_option("pass1_bins", bins)
            cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01)
            cbf.add_many(_dev(keys))
            cbf.add_many(_dev(keys[: n // 3]))
            if want is None:
                oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
                oc.update_keys(keys)
                oc.update_keys(keys[: n // 3])
                want = oc.check_keys(probe)
            assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().astype(np.uint32), want)
            del cbf
    finally:
        N.set_option("lookup_nibble_slices", old)
