"""CountingBloomFilter tables beyond one level of 32-bit LDS slices -- BASELINE cfg 4's 2^28 counters (1 GiB) -- through the
4-bit slice images of psk_nibble.hpp: lookups (k_nib_gather + k_nib_collect; countingbloom.py:166-174), unit-weight adds
(k_nib_apply; :135-155) and the validated remove (lookup -> amounts -> masked decrement; :186-208), bit-exact against the
oracle, against the direct kernels and with the options off; the cases the 4-bit images cannot hold (a key whose counters are
all 15 or more; a counter hit 16 times or more in one round) must come out exact through their redo / fallback."""

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

    names = ("partition", "partition_min_keys", "lookup_nibble_slices", "update_nibble_slices", "remove_optimistic")
    old = [N.get_option(k) for k in names]
    N.set_option("partition", 1)
    N.set_option("partition_min_keys", 1)
    N.set_option("lookup_nibble_slices", 2)   # also below the crossover (cells / 16 probes)
    yield N
    for k, v in zip(names, old):
        N.set_option(k, v)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _table(cbf):
    return cbf.table_tensor.cpu().numpy().view(np.uint32)[: cbf.number_bits]


@pytest.mark.parametrize("est", [28005615, 10_000_000, 3_600_000])
def test_cbf_lookups_through_nibble_slices(pa, oracle, force_partition, est):
    """2^28 counters (1024 slices of 2^18), 9.6e7 (366 slices, Barrett) and 3.45e7 (132 slices): min over k counters"""
    N = force_partition
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01)
    m, k = cbf.number_bits, cbf.number_hashes
    assert m >= 2**25
    n = 600_000
    keys = oracle.gen_keys16(5, n)
    w = (np.arange(n, dtype=np.int64) % 4) + 1
    oc = oracle.OracleCBF(m, k)
    cbf.add_many(_dev(keys), w.astype(np.uint32))
    oc.update_keys(keys, w)
    probe = np.concatenate([keys[: n // 2], oracle.gen_keys16(900_000_000, n // 2)])  # half of them were never added
    dp = _dev(probe)
    want = oc.check_keys(probe)
    assert 0 < int((want == 0).sum()) < probe.shape[0]
    got = cbf.check_many(dp).cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, want)
    N.set_option("lookup_nibble_slices", 0)
    assert np.array_equal(cbf.check_many(dp).cpu().numpy().astype(np.uint32), want)
    N.set_option("lookup_nibble_slices", 2)
    # check_alt: the min runs over ALL supplied hashes (countingbloom.py:174), here 9 per key
    hs = np.array([oracle.default_fnv_1a(bytes(kx), 9) for kx in probe[:3000]], dtype=np.uint64)
    hs = np.tile(hs, (100, 1))  # 300 k rows: a batch large enough for the partitioned path
    want_alt = np.tile(np.array([oc.check_alt(row) for row in hs[:3000]], dtype=np.uint32), 100)
    assert np.array_equal(np.asarray(cbf.check_alt_many(hs)).view(np.uint32), want_alt)
    # a key whose counters are all 15 or more: the 4-bit image says "15", the flag-guarded direct kernel gives the exact answer
    heavy = oracle.gen_keys16(123_456_789, 3)
    cbf.add_many(_dev(heavy), np.array([15, 16, 40_000], dtype=np.uint32))
    oc.update_keys(heavy, np.array([15, 16, 40_000], dtype=np.int64))
    probe2 = np.concatenate([heavy, probe[:200_000]])
    want2 = oc.check_keys(probe2)
    assert want2[:3].tolist() == [15, 16, 40_000]
    assert np.array_equal(cbf.check_many(_dev(probe2)).cpu().numpy().astype(np.uint32), want2)


def test_cbf_1GiB_unit_adds_and_validated_removes_through_nibble_deltas(pa, oracle, force_partition):
    """2^28 counters: a 5.2 M-key unit add brings more than cells / 8 probes -> one level of nibble-delta slices (no k_part_split);
    the validated remove = nibble lookup -> amounts (0 for absent keys) -> masked decrement.  One key repeated 40 times in the
    add batch carries past 15 in its slices: those workgroups fall back to exact atomics."""
    N = force_partition
    n = 5_200_000
    cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
    assert cbf.number_bits == 2**28
    oc = oracle.OracleCBF(2**28, 7)
    keys = oracle.gen_keys16(0, n)
    keys[1000:1040] = keys[7]            # 41 copies of key 7 in one batch
    cbf.add_many(_dev(keys))
    oc.update_keys(keys)
    assert np.array_equal(_table(cbf), oc.bloom), "unit add (nibble deltas)"
    assert cbf.elements_added == oc.els_added
    assert int(oc.check_keys(keys[7:8])[0]) >= 41
    # the same batch through the two-level 32-bit path must agree
    N.set_option("update_nibble_slices", 0)
    c2 = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
    c2.add_many(_dev(keys))
    assert torch.equal(c2.table_tensor, cbf.table_tensor)
    del c2
    N.set_option("update_nibble_slices", 1)
    # validated remove, every key present once: the optimistic decrement finds that every counter holds what the batch takes from it --
    # no lookup, no amounts, one pass over the table (4.9 M keys: more than cells / 8 probes)
    present = keys[1100:5_000_000]
    cbf.remove_many(_dev(present))
    oc.update_keys(present, -np.ones(present.shape[0], dtype=np.int64))
    assert np.array_equal(_table(cbf), oc.bloom), "validated remove, optimistic fast path"
    assert cbf.elements_added == oc.els_added
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    N.set_option("remove_optimistic", 0)                            # the same batch again through the exact path: put the keys back first
    cbf.add_many(_dev(present))
    cbf.remove_many(_dev(present))
    N.set_option("remove_optimistic", 1)
    assert np.array_equal(_table(cbf), oc.bloom), "validated remove, exact path (lookup + masked decrement)"
    # present keys + keys that were never added (no-ops: countingbloom.py:200-201): the optimistic decrement underflows, is undone, and the exact path decides
    absent = oracle.gen_keys16(800_000_000, 5_000_000)
    absent = absent[oc.check_keys(absent) == 0]                  # (a false positive would be "removed" by the reference too, and then the
    rm = np.concatenate([keys[5_000_000:], absent])              # result depends on the order inside the batch: not a well-formed stream)
    cbf.remove_many(_dev(rm))
    oc.update_keys(rm, -np.ones(rm.shape[0], dtype=np.int64))
    assert np.array_equal(_table(cbf), oc.bloom), "validated remove with absent keys (undo + exact path)"
    assert cbf.elements_added == oc.els_added
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    present = keys[5_000_000:]
    # lookups afterwards, present / removed / never seen
    probe = np.concatenate([keys[:300_000], present[:300_000], absent[:300_000]])
    assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().astype(np.uint32), oc.check_keys(probe))


def test_cbf_nibble_paths_non_power_of_two_table(pa, oracle, force_partition):
    """9.6e7 counters (Barrett reduction, the last slice is partial): unit add + validated remove + lookups"""
    cbf = pa.CountingBloomFilter(est_elements=10_000_000, false_positive_rate=0.01)
    m, k = cbf.number_bits, cbf.number_hashes
    assert 2**26 < m < 2**27
    n = 2_000_000  # 14 M probes >= cells / 8
    keys = oracle.gen_keys16(31, n)
    oc = oracle.OracleCBF(m, k)
    cbf.add_many(_dev(keys))
    oc.update_keys(keys)
    assert np.array_equal(_table(cbf), oc.bloom)
    absent = oracle.gen_keys16(700_000_000, n // 2 + 300_000)
    rm = np.concatenate([keys[: n // 2], absent[oc.check_keys(absent) == 0]])  # (no false positives: see above)
    cbf.remove_many(_dev(rm))
    oc.update_keys(rm, -np.ones(rm.shape[0], dtype=np.int64))
    assert np.array_equal(_table(cbf), oc.bloom)
    assert cbf.elements_added == oc.els_added
    probe = np.concatenate([keys[-200_000:], rm[-200_000:]])
    assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().astype(np.uint32), oc.check_keys(probe))


def test_small_add_batches_into_a_big_table_are_write_combined_automatically(pa, oracle):
    """no opt-in: unit-weight add_many batches too small to pay for a pass over the 1 GiB table are scattered when they are
    handed over and folded together later (adds commute: countingbloom.py:135-155); every read sees them"""
    from pyprobables_amd import _native as N

    assert N.get_option("auto_combine") == 1
    B = 400_000
    cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
    oc = oracle.OracleCBF(2**28, 7)
    for b in range(14):                                  # 5.6 M keys in 14 batches: the flush takes the pass over the table
        keys = oracle.gen_keys16(b * B, B)
        cbf.add_many(_dev(keys))
        oc.update_keys(keys)
    assert cbf.elements_added == oc.els_added            # (get_counters flushes)
    assert np.array_equal(_table(cbf), oc.bloom)
    # a few more batches, then a lookup: too few probes for a table pass -> drained with atomics; the lookup sees them
    for b in range(14, 17):
        keys = oracle.gen_keys16(b * B, B)
        cbf.add_many(_dev(keys))
        oc.update_keys(keys)
    probe = np.concatenate([oracle.gen_keys16(16 * B, 100_000), oracle.gen_keys16(900_000_000, 100_000)])
    assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().astype(np.uint32), oc.check_keys(probe))
    assert np.array_equal(_table(cbf), oc.bloom)
    # pending adds, then a validated remove of some of them and of absent keys: the remove flushes first
    keys = oracle.gen_keys16(17 * B, B)
    cbf.add_many(_dev(keys))
    oc.update_keys(keys)
    absent = oracle.gen_keys16(700_000_000, 50_000)
    rm = np.concatenate([keys[: B // 2], absent[oc.check_keys(absent) == 0]])
    cbf.remove_many(_dev(rm))
    oc.update_keys(rm, -np.ones(rm.shape[0], dtype=np.int64))
    assert np.array_equal(_table(cbf), oc.bloom)
    assert cbf.elements_added == oc.els_added
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    # clear drops what waits; host batches and a non-16-byte key length go the same way
    cbf.add_many(_dev(oracle.gen_keys16(0, B)))
    cbf.clear()
    assert int(cbf.table_tensor.abs().sum().item()) == 0 and cbf.elements_added == 0
    oc = oracle.OracleCBF(2**28, 7)
    k12 = np.ascontiguousarray(oracle.gen_keys16(5, 300_000)[:, :12])
    cbf.add_many(k12)                                    # host buffer, 12-byte keys
    oc.update_keys(k12)
    cbf.add_many(_dev(k12[:100_000]))
    oc.update_keys(k12[:100_000])
    assert np.array_equal(_table(cbf), oc.bloom)
    # the option off: the same batches take the direct kernels
    N.set_option("auto_combine", 0)
    try:
        c2 = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
        c2.add_many(k12)
        c2.add_many(_dev(k12[:100_000]))
        assert torch.equal(c2.table_tensor, cbf.table_tensor)
    finally:
        N.set_option("auto_combine", 1)


def test_opt_in_combining_mixes_scattered_and_key_lists(pa, oracle, request):
    """combine_updates=True: unit-weight batches wait as scattered probes, weighted ones as key lists; at the flush all adds
    of both land before all removes of both (well-formed stream)"""
    from pyprobables_amd import _native as N

    B = 300_000
    from _util import knob

    knob("combine_scatter", 1)   # (off by default: measured slower on BASELINE cfg 4's 1 M-key batches; bench build only)
    request.addfinalizer(lambda: N.set_option("combine_scatter", 0))
    cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01, combine_updates=True)
    oc = oracle.OracleCBF(2**28, 7)
    k0, k1, k2 = (oracle.gen_keys16(i * B, B) for i in range(3))
    w = (np.arange(B, dtype=np.int64) % 3) + 1
    cbf.add_many(_dev(k0))                                # scattered
    cbf.add_many(_dev(k1), w.astype(np.uint32))           # key list (weighted)
    cbf.remove_many(_dev(k0[: B // 2]))                   # scattered decrement
    cbf.remove_many(_dev(k1[: B // 2]), w[: B // 2].astype(np.uint32))  # key list
    cbf.add_many(_dev(k2))
    oc.update_keys(k0)
    oc.update_keys(k1, w)
    oc.update_keys(k0[: B // 2], -np.ones(B // 2, dtype=np.int64))
    oc.update_keys(k1[: B // 2], -w[: B // 2])
    oc.update_keys(k2)
    assert np.array_equal(_table(cbf), oc.bloom)
    assert cbf.elements_added == oc.els_added
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}


@pytest.mark.parametrize("est,fpr,seed", [
    (14_000_000, 0.2, 1),       # k = 2 (KT rounded up to 8), 4.7e7 counters
    (9_000_000, 0.03, 2),       # k = 5 (exact KT), 6.6e7 counters
    (5_000_000, 0.0002, 3),     # k = 12 (KT = 16: 512-thread tiles, one key per thread), 8.9e7 counters
    (30_000_000, 0.05, 4),      # k = 4, 1.9e8 counters (> 2^27)
])
def test_cbf_nibble_paths_seeded_mix(pa, oracle, force_partition, est, fpr, seed):
    """seeded differential runs over hash counts, table sizes and key layouts: unit and weighted adds, duplicate-heavy batches,
    validated removes with and without absent keys, lookups incl. saturating counters -- every table and answer equals the oracle's"""
    N = force_partition
    rng = np.random.default_rng(seed)
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=fpr)
    m, k = cbf.number_bits, cbf.number_hashes
    assert m >= 2**25
    oc = oracle.OracleCBF(m, k)
    need = m // (8 * k) + 50_000                          # keys that make a unit batch eligible for the table pass
    base = oracle.gen_keys16(seed * 100_000_000, need + 400_000)
    a1 = base[:need].copy()
    a1[rng.integers(0, need, 3000)] = base[5]             # one key 3000 times: its slices carry past 15 and fall back to atomics
    cbf.add_many(_dev(a1))
    oc.update_keys(a1)
    assert np.array_equal(_table(cbf), oc.bloom), "unit add"
    w = rng.integers(1, 5, 300_000).astype(np.int64)
    a2 = base[need:need + 300_000]
    cbf.add_many(_dev(a2), w.astype(np.uint32))           # weighted: the 32-bit paths
    oc.update_keys(a2, w)
    words = [("k%d-é€" % i) for i in range(40_000)]       # ragged str keys (code points > 255), host batch, below every threshold
    cbf.add_many(words)
    for x in words:
        oc.add_alt(oracle.default_fnv_1a(x, k))
    assert np.array_equal(_table(cbf), oc.bloom), "weighted + str adds"
    probe = np.concatenate([a1[:200_000], a2[:100_000], oracle.gen_keys16(777_000_000 + seed, 200_000)])
    assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().astype(np.uint32), oc.check_keys(probe)), "lookups"
    # validated removes: all present (optimistic path), then a mix with absent keys (undo + exact path)
    distinct = np.unique(a1, axis=0)
    distinct = distinct[~(distinct == base[5]).all(axis=1)]
    r1 = distinct[: need - 10_000] if distinct.shape[0] >= need - 10_000 else distinct
    cbf.remove_many(_dev(r1))
    oc.update_keys(r1, -np.ones(r1.shape[0], dtype=np.int64))
    assert np.array_equal(_table(cbf), oc.bloom), "validated remove, all present"
    absent = oracle.gen_keys16(888_000_000 + seed, need)
    absent = absent[oc.check_keys(absent) == 0]
    r2 = np.concatenate([a2[:100_000], absent])
    cbf.remove_many(_dev(r2))                             # (weights default to 1: a2's keys keep w - 1)
    oc.update_keys(r2, -np.ones(r2.shape[0], dtype=np.int64))
    assert np.array_equal(_table(cbf), oc.bloom), "validated remove with absent keys"
    assert cbf.elements_added == oc.els_added
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().astype(np.uint32), oc.check_keys(probe)), "lookups after the removes"


def test_borrowed_key_batches_are_hashed_where_they_lie(pa, oracle):
    """combine_updates="borrow" (PSK_DEVICE_BORROWED): device batches of 16-byte keys are neither copied nor read until the flush, which
    runs ONE pass 1 over all of them (KeysFixed16Multi); removes are plain decrements after the window's adds (well-formed stream);
    weighted and non-16-byte batches in the same window are copied as usual"""
    B, nb = 400_000, 14                                   # 5.6 M adds: the flush takes the pass over the table
    cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01, combine_updates="borrow")
    oc = oracle.OracleCBF(2**28, 7)
    held = []
    for b in range(nb):
        keys = oracle.gen_keys16(b * B, B)
        dk = _dev(keys)
        held.append(dk)                                   # (the sketch holds a reference too; the caller must not overwrite the tensor)
        cbf.add_many(dk)
        oc.update_keys(keys)
        if b >= 1:
            prev = oracle.gen_keys16((b - 1) * B, B // 2)
            cbf.remove_many(held[b - 1][: B // 2])        # a view into a borrowed tensor
            oc.update_keys(prev, -np.ones(B // 2, dtype=np.int64))
        if b == 5:                                        # a weighted batch and 12-byte keys in the same window: copied
            w = (np.arange(B, dtype=np.int64) % 3) + 1
            extra = oracle.gen_keys16(900_000_000, B)
            cbf.add_many(_dev(extra), w.astype(np.uint32))
            oc.update_keys(extra, w)
            k12 = np.ascontiguousarray(oracle.gen_keys16(950_000_000, 100_000)[:, :12])
            cbf.add_many(_dev(k12))
            oc.update_keys(k12)
    assert np.array_equal(_table(cbf), oc.bloom)
    assert cbf.elements_added == oc.els_added
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    # a short window (too few probes for a pass over the table: drained with atomics), ended by a lookup
    keys = oracle.gen_keys16(nb * B, B)
    dk = _dev(keys)
    cbf.add_many(dk)
    oc.update_keys(keys)
    probe = np.concatenate([keys[:100_000], oracle.gen_keys16(990_000_000, 100_000)])
    assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().astype(np.uint32), oc.check_keys(probe))
    assert np.array_equal(_table(cbf), oc.bloom)
    # clear drops what waits
    cbf.add_many(dk)
    cbf.clear()
    assert int(cbf.table_tensor.abs().sum().item()) == 0 and cbf.elements_added == 0


def test_optimistic_remove_with_a_key_repeated_in_the_batch(pa, oracle, force_partition):
    """a key held 40 times, removed 20 times in ONE batch next to 5 M other present keys: its counters take more hits than a 4-bit
    delta holds, so their slices decrement with atomics inside the optimistic pass -- still every counter holds what the batch takes
    (40 >= 20), no flag, no undo; the result equals the reference's 20 sequential removes (countingbloom.py:186-208)"""
    n = 5_000_000
    cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
    oc = oracle.OracleCBF(2**28, 7)
    keys = oracle.gen_keys16(0, n)
    hot = oracle.gen_keys16(999_000_000, 1)
    cbf.add_many(_dev(keys))
    cbf.add_many(_dev(hot), 40)
    oc.update_keys(keys)
    oc.update_keys(hot, np.array([40], dtype=np.int64))
    rm = keys.copy()
    rm[np.arange(100, 100 + 20 * 1000, 1000)] = hot[0]   # 20 copies of the hot key replace 20 ordinary keys
    cbf.remove_many(_dev(rm))
    oc.update_keys(rm, -np.ones(n, dtype=np.int64))
    assert np.array_equal(_table(cbf), oc.bloom)
    assert cbf.elements_added == oc.els_added
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    assert int(cbf.check_many(_dev(np.repeat(hot, 300_000, axis=0))).cpu().numpy()[0]) == 20


def test_destroy_applies_waiting_updates_to_a_caller_owned_table(pa, oracle):
    """C ABI, ext_table: a handle over the caller's table is destroyed while automatically write-combined adds still wait in its
    segments -- psk_destroy applies them first (the table outlives the handle; include/psk.h)"""
    import ctypes as C
    from pyprobables_amd import _native as N

    L = N.lib()
    m, k, n = 2**25 + 12_345, 5, 150_000                 # more than 2^24 counters: the batch waits as scattered probes
    table = torch.zeros(int(L.psk_cbf_table_bytes(m)) // 4, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    h = C.c_void_p()
    N.check(L.psk_cbf_create(m, k, 0, table.data_ptr(), C.byref(h)))
    keys = oracle.gen_keys16(31, n)
    dk = _dev(keys)
    torch.cuda.synchronize()
    N.check(L.psk_cbf_add(h, N.KEYS_FIXED, dk.data_ptr(), None, n, 16, None, N.DEVICE, None))
    torch.cuda.synchronize()
    assert int(table.count_nonzero().item()) == 0        # nothing has reached the table yet
    N.check(L.psk_destroy(h))
    oc = oracle.OracleCBF(m, k)
    oc.update_keys(keys)
    assert np.array_equal(table.cpu().numpy().view(np.uint32)[:m], oc.bloom)


def test_repeated_lookups_keep_the_4bit_images_until_the_table_changes(pa, oracle, force_partition):
    """read-mostly tables: the second lookup in a row that finds the table unchanged leaves its 4-bit slice images behind and the following
    ones load them instead of the 32-bit table (psk_sketch::shadow); EVERY way of changing the table must drop them -- adds (direct,
    write-combined, weighted), removes, clear, writes through the table tensor, import -- and so must a lookup on another stream"""
    N = force_partition
    hits = lambda: N.get_option("cbf_lookup_shadow_hits")
    cbf = pa.CountingBloomFilter(est_elements=10_000_000, false_positive_rate=0.01)   # 9.6e7 counters, non power of two
    m, k = cbf.number_bits, cbf.number_hashes
    oc = oracle.OracleCBF(m, k)
    base = oracle.gen_keys16(77, 700_000)
    cbf.add_many(_dev(base[:400_000]))
    oc.update_keys(base[:400_000])
    probe = np.concatenate([base[:300_000], base[400_000:]])       # the second half is absent for now
    dp = _dev(probe)

    def looks(times, expect_hits):
        h0 = hits()
        for _ in range(times):
            assert np.array_equal(cbf.check_many(dp).cpu().numpy().astype(np.uint32), oc.check_keys(probe))
        assert hits() - h0 == expect_hits, (hits() - h0, expect_hits)

    looks(4, 2)                                  # plain, build, load, load
    cbf.add_many(_dev(base[400_000:450_000]))    # a small unit batch (waits as scattered probes; the lookup's flush applies it)
    oc.update_keys(base[400_000:450_000])
    looks(3, 1)
    w = np.full(20_000, 3, dtype=np.uint32)
    cbf.add_many(_dev(base[450_000:470_000]), w)  # weighted: the 32-bit paths
    oc.update_keys(base[450_000:470_000], w.astype(np.int64))
    looks(3, 1)
    cbf.remove_many(_dev(base[:100_000]))        # validated remove (its own internal lookup must neither build nor load the images)
    oc.update_keys(base[:100_000], -np.ones(100_000, dtype=np.int64))
    looks(3, 1)
    big = oracle.gen_keys16(5_000_000, 2_000_000)  # a batch large enough for the table pass
    cbf.add_many(_dev(big))
    oc.update_keys(big)
    looks(3, 1)
    cbf.remove_many(_dev(big))                   # optimistic decrement of the whole batch
    oc.update_keys(big, -np.ones(big.shape[0], dtype=np.int64))
    looks(3, 1)
    # a write from outside through the table tensor: taking the property tells the engine -- and while somebody holds the tensor NOTHING is
    # kept (round 4: a write at any later time would make kept images stale without the engine knowing) ...
    t = cbf.table_tensor
    idx = [int(h % m) for h in oracle.default_fnv_1a(bytes(probe[-1]), k)]
    for c in set(idx):
        t[c] += 2
        oc.bloom[c] += 2
    looks(3, 0)
    for c in set(idx):       # ... so a later write through the same tensor, never announced, is seen as well
        t[c] += 1
        oc.bloom[c] += 1
    looks(3, 0)
    cbf.table_released()     # the holder is done (it takes the property again before it writes any more): images are kept again
    looks(3, 1)
    # another stream: the images were built on the first one
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()
    h0 = hits()
    with torch.cuda.stream(s2):
        got = cbf.check_many(dp)
    s2.synchronize()
    assert np.array_equal(got.cpu().numpy().astype(np.uint32), oc.check_keys(probe)) and hits() == h0
    # several rounds in ONE call: the first round leaves the images behind for the others
    N.set_option("partition_max_keys", 200_000)
    try:
        cbf.add_many(_dev(base[470_000:471_000]))
        oc.update_keys(base[470_000:471_000])
        h0 = hits()
        assert np.array_equal(cbf.check_many(dp).cpu().numpy().astype(np.uint32), oc.check_keys(probe))
        assert hits() - h0 == 1                  # (counted per call)
    finally:
        N.set_option("partition_max_keys", 1 << 26)
    cbf.clear()
    oc = oracle.OracleCBF(m, k)
    looks(3, 1)
    # below the crossover of the plain pass (here 1.5e6 <= probes < 6e6) a batch goes direct -- until the third lookup in a row finds the
    # table unchanged: that one takes the slices and leaves the images behind, the following ones load them
    N.set_option("lookup_nibble_slices", 1)
    cbf.add_many(_dev(base[:300_000]))
    oc.update_keys(base[:300_000])
    small, want_small = dp[:400_000], oc.check_keys(probe[:400_000])
    h0 = hits()
    for _ in range(5):
        assert np.array_equal(cbf.check_many(small).cpu().numpy().astype(np.uint32), want_small)
    assert hits() - h0 == 2
    tiny = dp[:100_000]                           # fewer than cells / 64 probes: always direct
    assert np.array_equal(cbf.check_many(tiny).cpu().numpy().astype(np.uint32), want_small[:100_000]) and hits() - h0 == 2
    N.set_option("lookup_nibble_slices", 2)
    # the option off: same answers, nothing kept
    N.set_option("cbf_lookup_shadow", 0)
    try:
        cbf.add_many(_dev(base[:50_000]))
        oc.update_keys(base[:50_000])
        looks(3, 0)
    finally:
        N.set_option("cbf_lookup_shadow", 1)


@pytest.mark.parametrize("update_pipe,lookup_pipe", [(1, 0), (3, 1), (0, 1), (0, 0)])
@pytest.mark.parametrize("est", [28005615, 10_000_000])
def test_pipelined_table_passes_and_their_ab_partners_agree_with_the_oracle(pa, oracle, force_partition, est, update_pipe, lookup_pipe):
    """Round 4: k_nib_apply_pipe (persistent workgroups, the fold of one slice under the probe groups of the next; options 1 =
    nontemporal, 3 = plain table accesses) and k_nib_gather_pipe (option `nibble_lookup_pipe`), against k_nib_apply / k_nib_gather
    (0) and the oracle: unit adds with a key repeated 41 times (its slices overflow their 4-bit deltas: exact atomics, and the slice
    behind them goes in unpipelined), the optimistic decrement, its undo + exact path, lookups.  est = 10 M: 9.6e7 counters, Barrett,
    the table ends inside the last slice.  countingbloom.py:135-208."""
    from _util import knob, knob_value

    N = force_partition
    old = (knob_value("nibble_update_pipe", 1), knob_value("nibble_lookup_pipe", 0))
    if (update_pipe, lookup_pipe) != (1, 0):  # (the A/B partners exist in the bench build only; the shipped library runs (1, 0))
        knob("nibble_update_pipe", update_pipe)
        knob("nibble_lookup_pipe", lookup_pipe)
    try:
        cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01)
        m, k = cbf.number_bits, cbf.number_hashes
        n = max(2_400_000, m // (8 * k) + 200_000)   # more than cells / 8 probes: the pass over the table
        oc = oracle.OracleCBF(m, k)
        keys = oracle.gen_keys16(11, n)
        keys[2000:2040] = keys[9]
        cbf.add_many(_dev(keys))
        oc.update_keys(keys)
        assert np.array_equal(_table(cbf), oc.bloom), "unit add"
        present = keys[2100 : n - 100_000]
        cbf.remove_many(_dev(present))
        oc.update_keys(present, -np.ones(present.shape[0], dtype=np.int64))
        assert np.array_equal(_table(cbf), oc.bloom), "optimistic decrement"
        absent = oracle.gen_keys16(700_000_000, n)
        absent = absent[oc.check_keys(absent) == 0]
        rm = np.concatenate([keys[n - 100_000 :], absent])
        cbf.remove_many(_dev(rm))
        oc.update_keys(rm, -np.ones(rm.shape[0], dtype=np.int64))
        assert np.array_equal(_table(cbf), oc.bloom), "decrement undone, exact path"
        assert cbf.elements_added == oc.els_added
        probe = np.concatenate([keys[:400_000], absent[:400_000], keys[n - 400_000 :]])
        for _ in range(2):  # (the second lookup of an unchanged table may load kept images: k_nib_gather either way)
            assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().astype(np.uint32), oc.check_keys(probe))
    finally:
        if (update_pipe, lookup_pipe) != (1, 0):
            N.set_option("nibble_update_pipe", old[0])
            N.set_option("nibble_lookup_pipe", old[1])
