"""remove_many is a TRANSACTION (round 4): unordered execution gives the reference's table whenever the result does not depend on
the order inside the batch; the engine checks that while it decrements and, where it fails, puts the counters back and runs the
batch in order (k_cbf_ordered).  So ``add_many`` / ``remove_many`` match the sequential reference (countingbloom.py:135-155,
:186-208) for ILL-FORMED streams too -- removes of absent keys, duplicates beyond their count, keys that run a shared counter dry,
partial removals, frozen counters -- on every path: direct kernels, 32-bit slices, 4-bit slices, update windows."""

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

    names = ("partition", "partition_min_keys", "update_window", "remove_exact")
    old = [N.get_option(k) for k in names]
    yield N
    for k, v in zip(names, old):
        N.set_option(k, v)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _same(cbf, oc):
    got = cbf.table_tensor.cpu().numpy().view(np.uint32)[: cbf.number_bits]
    assert np.array_equal(got, oc.bloom)
    assert cbf.elements_added == oc.els_added


def _rm(cbf, oc, keys, w=None):
    if w is None:
        cbf.remove_many(_dev(keys))
        oc.update_keys(keys, -np.ones(len(keys), dtype=np.int64))
    else:
        cbf.remove_many(_dev(keys), w.astype(np.uint32))
        oc.update_keys(keys, -w.astype(np.int64))


# (est_elements, keys per batch, forced partitioning): the direct kernels, the 32-bit slices, the 4-bit slices of a big table
PATHS = [(20_000, 3_000, False), (400_000, 300_000, True), (3_600_000, 700_000, True)]


@pytest.mark.parametrize("est,n,part", PATHS)
def test_ill_formed_remove_batches_match_the_sequential_reference(pa, oracle, N, est, n, part):
    if part:
        N.set_option("partition", 1)
        N.set_option("partition_min_keys", 1)
    N.set_option("update_window", 0)  # every batch at once: the per-batch transaction itself
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    keys = oracle.gen_keys16(61, n)
    cbf.add_many(_dev(keys))
    oc.update_keys(keys)
    r0 = N.get_option("cbf_ordered_replays")
    # absent keys mixed with present ones: no-ops, NOT order-dependent -- no replay
    absent = oracle.gen_keys16(10**9, n // 4)
    absent = absent[oc.check_keys(absent) == 0]  # (a false positive WOULD be removed -- and may run a real key's counter dry)
    mixed = np.concatenate([absent, keys[: n // 4]])
    _rm(cbf, oc, mixed)
    _same(cbf, oc)
    assert N.get_option("cbf_ordered_replays") == r0
    # every key twice, it was added once: the second removal of each must be a no-op, whichever copy comes first
    dup = np.concatenate([keys[n // 4: n // 2], keys[n // 4: n // 2]])
    _rm(cbf, oc, dup)
    _same(cbf, oc)
    assert N.get_option("cbf_ordered_replays") == r0 + 1
    # ... and again, interleaved with keys that are there and keys that are gone
    dup2 = np.concatenate([keys[n // 2: n // 2 + 500], keys[: 500], keys[n // 2: n // 2 + 500][::-1]])
    _rm(cbf, oc, dup2)
    _same(cbf, oc)
    # partial removals: three copies in, five asked for
    cbf.add_many(_dev(keys[-2000:]), np.full(2000, 2, dtype=np.uint32))
    oc.update_keys(keys[-2000:], np.full(2000, 2, dtype=np.int64))
    _rm(cbf, oc, keys[-2000:], np.full(2000, 5, dtype=np.int64))
    _same(cbf, oc)
    # a frozen counter (2^32 - 1) stays; keys that share it lose their other counters only
    cbf.add_many(_dev(keys[-10:]), np.full(10, 2**32 - 1, dtype=np.uint32))
    oc.update_keys(keys[-10:], np.full(10, 2**32 - 1, dtype=np.int64))
    cbf.add_many(_dev(keys[: n // 8]))
    oc.update_keys(keys[: n // 8])
    _rm(cbf, oc, np.concatenate([keys[-10:], keys[: n // 8]]))
    _same(cbf, oc)
    assert cbf.batch_diagnostics()["violations"] == 0


def test_well_formed_batches_never_replay(pa, oracle, N):
    N.set_option("update_window", 0)
    cbf = pa.CountingBloomFilter(est_elements=400_000, false_positive_rate=0.01)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    keys = oracle.gen_keys16(71, 600_000)
    w = 1 + (np.arange(len(keys)) % 4)
    cbf.add_many(_dev(keys), w.astype(np.uint32))
    oc.update_keys(keys, w.astype(np.int64))
    r0 = N.get_option("cbf_ordered_replays")
    _rm(cbf, oc, keys[:300_000], w[:300_000])
    _rm(cbf, oc, keys[300_000:], np.ones(300_000, dtype=np.int64))
    _same(cbf, oc)
    assert N.get_option("cbf_ordered_replays") == r0


def test_ill_formed_stream_through_the_update_window(pa, oracle, N):
    """the default API on a big table: removes of absent keys, duplicates beyond their count and a saturated counter, in small
    batches that wait in the window -- the oracle's table, not a violation tally"""
    cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    B = 200_000
    keys = oracle.gen_keys16(81, 8 * B)
    sat = keys[:4]
    cbf.add_many(_dev(sat), np.full(4, 2**32 - 1, dtype=np.uint32))
    oc.update_keys(sat, np.full(4, 2**32 - 1, dtype=np.int64))
    r0 = N.get_option("update_window_replays")
    for b in range(8):
        a = keys[b * B:(b + 1) * B]
        cbf.add_many(_dev(a))
        oc.update_keys(a)
        if b == 2:
            _rm(cbf, oc, np.concatenate([a[: B // 4], oracle.gen_keys16(5 * 10**8, B // 4)]))       # absent keys (and a few false positives)
        elif b == 4:
            _rm(cbf, oc, np.concatenate([a[: B // 4], a[: B // 4]]))                                # twice, added once
        elif b == 5:
            _rm(cbf, oc, sat)                                                                      # frozen: stay
        elif b >= 1:
            _rm(cbf, oc, keys[(b - 1) * B + B // 2: b * B])
    _same(cbf, oc)
    assert N.get_option("update_window_replays") >= r0 + 1  # (the 4-key batch is not window material: it flushes what waits)
    assert cbf.batch_diagnostics()["violations"] == 0
