"""CountingBloomFilter UPDATE WINDOWS (psk_window.hpp, round 4): small add_many / remove_many batches into a big table wait
together, in arrival order, and reach the table in ONE pass -- through the plain default API, with the reference's semantics
(countingbloom.py:135-155, :186-208) for ANY stream: the fold proves, phase by phase, that every remove would have succeeded at its
own position of the stream; a window for which it cannot (removes of absent keys ...) is undone and replayed batch by batch.
Every case compares the whole table and ``elements_added`` with the sequential oracle."""

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


@pytest.fixture(params=[4, 8], ids=["nibble-image", "byte-image"])
def N(request):
    """every case runs under both forms of the fold's LDS image (option update_window_image): 4 bits per counter, one workgroup per
    2^18-counter slice (default), and 8 bits, two workgroups per slice"""
    from pyprobables_amd import _native as N

    import gc

    from _util import knob

    # the fold / replay counters the tests read are PROCESS-wide: a sketch of an earlier test that is collected in the middle of this one
    # flushes its waiting window in psk_destroy and is counted here -- collect what is garbage first
    gc.collect()
    names = ("update_window", "update_window_keys", "update_window_force_fail")
    old = [N.get_option(k) for k in names]
    if request.param != 4:
        knob("update_window_image", request.param)  # (the byte-image fold is the round-4 A/B partner: bench build only)
    yield N
    for k, v in zip(names, old):
        N.set_option(k, v)
    if request.param != 4:
        N.set_option("update_window_image", 4)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _table(cbf):
    return cbf.table_tensor.cpu().numpy().view(np.uint32)[: cbf.number_bits]


def _stream(oracle, nb, B, seed=11):
    """BASELINE cfg 4's shape: batch b adds B keys and (b >= 1) removes the first half of batch b - 1"""
    keys = oracle.gen_keys16(seed, nb * B)
    ops = []
    for b in range(nb):
        ops.append((False, keys[b * B:(b + 1) * B]))
        if b >= 1:
            ops.append((True, keys[(b - 1) * B:(b - 1) * B + B // 2]))
    return ops


def _run(cbf, oc, ops):
    for rem, kk in ops:
        if rem:
            cbf.remove_many(_dev(kk))
            oc.update_keys(kk, -np.ones(len(kk), dtype=np.int64))
        else:
            cbf.add_many(_dev(kk))
            oc.update_keys(kk)


def _same(cbf, oc):
    assert np.array_equal(_table(cbf), oc.bloom)
    assert cbf.elements_added == oc.els_added


@pytest.mark.parametrize("est", [3_600_000, 10_000_000])
def test_mixed_stream_is_folded_in_one_pass_and_matches_the_oracle(pa, oracle, N, est):
    """3.45e7 counters (132 slices of 2^18: two fold workgroups per slice) and 9.6e7 (Barrett indices)"""
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01)
    m, k = cbf.number_bits, cbf.number_hashes
    assert m > 2**24
    B = 200_000 if est < 5_000_000 else 400_000
    assert B * k < m // 8  # every batch is "small": it waits
    ops = _stream(oracle, 12, B)
    oc = oracle.OracleCBF(m, k)
    folds, replays = N.get_option("update_window_folds"), N.get_option("update_window_replays")
    _run(cbf, oc, ops)
    _same(cbf, oc)  # (reading the table flushes the window)
    assert N.get_option("update_window_folds") == folds + 1 and N.get_option("update_window_replays") == replays
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    # lookups see the folded table
    probe = np.concatenate([ops[0][1][:5000], ops[-2][1][:5000], oracle.gen_keys16(777, 5000)])
    assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().view(np.uint32), oc.check_keys(probe))


def test_window_over_a_loaded_table_and_across_flushes(pa, oracle, N):
    """removes of keys that an EARLIER window added (the proof reads the table's own counters), lookups between the batches"""
    cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01)
    m, k = cbf.number_bits, cbf.number_hashes
    oc = oracle.OracleCBF(m, k)
    base = oracle.gen_keys16(3, 2_000_000)
    cbf.add_many(_dev(base))  # a big batch: straight to the table
    oc.update_keys(base)
    B = 250_000
    fresh = oracle.gen_keys16(4, 8 * B)
    folds = N.get_option("update_window_folds")
    for r in range(2):
        for b in range(4):
            a = fresh[(4 * r + b) * B:(4 * r + b + 1) * B]
            d = base[(4 * r + b) * B // 2:(4 * r + b + 1) * B // 2]  # present since the first batch
            cbf.add_many(_dev(a))
            oc.update_keys(a)
            cbf.remove_many(_dev(d))
            oc.update_keys(d, -np.ones(len(d), dtype=np.int64))
        got = cbf.check_many(_dev(fresh[: 4 * (r + 1) * B: 997])).cpu().numpy().view(np.uint32)  # flushes
        assert np.array_equal(got, oc.check_keys(fresh[: 4 * (r + 1) * B: 997]))
    _same(cbf, oc)
    assert N.get_option("update_window_folds") == folds + 2


def test_removes_of_absent_keys_undo_the_fold_and_replay(pa, oracle, N):
    """countingbloom.py:200-201: a remove of an absent key is a no-op -- the window cannot prove such a stream, undoes its fold
    and replays the batches one by one; the result is the oracle's table, not a violation tally"""
    cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01)
    m, k = cbf.number_bits, cbf.number_hashes
    B = 200_000
    ops = _stream(oracle, 12, B)
    absent = oracle.gen_keys16(999, B // 2)
    ops.insert(7, (True, np.concatenate([absent, ops[4][1][B // 2:B // 2 + 1000]])))  # never added + present ones, mixed
    oc = oracle.OracleCBF(m, k)
    folds, replays = N.get_option("update_window_folds"), N.get_option("update_window_replays")
    _run(cbf, oc, ops)
    _same(cbf, oc)
    assert N.get_option("update_window_replays") == replays + 1 and N.get_option("update_window_folds") == folds
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    # the stream keeps coming, well-formed again: after the back-off the fold is back
    more = _stream(oracle, 12, B, seed=12)
    for _ in range(10):
        _run(cbf, oc, more[:3])
        _same(cbf, oc)
        more_rm = [(True, more[0][1][B // 2:]), (True, more[1][1])]  # put the counters back for the next turn
        _run(cbf, oc, more_rm)
    _run(cbf, oc, more)
    _same(cbf, oc)
    assert N.get_option("update_window_folds") >= folds + 1


def test_forced_failure_undoes_exactly(pa, oracle, N):
    """the undo kernel alone: a well-formed window whose verdict is forced to 'failed' must come out the same through the replay"""
    cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01)
    m, k = cbf.number_bits, cbf.number_hashes
    oc = oracle.OracleCBF(m, k)
    pre = oracle.gen_keys16(21, 1_500_000)
    cbf.add_many(_dev(pre), np.full(len(pre), 3, dtype=np.uint32))
    oc.update_keys(pre, np.full(len(pre), 3, dtype=np.int64))
    N.set_option("update_window_force_fail", 1)
    replays = N.get_option("update_window_replays")
    _run(cbf, oc, _stream(oracle, 10, 250_000, seed=5))
    _same(cbf, oc)
    assert N.get_option("update_window_replays") == replays + 1


def test_counters_beyond_the_byte_image_take_the_atomics(pa, oracle, N):
    """a part that meets a counter of 254 or more drops its image and applies its probes to the table itself -- exact"""
    cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01)
    m, k = cbf.number_bits, cbf.number_hashes
    oc = oracle.OracleCBF(m, k)
    B = 200_000
    ops = _stream(oracle, 12, B, seed=31)
    hot = ops[2][1][:50]
    cbf.add_many(_dev(hot), np.full(50, 1000, dtype=np.uint32))  # 350 counters at 1000 and more
    oc.update_keys(hot, np.full(50, 1000, dtype=np.int64))
    frozen = ops[4][1][:3]
    cbf.add_many(_dev(frozen), np.full(3, 2**32 - 1, dtype=np.uint32))  # and 21 frozen ones (2^32 - 1)
    oc.update_keys(frozen, np.full(3, 2**32 - 1, dtype=np.int64))
    _run(cbf, oc, ops)  # batch 2 / 4 add to and batch 3 / 5's removes take from those counters
    _same(cbf, oc)


def test_duplicates_inside_the_window(pa, oracle, N):
    """the same keys added several times and removed as often, and 300 copies of one key (a counter that crosses 254 mid-window)"""
    cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01)
    m, k = cbf.number_bits, cbf.number_hashes
    oc = oracle.OracleCBF(m, k)
    kk = oracle.gen_keys16(41, 150_000)
    rep = np.concatenate([kk[:1]] * 300 + [kk[1:100_000]])
    ops = [(False, kk), (False, kk), (True, kk), (False, rep), (True, kk), (False, kk), (True, rep), (False, kk), (True, kk), (True, kk)]
    ops = ops * 2
    _run(cbf, oc, ops)
    _same(cbf, oc)


def test_option_off_and_host_batches(pa, oracle, N):
    cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01)
    m, k = cbf.number_bits, cbf.number_hashes
    oc = oracle.OracleCBF(m, k)
    ops = _stream(oracle, 12, 200_000, seed=51)
    folds = N.get_option("update_window_folds")
    for rem, kk in ops:  # numpy batches: copied straight from the caller's buffer, which may be reused at once
        buf = kk.copy()
        (cbf.remove_many if rem else cbf.add_many)(buf)
        buf[:] = 0
        oc.update_keys(kk, -np.ones(len(kk), dtype=np.int64) if rem else None)
    _same(cbf, oc)
    assert N.get_option("update_window_folds") == folds + 1
    N.set_option("update_window", 0)
    cbf.clear()
    oc = oracle.OracleCBF(m, k)
    _run(cbf, oc, ops[:8])
    _same(cbf, oc)
    assert N.get_option("update_window_folds") == folds + 1


def test_counters_around_the_nibble_image_limit(pa, oracle, N):
    """keys added 12 .. 17 times inside ONE window and partly removed again: counters that end at 13, 14 (the last values a 4-bit image
    follows), 15 and beyond (the slice drops its image and takes the atomics), next to ordinary traffic; then a second window that removes
    from those counters (loaded as 15 = "cannot follow" by the nibble image) -- countingbloom.py:135-155, :186-208"""
    cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01)
    m, k = cbf.number_bits, cbf.number_hashes
    oc = oracle.OracleCBF(m, k)
    base = oracle.gen_keys16(71, 300_000)
    hot = base[:6]
    ops = [(False, base[:150_000])]
    for r in range(17):  # key i of `hot` is added 12 + i times in all (one copy came with the first batch)
        sel = np.concatenate([hot[i:i + 1] for i in range(6) if r < 11 + i] + [base[150_000 + r * 5_000:150_000 + (r + 1) * 5_000]])
        ops.append((False, sel))
        if r % 4 == 3:
            ops.append((True, base[r * 2_000:(r + 1) * 2_000 + 6]))  # ordinary removes in between (keys 6.. of the first batch)
    ops.append((True, np.concatenate([hot, hot])))  # two copies of each hot key leave again, still in the same window
    _run(cbf, oc, ops)
    _same(cbf, oc)
    ops2 = [(True, hot), (False, base[200_000:260_000]), (True, hot), (True, base[200_000:230_000])]
    _run(cbf, oc, ops2)
    _same(cbf, oc)


def test_a_fold_leaves_the_lookups_kept_images_up_to_date(pa, oracle, N):
    """lookups of a big table keep 4-bit slice images while it does not change (psk_sketch::shadow); a window fold with nibble images ends
    with exactly those images in LDS and writes them back, so the lookup behind the flush loads them instead of reading the table again --
    and must answer exactly (countingbloom.py:166-174) for present, removed and absent keys.  The byte-image fold leaves them stale.
    (Option update_window_shadow; off by default -- the saving measured within the noise of a round of updates + lookups.)"""
    cbf = pa.CountingBloomFilter(est_elements=20_000_000, false_positive_rate=0.01)   # 1.9e8 counters, 732 slices: runs short enough for the image path
    m, k = cbf.number_bits, cbf.number_hashes
    oc = oracle.OracleCBF(m, k)
    _run(cbf, oc, _stream(oracle, 6, 200_000, seed=91))
    probe = np.concatenate([oracle.gen_keys16(91, 500_000), oracle.gen_keys16(999_000_000, 200_000)])
    dp = _dev(probe)
    from _util import knob, knob_value

    old = N.get_option("lookup_nibble_slices")
    old_sh = knob_value("update_window_shadow", 0)
    knob("update_window_shadow", 1)  # (off by default: measured without gain; bench build only)
    N.set_option("lookup_nibble_slices", 2)  # (the 4-bit lookup path whatever the batch size)
    try:
        for _ in range(3):  # plain, build the images, load them
            assert np.array_equal(cbf.check_many(dp).cpu().numpy().astype(np.uint32), oc.check_keys(probe))
        hits, writes = N.get_option("cbf_lookup_shadow_hits"), N.get_option("update_window_shadow_writes")
        folds = N.get_option("update_window_folds")
        for rnd in range(3):
            ops = _stream(oracle, 12, 400_000, seed=120 + rnd)     # fresh keys: adds + removes of half of them; 7 M operations: one window, one fold
            _run(cbf, oc, ops)
            got = cbf.check_many(dp).cpu().numpy().astype(np.uint32)   # the flush folds the window, then the lookup runs
            assert np.array_equal(got, oc.check_keys(probe))
            more = np.concatenate([ops[0][1][:50_000], ops[2][1][:50_000]])  # keys of this window: removed ones and live ones
            assert np.array_equal(cbf.check_many(_dev(more)).cpu().numpy().astype(np.uint32), oc.check_keys(more))
        assert N.get_option("update_window_folds") == folds + 3
        nib = knob_value("update_window_image", 4) != 8
        assert N.get_option("update_window_shadow_writes") - writes == (3 if nib else 0)
        if nib:
            assert N.get_option("cbf_lookup_shadow_hits") - hits >= 3   # the lookup behind every fold loaded the images it left
        _same(cbf, oc)
    finally:
        N.set_option("lookup_nibble_slices", old)
        N.set_option("update_window_shadow", old_sh)


def test_combined_update_after_window_batches_keeps_the_order(pa, oracle, N):
    """a C caller may mix psk_cbf_add / psk_cbf_remove (update window) with psk_cbf_update_combined (key lists) on ONE handle.
    The window's batches arrived first, so they reach the table first: a windowed remove of an ABSENT key X is a no-op
    (countingbloom.py:198-201) and the combined add of X that follows leaves X present -- not removed."""
    n = 120_000
    keys = oracle.gen_keys16(4242, n)
    dk = _dev(keys)
    cbf = pa.CountingBloomFilter(est_elements=7_100_000, false_positive_rate=0.01)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    L, h = N.lib(), cbf._tab.handle
    st = cbf._tab.stream
    args = (N.KEYS_FIXED, dk.data_ptr(), None, n, 16)
    N.check(L.psk_cbf_remove(h, *args, None, N.DEVICE, st))               # absent keys: waits in the window, removes nothing
    N.check(L.psk_cbf_update_combined(h, *args, None, 0, N.DEVICE, st))    # the same keys, added through the combined path
    N.check(L.psk_cbf_add(h, N.KEYS_FIXED, dk[: n // 2].data_ptr(), None, n // 2, 16, None, N.DEVICE, st))   # window again
    N.check(L.psk_cbf_update_combined(h, N.KEYS_FIXED, dk[: n // 4].data_ptr(), None, n // 4, 16, None, 1, N.DEVICE, st))  # combined remove
    oc.update_keys(keys, -np.ones(n, dtype=np.int64))
    oc.update_keys(keys)
    oc.update_keys(keys[: n // 2])
    oc.update_keys(keys[: n // 4], -np.ones(n // 4, dtype=np.int64))
    cbf._dirty = True
    assert np.array_equal(_table(cbf), oc.bloom)
    assert np.array_equal(cbf.check_many(dk).cpu().numpy().astype(np.uint32), oc.check_keys(keys))


# ----------------------------------------------------------------------------------- borrowed batches (round 5)
def _run_lent(cbf, oc, ops, hold):
    """every batch in its OWN device tensor (the window's pieces are not contiguous), kept alive by the sketch, not by the test"""
    for rem, kk in ops:
        t = _dev(kk)
        if hold is not None:
            hold.append(t.data_ptr())
        if rem:
            cbf.remove_many(t)
            oc.update_keys(kk, -np.ones(len(kk), dtype=np.int64))
        else:
            cbf.add_many(t)
            oc.update_keys(kk)
        del t


@pytest.mark.parametrize("est", [3_600_000, 10_000_000])
def test_borrowed_batches_are_hashed_where_they_lie(pa, oracle, N, est):
    """CountingBloomFilter(borrow_keys=True): the window keeps pointers to the caller's tensors (PSK_DEVICE_BORROWED) -- same table, same
    counts, one fold, and no key list is ever allocated"""
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01, borrow_keys=True)
    m, k = cbf.number_bits, cbf.number_hashes
    B = 200_000 if est < 5_000_000 else 400_000
    ops = _stream(oracle, 12, B)
    ops[3] = (ops[3][0], ops[3][1][:-777])     # batches that do not end on a tile boundary
    ops[6] = (ops[6][0], ops[6][1][:70_001])
    oc = oracle.OracleCBF(m, k)
    folds, replays = N.get_option("update_window_folds"), N.get_option("update_window_replays")
    _run_lent(cbf, oc, ops, None)
    assert cbf._tab.get_option("window_pending_batches") == len(ops)
    assert len(cbf._borrowed) == len(ops)
    _same(cbf, oc)
    assert cbf._tab.get_option("window_pending_batches") == 0
    assert N.get_option("update_window_folds") == folds + 1 and N.get_option("update_window_replays") == replays
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    probe = np.concatenate([ops[0][1][:5000], ops[-2][1][:5000], oracle.gen_keys16(777, 5000)])
    assert np.array_equal(cbf.check_many(_dev(probe)).cpu().numpy().view(np.uint32), oc.check_keys(probe))
    assert not cbf._borrowed  # (the references went with the flush)


def test_borrowed_window_replays_ill_formed_streams_from_the_callers_tensors(pa, oracle, N):
    cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01, borrow_keys=True)
    m, k = cbf.number_bits, cbf.number_hashes
    B = 200_000
    ops = _stream(oracle, 12, B)
    absent = oracle.gen_keys16(999, B // 2)
    ops.insert(7, (True, np.concatenate([absent, ops[4][1][B // 2:B // 2 + 1000]])))
    oc = oracle.OracleCBF(m, k)
    replays = N.get_option("update_window_replays")
    _run_lent(cbf, oc, ops, None)
    _same(cbf, oc)
    assert N.get_option("update_window_replays") == replays + 1


def test_borrowed_copied_and_host_batches_mix_in_one_window(pa, oracle, N):
    """a lent tensor, a host array (copied into the list), a misaligned device view (copied), an adds-only run of several lent batches"""
    cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01, borrow_keys=True)
    m, k = cbf.number_bits, cbf.number_hashes
    oc = oracle.OracleCBF(m, k)
    B = 150_000
    keys = oracle.gen_keys16(21, 8 * B)
    flat = torch.zeros(B * 16 + 8, dtype=torch.uint8, device="cuda")
    folds = N.get_option("update_window_folds")
    for b in range(8):
        kk = keys[b * B:(b + 1) * B]
        if b % 3 == 0:
            cbf.add_many(_dev(kk))                       # lent
        elif b % 3 == 1:
            cbf.add_many(kk)                             # host: copied
        else:
            flat[8:].copy_(_dev(kk).reshape(-1))
            cbf.add_many(flat[8:].view(B, 16))           # 8 bytes off a 16-byte boundary: copied -- and `flat` may be overwritten right away
            flat.zero_()
        oc.update_keys(kk)
        if b >= 1:
            d = keys[(b - 1) * B:(b - 1) * B + B // 2]
            cbf.remove_many(_dev(d))
            oc.update_keys(d, -np.ones(len(d), dtype=np.int64))
    _same(cbf, oc)
    assert N.get_option("update_window_folds") == folds + 1
    # adds only, several lent batches: one phase whose keys do not lie end to end
    more = oracle.gen_keys16(22, 6 * B)
    for b in range(6):
        cbf.add_many(_dev(more[b * B:(b + 1) * B]))
        oc.update_keys(more[b * B:(b + 1) * B])
    _same(cbf, oc)


def test_c_abi_borrowed_add_remove(pa, oracle, N):
    """psk_cbf_add / psk_cbf_remove with PSK_DEVICE_BORROWED on a table too small for windows: applied at once, nothing kept"""
    cbf = pa.CountingBloomFilter(est_elements=100_000, false_positive_rate=0.01, borrow_keys=True)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    kk = oracle.gen_keys16(5, 120_000)
    t = _dev(kk)
    cbf.add_many(t)
    assert cbf._tab.get_option("window_pending_batches") == 0
    t.zero_()  # (not waiting: free to go)
    oc.update_keys(kk)
    cbf.remove_many(_dev(kk[:50_000]))
    oc.update_keys(kk[:50_000], -np.ones(50_000, dtype=np.int64))
    _same(cbf, oc)


@pytest.mark.parametrize("tile", [2048, 4096])
def test_window_tile_sizes_agree(pa, oracle, N, tile):
    """option update_window_tile: 2048- and 4096-key pass-1 tiles forced on a table of few slices (4096 there overflows the fold's registers for
    some slices: they take the atomics) -- the oracle's table either way"""
    old = N.get_option("update_window_tile")
    try:
        N.set_option("update_window_tile", tile)
        cbf = pa.CountingBloomFilter(est_elements=3_600_000, false_positive_rate=0.01)
        oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
        _run(cbf, oc, _stream(oracle, 10, 200_000, seed=31))
        _same(cbf, oc)
    finally:
        N.set_option("update_window_tile", old)
