"""Seeded differential fuzzing of the HIP engine against the C oracle: random table geometries, key layouts,
batch sizes and both kernel families (direct / partitioned).  Bit-exact or it fails."""

import os

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


EXTRA = int(os.environ.get("PSK_FUZZ_EXTRA", "0"))  # soak runs: PSK_FUZZ_EXTRA=500 python -m pytest tests/test_gpu_fuzz.py -m gpu


@pytest.fixture()
def engine_options():
    from pyprobables_amd import _native as N

    old = {k: N.get_option(k) for k in ("partition", "partition_min_keys", "partition_max_keys", "partition_cache_bytes", "partition_two_level_slices",
                                        "bloom_lookup", "even_tiles", "dense_walk_groups", "cms_small_weights", "pass1_bins")}
    yield N
    for k, v in old.items():
        N.set_option(k, v)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _random_keys(rng, n):
    """-> (what to hand to the engine, the same keys as a list of bytes for the oracle)"""
    kind = rng.integers(0, 4)
    if kind == 0:  # fixed length, device resident
        L = int(rng.choice([1, 4, 5, 8, 11, 16, 16, 16, 24, 40]))
        a = rng.integers(0, 256, size=(n, L), dtype=np.uint8)
        return _dev(a), [bytes(r) for r in a]
    if kind == 1:  # fixed length, host array
        L = int(rng.choice([2, 8, 16, 31]))
        a = rng.integers(0, 256, size=(n, L), dtype=np.uint8)
        return a, [bytes(r) for r in a]
    if kind == 2:  # ragged bytes (with duplicates and empties)
        pool = [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in rng.integers(0, 33, size=max(n // 3, 1))]
        ks = [pool[i] for i in rng.integers(0, len(pool), size=n)]
        return ks, ks
    # latin-1 strings (one byte per code point)
    ks = ["k%d-é%s" % (int(i), "x" * int(l)) for i, l in zip(rng.integers(0, n, size=n), rng.integers(0, 9, size=n))]
    return ks, [k.encode("latin-1") for k in ks]


@pytest.mark.parametrize("seed", range(24 + EXTRA))
def test_fuzz_bloom(pa, oracle, engine_options, seed):
    rng = np.random.default_rng(1000 + seed)
    engine_options.set_option("partition", int(rng.integers(0, 2)))
    engine_options.set_option("partition_min_keys", int(rng.choice([1, 1, 4096])))
    engine_options.set_option("partition_max_keys", int(rng.choice([2048, 1 << 25])))
    engine_options.set_option("partition_cache_bytes", int(rng.choice([0, 1 << 16, 240 << 20])))
    engine_options.set_option("partition_two_level_slices", int(rng.choice([0, 2, 512])))
    engine_options.set_option("bloom_lookup", int(seed % 5))   # keyed probes / return trip / chosen per call / tile flags / lazy gathers
    engine_options.set_option("even_tiles", int(seed // 3 % 2))
    engine_options.set_option("dense_walk_groups", (0, 40, 1 << 30)[seed // 2 % 3])  # pass 2: chunked walk / by segment length / end-to-end walk
    engine_options.set_option("pass1_bins", int(seed // 2 % 2))  # pass 1 through fixed-capacity bins / the counting sort (16- and 8-byte keys)
    est = int(rng.choice([50, 3000, 40_000, 200_000, 1_000_000]))
    fpr = float(rng.choice([0.3, 0.05, 0.01, 0.001, 1e-6]))
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.choice([0, 1, 7, 100, 5000, 30_000]))
        eng, ref = _random_keys(rng, n)
        blm.add_many(eng)
        if ref:
            ob.add_varlen(ref)
        probe_eng, probe_ref = _random_keys(rng, int(rng.choice([1, 64, 3000, 20_000])))
        got = blm.check_many(probe_eng)
        got = got.cpu().numpy() if hasattr(got, "cpu") else got
        assert np.array_equal(got.astype(np.uint8), ob.check_varlen(probe_ref))
        if ref:
            again = blm.check_many(eng)
            again = again.cpu().numpy() if hasattr(again, "cpu") else again
            assert bool(again.all())
    assert np.array_equal(np.frombuffer(bytes(blm.bloom), dtype=np.uint8), ob.bloom)
    assert blm._cnt_number_bits_set() == ob.bits_set()


@pytest.mark.parametrize("seed", range(16 + EXTRA))
def test_fuzz_cms(pa, oracle, engine_options, seed):
    rng = np.random.default_rng(2000 + seed)
    engine_options.set_option("partition", int(rng.integers(0, 2)))
    engine_options.set_option("partition_min_keys", int(rng.choice([1, 1, 4096])))
    engine_options.set_option("partition_two_level_slices", int(rng.choice([0, 2, 512])))
    engine_options.set_option("dense_walk_groups", (0, 40, 1 << 30)[seed // 2 % 3])
    engine_options.set_option("cms_small_weights", (1, 2, 1, 0)[seed // 3 % 4])  # weighted adds: compact probe format by the hint / always / never
    width = int(rng.choice([7, 1000, 4096, 65_536, 100_003, 1 << 18]))
    depth = int(rng.choice([1, 3, 5, 8, 11]))
    cms = pa.CountMinSketch(width=width, depth=depth)
    oc = oracle.OracleCMS(width, depth)
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.choice([1, 50, 4000, 25_000]))
        keys = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
        keys[n // 2:] = keys[: n - n // 2]  # duplicates inside the batch
        wmax = int(rng.choice([1, 7, 1000, 100_000]))
        w = rng.integers(1, wmax + 1, size=n).astype(np.int32)
        if rng.integers(0, 3) == 0:
            cms.add_many(_dev(keys))
            oc.add_keys(keys)
        else:
            cms.add_many(_dev(keys), _dev(w))
            oc.add_keys(keys, w)
        if rng.integers(0, 2):
            m = n // 3
            cms.remove_many(keys[:m], w[:m])
            oc.remove_keys(keys[:m], w[:m])
        assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
        assert cms.elements_added == oc.els_added
        for q, oq in (("min", oracle.Q_MIN), ("mean", oracle.Q_MEAN), ("mean-min", oracle.Q_MEANMIN)):
            if q == "mean-min" and width < 2:
                continue
            cms.query_type, oc.query = q, oq
            got = cms.check_many(_dev(keys[:500])).cpu().numpy()
            assert np.array_equal(got.astype(np.int64), oc.check_keys(keys[:500])), q
        cms.query_type, oc.query = "min", oracle.Q_MIN


@pytest.mark.parametrize("seed", range(12 + EXTRA))
def test_fuzz_cbf(pa, oracle, engine_options, seed):
    rng = np.random.default_rng(3000 + seed)
    engine_options.set_option("partition", int(rng.integers(0, 2)))
    engine_options.set_option("partition_min_keys", int(rng.choice([1, 4096])))
    engine_options.set_option("partition_two_level_slices", int(rng.choice([0, 2, 512])))
    engine_options.set_option("dense_walk_groups", (0, 40, 1 << 30)[seed % 3])
    est = int(rng.choice([200, 5000, 30_000, 300_000]))
    fpr = float(rng.choice([0.1, 0.01, 0.001]))
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=fpr)
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    live = []
    for _ in range(int(rng.integers(2, 5))):
        n = int(rng.choice([1, 300, 6000, 20_000]))
        keys = rng.integers(0, 256, size=(n, 12), dtype=np.uint8)
        w = rng.integers(1, 6, size=n).astype(np.uint32)
        cbf.add_many(_dev(keys), w)
        oc.update_keys(keys, w.astype(np.int64))
        live.append((keys, w))
        if rng.integers(0, 2) and live:  # well-formed removal: take back (part of) an earlier batch exactly once
            k0, w0 = live.pop(int(rng.integers(0, len(live))))
            m = len(k0) // 2
            if m:
                cbf.remove_many(_dev(k0[:m]), w0[:m])
                oc.update_keys(k0[:m], -w0[:m].astype(np.int64))
                if len(k0) - m:
                    live.append((k0[m:], w0[m:]))
        assert np.array_equal(np.frombuffer(bytes(cbf.bloom), dtype=np.uint32), oc.bloom)
        assert cbf.elements_added == oc.els_added
        probe = rng.integers(0, 256, size=(1000, 12), dtype=np.uint8)
        probe[:500] = keys[:500] if n >= 500 else probe[:500]
        assert np.array_equal(cbf.check_many(probe), oc.check_keys(probe))
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}


# large tables: 2^20-bit Bloom slices / 2^15-cell counter slices at their maximum, every k class, both partition levels
@pytest.mark.parametrize("seed", range(20 + EXTRA))
def test_fuzz_big_tables(pa, oracle, engine_options, seed):
    rng = np.random.default_rng(5000 + seed)
    engine_options.set_option("partition", 1)
    engine_options.set_option("partition_min_keys", 1)
    engine_options.set_option("dense_walk_groups", (1 << 30, 40, 0)[seed // 2 % 3])
    engine_options.set_option("partition_two_level_slices", int(rng.choice([2048, 64, 256])))
    k_fpr = {3: 0.12, 4: 0.06, 5: 0.03, 6: 0.016, 7: 0.008, 8: 0.004, 10: 0.001, 13: 0.00012}
    k = int(rng.choice(list(k_fpr)))
    n = int(rng.choice([150_000, 600_000, 1_500_000]))
    keys = oracle.gen_keys16(seed * 7_000_000, n)
    dk = _dev(keys)
    if seed % 2 == 0:  # Bloom, 2^28 .. 2^29 bits
        m_target = int(rng.choice([2**28, 3 * 2**27, 2**29]))
        est = int(m_target * 0.4804530139182 / -np.log(k_fpr[k]))
        blm = pa.BloomFilter(est_elements=est, false_positive_rate=k_fpr[k])
        assert blm.number_hashes == k and blm.number_bits >= 2**27
        ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
        blm.add_many(dk[: n // 2])
        ob.add_keys(keys[: n // 2])
        assert np.array_equal(np.frombuffer(bytes(blm.bloom), dtype=np.uint8), ob.bloom)
        assert np.array_equal(blm.check_many(dk).cpu().numpy().astype(np.uint8), ob.check_keys(keys))
    else:  # counters, 2^24 .. 2^27 cells
        cells_target = int(rng.choice([2**24, 2**26, 2**27]))
        est = int(cells_target * 0.4804530139182 / -np.log(k_fpr[k]))
        cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=k_fpr[k])
        assert cbf.number_hashes == k
        oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
        w = rng.integers(1, 9, size=n).astype(np.uint32)
        w[:: 4001] = 3000  # beyond the level-1 inline field of the two-level path
        cbf.add_many(dk, _dev(w.view(np.int32)))
        oc.update_keys(keys, w.astype(np.int64))
        cbf.add_many(dk[: n // 3])
        oc.update_keys(keys[: n // 3], np.ones(n // 3, dtype=np.int64))
        assert np.array_equal(np.frombuffer(bytes(cbf.bloom), dtype=np.uint32), oc.bloom)
        assert cbf.elements_added == oc.els_added
        assert np.array_equal(cbf.check_many(dk[:50_000]).cpu().numpy().view(np.uint32), oc.check_keys(keys[:50_000]))


# CountingBloomFilter state machine on tables large enough for the 4-bit slice images (psk_nibble.hpp): random sequences of unit / weighted
# adds of every size class (direct, automatically write-combined, table pass), validated removes with present and absent keys, lookups above
# and below the crossovers (repeated: the kept images of an unchanged table), clear, writes through the table tensor -- the whole table,
# elements_added and every answer equal the oracle's after every operation
@pytest.mark.parametrize("seed", range(8 + EXTRA // 20))
def test_fuzz_cbf_big_table_state_machine(pa, oracle, engine_options, seed):
    N = engine_options
    rng = np.random.default_rng(9000 + seed)
    names = ("lookup_nibble_slices", "nibble_min_lg_lookup", "nibble_min_lg_update", "auto_combine_keys", "remove_optimistic", "cbf_lookup_shadow")
    old = {k: N.get_option(k) for k in names}
    try:
        N.set_option("partition", 1)
        N.set_option("partition_min_keys", int(rng.choice([1, 4096])))
        N.set_option("nibble_min_lg_lookup", int(rng.choice([20, 23])))
        N.set_option("nibble_min_lg_update", int(rng.choice([20, 24])))
        N.set_option("lookup_nibble_slices", int(rng.choice([1, 1, 2])))
        N.set_option("auto_combine_keys", int(rng.choice([1 << 24, 1 << 19])))
        N.set_option("remove_optimistic", int(rng.integers(0, 2)))
        k_fpr = {4: 0.06, 5: 0.03, 7: 0.008, 10: 0.001}
        k = int(rng.choice(list(k_fpr)))
        cells_target = int(rng.choice([2**23 + 12345, 2**24, 3 * 2**23, 2**25 + 999, 2**26]))
        est = int(cells_target * 0.4804530139182 / -np.log(k_fpr[k]))
        cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=k_fpr[k])
        m = cbf.number_bits
        assert cbf.number_hashes == k and m >= 2**22
        oc = oracle.OracleCBF(m, k)
        pool = oracle.gen_keys16(seed * 50_000_000, 6_000_000)
        live = []          # (start, count) ranges of the pool that are in the filter exactly once more than removed
        cursor = 0

        def table_ok(what):
            assert np.array_equal(cbf.table_tensor.cpu().numpy().view(np.uint32)[:m], oc.bloom), what
            assert cbf.elements_added == oc.els_added, what

        for step in range(int(rng.integers(6, 11))):
            op = int(rng.choice([0, 0, 1, 2, 2, 3, 3, 3, 4, 5]))
            if op == 0 or not live:      # unit add of a fresh range
                n = int(rng.choice([3_000, 60_000, 400_000, 1_500_000]))
                n = min(n, pool.shape[0] - cursor)
                if n <= 0:
                    continue
                ks = pool[cursor:cursor + n]
                cbf.add_many(_dev(ks) if rng.integers(0, 4) else ks)
                oc.update_keys(ks)
                live.append((cursor, n))
                cursor += n
                what = f"step {step}: unit add of {n}"
            elif op == 1:                # weighted add on top of a live range
                s0, n0 = live[int(rng.integers(0, len(live)))]
                n = min(n0, int(rng.choice([2_000, 200_000])))
                w = rng.integers(1, 5, size=n).astype(np.uint32)
                cbf.add_many(_dev(pool[s0:s0 + n]), _dev(w.view(np.int32)))
                oc.update_keys(pool[s0:s0 + n], w.astype(np.int64))
                what = f"step {step}: weighted add of {n}"   # (the extra weight stays in: only one unit per key is ever taken back below)
            elif op == 2:                # validated remove: (part of) a live range, sometimes with absent keys mixed in
                s0, n0 = live.pop(int(rng.integers(0, len(live))))
                n = n0 if rng.integers(0, 2) else max(n0 // 2, 1)
                ks = pool[s0:s0 + n]
                if n0 - n:
                    live.append((s0 + n, n0 - n))
                if rng.integers(0, 2):
                    absent = oracle.gen_keys16(3_000_000_000 + step * 1_000_000 + seed, int(rng.choice([100, 50_000])))
                    absent = absent[oc.check_keys(absent) == 0]
                    ks = np.concatenate([ks, absent])
                cbf.remove_many(_dev(ks))
                oc.update_keys(ks, -np.ones(ks.shape[0], dtype=np.int64))   # (the oracle's remove of an absent key is the reference's no-op)
                what = f"step {step}: remove of {ks.shape[0]}"
            elif op == 3:                # lookups, repeated: present, removed and never-seen keys
                n = int(rng.choice([5_000, 300_000, 1_200_000]))
                hi = max(cursor, 1)
                s0 = int(rng.integers(0, hi))
                probe = np.concatenate([pool[s0:s0 + n // 2], oracle.gen_keys16(4_000_000_000 + step, n - n // 2)])
                dp = _dev(probe)
                want = oc.check_keys(probe)
                for rep in range(int(rng.integers(1, 5))):
                    assert np.array_equal(cbf.check_many(dp).cpu().numpy().view(np.uint32), want), f"step {step}: lookup {rep} of {n}"
                continue
            elif op == 4:                # a write from outside through the table tensor
                t = cbf.table_tensor
                cells = rng.integers(0, m, size=50)
                for c in np.unique(cells):
                    t[int(c)] += 1
                    oc.bloom[int(c)] += 1
                what = f"step {step}: outside write"
            else:
                if rng.integers(0, 3):
                    continue
                cbf.clear()
                oc = oracle.OracleCBF(m, k)
                live, what = [], f"step {step}: clear"
            if rng.integers(0, 2):
                table_ok(what)
        table_ok("end")
        assert cbf.batch_diagnostics()["saturated"] == 0
    finally:
        for kx, v in old.items():
            N.set_option(kx, v)
