"""BASELINE.json configurations at (or near) full size on one MI355X, bit-exact against the C oracle where the
oracle finishes in seconds, plus size-independent properties over the whole stream.

cfg 2  BloomFilter m = 2^28, k = 7: insert 10M keys, check them + 10M fresh keys       (full compare)
cfg 3  CountMinSketch 2^20 x 5: 100M weighted adds                                       (full compare)
cfg 4  CountingBloomFilter m = 2^28 (1 GiB): 50M-op add/remove stream in 1M batches      (full compare after every batch)
cfg 5  BloomFilter m = 2^31: two rank-shards merged by OR == single-stream filter        (20M keys compared; the collective
       form -- all_to_all + OR kernel + all_gather with two ranks -- is tests/test_gpu_bench_multi.py, and over real RCCL tests/test_gpu_rccl_multi.py and examples/psk_merge_demo.c)
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SEED = 0x5EED


@pytest.fixture(scope="module")
def pa():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pyprobables_amd

    return pyprobables_amd


def dev_keys(start, n):
    from pyprobables_amd import _native as N

    t = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    N.check(N.lib().psk_gen_keys16(t.data_ptr(), start, n, SEED, 0, torch.cuda.current_stream().cuda_stream or None))
    return t


def dev_weights(start, n):
    from pyprobables_amd import _native as N

    t = torch.empty(n, dtype=torch.int32, device="cuda")
    N.check(N.lib().psk_gen_weights(t.data_ptr(), start, n, SEED, 0, torch.cuda.current_stream().cuda_stream or None))
    return t


def test_device_generators_match_oracle(pa, oracle, golden):
    k = dev_keys(123_456, 100_000).cpu().numpy()
    assert np.array_equal(k, oracle.gen_keys16(123_456, 100_000))
    assert bytes(dev_keys(0, 1).cpu().numpy()[0]).hex() == golden["keygen"]["key0"]
    w = dev_weights(77, 100_000).cpu().numpy()
    assert np.array_equal(w, oracle.gen_weights(77, 100_000))


def test_cfg2_bloom_2p28_10M_full(pa, oracle):
    n = 10_000_000
    blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
    assert (blm.number_bits, blm.number_hashes) == (2**28, 7)
    keys = dev_keys(0, n)
    blm.add_many(keys)
    ob = oracle.OracleBloom(2**28, 7)
    ob.add_keys(oracle.gen_keys16(0, n))
    tab = blm.table_tensor.cpu().numpy().view(np.uint8)[: ob.bloom.size]
    assert np.array_equal(tab, ob.bloom)                      # all 32 MiB, bit for bit
    assert blm.elements_added == n
    assert blm._cnt_number_bits_set() == ob.bits_set()
    assert bool(blm.check_many(keys).all())                   # every inserted key is found
    fresh = blm.check_many(dev_keys(n, n)).cpu().numpy().view(np.uint8)
    exp = ob.check_keys(oracle.gen_keys16(n, n))
    assert np.array_equal(fresh, exp)                         # 10M membership answers incl. the false positives
    assert 0 < int(exp.sum()) < n // 1000                     # fpr at 36 % load is far below the design 1 %
    t0 = blm.table_tensor.clone()
    blm.add_many(keys)                                        # idempotence at full size
    assert torch.equal(t0, blm.table_tensor)


def test_cfg3_cms_100M_weighted_full(pa, oracle):
    d, passes = 10_000_000, 10
    cms = pa.CountMinSketch(width=2**20, depth=5)
    oc = oracle.OracleCMS(2**20, 5)
    keys_h = oracle.gen_keys16(0, d)
    keys = dev_keys(0, d)
    for p in range(passes):  # update i uses key (i mod 10M) and weight w(i)
        cms.add_many(keys, dev_weights(p * d, d))
        oc.add_keys(keys_h, oracle.gen_weights(p * d, d))
    bins = cms.table_tensor.cpu().numpy()[: oc.bins.size]
    assert np.array_equal(bins, oc.bins)                       # all 5,242,880 counters
    assert cms.elements_added == oc.els_added
    assert int(bins.astype(np.int64).sum()) == 5 * oc.els_added  # every update lands once per row
    assert cms.batch_diagnostics()["saturated"] == 0
    got = cms.check_many(keys[:1_000_000]).cpu().numpy()
    assert np.array_equal(got, oc.check_keys(keys_h[:1_000_000]).astype(np.int32))


def test_cfg4_cbf_1GiB_mixed_stream(pa, oracle):
    """all 50 batches of the cfg-4 stream, the whole 1 GiB table compared with the oracle after EVERY batch (on the device:
    the oracle's table is uploaded, 1 GiB over PCIe per batch)"""
    from pyprobables_amd import _native as N

    B, nb = 1_000_000, 50
    cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
    assert (cbf.number_bits, cbf.number_hashes) == (2**28, 7)
    oc = oracle.OracleCBF(2**28, 7)
    for b in range(nb):
        cbf.add_many(dev_keys(b * B, B))
        oc.update_keys(oracle.gen_keys16(b * B, B))
        if b >= 1:
            cbf.remove_many(dev_keys((b - 1) * B, B // 2))   # each key removed at most once: well-formed stream
            oc.update_keys(oracle.gen_keys16((b - 1) * B, B // 2), -np.ones(B // 2, dtype=np.int64))
        want = torch.from_numpy(oc.bloom.view(np.int32)).cuda()
        assert torch.equal(cbf.table_tensor[: want.numel()], want), f"table differs after batch {b}"
        del want
        assert cbf.elements_added == oc.els_added
        # lookups after every batch (the 1 GiB table's partitioned lookup: nibble slices): this batch's keys -- present --, the keys
        # just removed -- mostly back to 0 -- and keys of the next batch -- absent
        starts = [max(b - 1, 0) * B, b * B, (b + 1) * B]
        got = cbf.check_many(torch.cat([dev_keys(s0, 100_000) for s0 in starts])).cpu().numpy().view(np.uint32)
        want_c = oc.check_keys(np.concatenate([oracle.gen_keys16(s0, 100_000) for s0 in starts]))
        assert np.array_equal(got, want_c), f"lookups differ after batch {b}"
        if b % 8 == 7:  # (0.3 M keys sit below the crossover of the nibble-slice lookup: force it now and then)
            N.set_option("lookup_nibble_slices", 2)
            try:
                got = cbf.check_many(torch.cat([dev_keys(s0, 100_000) for s0 in starts])).cpu().numpy().view(np.uint32)
            finally:
                N.set_option("lookup_nibble_slices", 1)
            assert np.array_equal(got, want_c), f"nibble-slice lookups differ after batch {b}"
    expect = nb * B - (nb - 1) * (B // 2)
    assert cbf.elements_added == expect
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}
    total = int(cbf.table_tensor.view(torch.int32).to(torch.int64).sum().item())
    assert total == 7 * expect                                 # k increments per live insert, nothing lost
    last = cbf.check_many(dev_keys((nb - 1) * B, B)).cpu().numpy().view(np.uint32)
    assert np.array_equal(last, oc.check_keys(oracle.gen_keys16((nb - 1) * B, B)))


def test_cfg4_cbf_1GiB_mixed_stream_write_combined(pa, oracle):
    """the same 50-batch stream with combine_updates=True (what bench.py --config cfg4 runs): 1M-key batches wait on the
    device and reach the 1 GiB table as a few large partitioned updates; table and elements_added compared with the oracle after
    every fifth batch, lookups in between"""
    B, nb = 1_000_000, 50
    cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01, combine_updates=True)
    oc = oracle.OracleCBF(2**28, 7)
    for b in range(nb):
        cbf.add_many(dev_keys(b * B, B))
        oc.update_keys(oracle.gen_keys16(b * B, B))
        if b >= 1:
            cbf.remove_many(dev_keys((b - 1) * B, B // 2))
            oc.update_keys(oracle.gen_keys16((b - 1) * B, B // 2), -np.ones(B // 2, dtype=np.int64))
        quiet = 20 <= b < 40           # one long window: 20 batches wait together (30 M keys in the two lists)
        if (b % 5 == 2 and not quiet) or b in (40, nb - 1):  # (every read flushes what waits)
            want = torch.from_numpy(oc.bloom.view(np.int32)).cuda()
            assert torch.equal(cbf.table_tensor[: want.numel()], want), f"table differs after batch {b}"
            del want
            assert cbf.elements_added == oc.els_added
        elif b % 5 == 0 and not quiet:  # a lookup in the middle of a window (it flushes too): present / just removed / absent keys
            starts = [max(b - 1, 0) * B, b * B, (b + 1) * B]
            got = cbf.check_many(torch.cat([dev_keys(s0, 100_000) for s0 in starts])).cpu().numpy().view(np.uint32)
            assert np.array_equal(got, oc.check_keys(np.concatenate([oracle.gen_keys16(s0, 100_000) for s0 in starts]))), f"lookups differ after batch {b}"
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}


def test_cfg4_cbf_1GiB_large_batches_take_the_two_level_path(pa, oracle):
    """8192 slices: one 10 M-key batch brings enough probes for coarse buckets -> k_part_split -> per-slice fold"""
    n = 10_000_000
    cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
    oc = oracle.OracleCBF(2**28, 7)
    w = (1 + (np.arange(n) % 3)).astype(np.uint32)
    cbf.add_many(dev_keys(0, n), torch.from_numpy(w.view(np.int32)).cuda())
    cbf.add_many(dev_keys(n // 2, n))                          # unit weights, overlapping keys
    oc.update_keys(oracle.gen_keys16(0, n), w.astype(np.int64))
    oc.update_keys(oracle.gen_keys16(n // 2, n), np.ones(n, dtype=np.int64))
    assert np.array_equal(cbf.table_tensor.cpu().numpy().view(np.uint32), oc.bloom)
    assert cbf.elements_added == oc.els_added
    assert cbf.batch_diagnostics() == {"violations": 0, "saturated": 0}


def test_bloom_4gbit_takes_the_two_level_path(pa, oracle):
    """m just below 2^32 bits: 4022 slices of 2^20 bits -> coarse buckets + k_part_split for inserts; lookups direct"""
    n = 3_000_000
    blm = pa.BloomFilter(est_elements=440_000_000, false_positive_rate=0.01)
    assert 2**31 < blm.number_bits < 2**32 and blm.number_hashes == 7
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    keys = oracle.gen_keys16(42, n)
    dk = torch.from_numpy(keys).cuda()
    blm.add_many(dk[: n // 2])
    blm.add_many(dk[n // 3:])
    ob.add_keys(keys)
    assert torch.equal(blm.table_tensor.cpu()[: ob.bloom.size // 4].view(torch.uint8), torch.from_numpy(ob.bloom[: ob.bloom.size // 4 * 4]))
    assert blm._cnt_number_bits_set() == ob.bits_set()
    probe = oracle.gen_keys16(42 + n - 100_000, 200_000)
    assert np.array_equal(blm.check_many(torch.from_numpy(probe).cuda()).cpu().numpy().astype(np.uint8), ob.check_keys(probe))


def test_cms_large_non_pow2_width_two_level(pa, oracle):
    """width 10^8 + 7 x depth 5 (2 GB of bins, 15259 slices: coarse buckets + k_part_split) with a non power-of-two modulus:
    all 5 x (10^8 + 7) bins against the oracle after 14 M weighted updates (13 M in one call: the two-level path is taken from
    cells / 8 probes per call; 1 M through the direct kernels), then lookups of inserted and fresh keys"""
    width, depth, n = 100_000_007, 5, 14_000_000
    cms = pa.CountMinSketch(width=width, depth=depth)
    oc = oracle.OracleCMS(width, depth)
    keys = oracle.gen_keys16(5, n)
    w = oracle.gen_weights(5, n)
    dk, dw = torch.from_numpy(keys).cuda(), torch.from_numpy(w).cuda()
    cms.add_many(dk[:13_000_000], dw[:13_000_000])
    cms.add_many(dk[13_000_000:], dw[13_000_000:])
    oc.add_keys(keys, w)
    assert np.array_equal(cms.table_tensor.cpu().numpy()[: oc.bins.size], oc.bins)
    assert cms.elements_added == oc.els_added
    probe = oracle.gen_keys16(5 + n - 250_000, 500_000)
    assert np.array_equal(cms.check_many(torch.from_numpy(probe).cuda()).cpu().numpy().astype(np.int64), oc.check_keys(probe))


def test_bloom_non_pow2_just_below_2p31_bits(pa, oracle):
    """the largest single-level geometry with a non power-of-two modulus (2039 slices of 2^20 bits): the 32-bit remainder of
    the partitioned path (reduce_small: r = h - q*m < 2m, just below 2^32 here) against the oracle's plain h % m"""
    n = 3_000_000
    blm = pa.BloomFilter(est_elements=223_000_000, false_positive_rate=0.01)
    m = blm.number_bits
    assert 2**31 - 2**24 < m < 2**31 and m & (m - 1) and blm.number_hashes == 7
    ob = oracle.OracleBloom(m, 7)
    keys = oracle.gen_keys16(99, n)
    dk = torch.from_numpy(keys).cuda()
    blm.add_many(dk[: n // 2])
    blm.add_many(dk[n // 3:])
    ob.add_keys(keys)
    nb = ob.bloom.size // 4 * 4
    assert torch.equal(blm.table_tensor.cpu().view(torch.uint8)[:nb], torch.from_numpy(ob.bloom[:nb]))
    assert blm._cnt_number_bits_set() == ob.bits_set()
    probe = oracle.gen_keys16(99 + n - 300_000, 600_000)     # half inserted, half fresh; large enough for the partitioned lookup
    assert np.array_equal(blm.check_many(torch.from_numpy(probe).cuda()).cpu().numpy().astype(np.uint8), ob.check_keys(probe))


def test_bloom_beyond_2p32_bits_uses_64bit_indices(pa, oracle):
    """m > 2^32: bit indices no longer fit 32 bits, the partitioned path steps aside, the direct kernels carry on"""
    n = 1_000_000
    blm = pa.BloomFilter(est_elements=500_000_000, false_positive_rate=0.01)
    assert blm.number_bits > 2**32
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    keys = oracle.gen_keys16(7, n)
    dk = torch.from_numpy(keys).cuda()
    blm.add_many(dk[: n // 2])
    ob.add_keys(keys[: n // 2])
    assert blm._cnt_number_bits_set() == ob.bits_set()
    assert np.array_equal(blm.check_many(dk).cpu().numpy().astype(np.uint8), ob.check_keys(keys))
    top = np.flatnonzero(ob.bloom[2**29:])[:5] + 2**29        # some set bytes above bit 2^32
    assert len(top) and np.array_equal(blm.table_tensor.view(torch.uint8)[torch.from_numpy(top).cuda()].cpu().numpy(), ob.bloom[top])


def test_cfg5_bloom_2p31_two_shards_or_merge(pa, oracle):
    from pyprobables_amd import _native as N

    n = 10_000_000  # per shard
    a = pa.BloomFilter(est_elements=224044920, false_positive_rate=0.01)
    b = pa.BloomFilter(est_elements=224044920, false_positive_rate=0.01)
    assert (a.number_bits, a.number_hashes) == (2**31, 7)
    a.add_many(dev_keys(0, n))       # "rank 0" replica
    b.add_many(dev_keys(n, n))       # "rank 1" replica
    t = a._tab
    N.check(N.lib().psk_table_or(t.ptr, b._tab.ptr, t.nwords, t.device, t.stream))   # local form of allreduce(OR)
    ob = oracle.OracleBloom(2**31, 7)
    ob.add_keys(oracle.gen_keys16(0, 2 * n))
    tab = a.table_tensor.cpu().numpy().view(np.uint8)[: ob.bloom.size]
    assert np.array_equal(tab, ob.bloom)                       # 256 MiB, merged == single-stream
    assert bool(a.check_many(dev_keys(n // 2, n)).all())
