"""Variable-length keys -- the reference's native key type (``fnv_1a`` walks ``list(key)`` / ``map(ord, key)``, hashes.py:98) -- through
the windowed loaders (psk_device.hpp ``walk_key_bytes`` / ``walk_key_elems``): ragged batches handed over as ``(blob, offsets)`` pairs on
the device and on the host, at every start alignment, with empty keys, keys at the very end of the blob, blobs shorter than one window,
on the direct AND the partitioned kernels, bit-exact against the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pyprobables_amd

    return pyprobables_amd


@pytest.fixture()
def part(request):
    """direct kernels / partitioned kernels"""
    from pyprobables_amd import _native as N

    names = ("partition", "partition_min_keys")
    old = [N.get_option(k) for k in names]
    N.set_option("partition", 1 if request.param else 0)
    N.set_option("partition_min_keys", 1)
    yield request.param
    for k, v in zip(names, old):
        N.set_option(k, v)


def _ragged(rng, n, lo, hi, lead=0):
    lens = rng.integers(lo, hi + 1, size=n)
    lens[rng.integers(0, n, size=max(1, n // 50))] = 0  # empty keys in between
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    blob = rng.integers(0, 256, size=int(offs[-1]) + lead, dtype=np.uint8)
    return blob, offs + lead


def _keys(blob, offs):
    raw = blob.tobytes()
    return [raw[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]


def _table(f, dtype=np.uint8):
    return np.frombuffer(bytes(f.bloom), dtype=dtype)


@pytest.mark.parametrize("part", [False, True], indirect=True)
@pytest.mark.parametrize("est,fpr", [(400_000, 0.01), (28005615 // 16, 0.01), (400_000, 0.1), (2**22 // 8, 0.02)])  # Barrett / power of two (32-bit chains); k = 3 / 6: the round-up kernel
@pytest.mark.parametrize("lead", [0, 1, 2, 3])
def test_bloom_device_ragged_pairs_vs_oracle(pa, oracle, part, est, fpr, lead):
    rng = np.random.default_rng(100 + lead)
    n = 50_000
    blob, offs = _ragged(rng, n, 0, 70, lead)
    keys = _keys(blob, offs)
    # the blob tensor is EXACTLY as long as the keys need: the last key ends at the end of the allocation's payload
    dblob = torch.from_numpy(blob).cuda()
    doffs = torch.from_numpy(offs).cuda()
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    half = n // 2
    blm.add_many((dblob, doffs[: half + 1]))
    ob.add_varlen(keys[:half])
    assert np.array_equal(_table(blm), ob.bloom)
    assert blm.elements_added == half
    got = blm.check_many((dblob, doffs)).cpu().numpy().astype(np.uint8)
    assert np.array_equal(got, ob.check_varlen(keys))
    # the same pair on the host, and the list of bytes it stands for
    assert np.array_equal(np.asarray(blm.check_many((blob, offs))).astype(np.uint8), got)
    assert np.array_equal(np.asarray(blm.check_many(keys)).astype(np.uint8), got)


@pytest.mark.parametrize("part", [False, True], indirect=True)
def test_tiny_blobs_and_single_keys(pa, oracle, part):
    """blobs shorter than one 16-byte window, a blob of one byte, all keys empty"""
    for raw_keys in ([b"a"], [b"", b"xy", b""], [b"0123456789abcde"], [b"", b"", b""], [b"abc", b"defgh", b"ijklmno"]):
        blob = np.frombuffer(b"".join(raw_keys), dtype=np.uint8).copy()
        offs = np.zeros(len(raw_keys) + 1, dtype=np.int64)
        np.cumsum([len(k) for k in raw_keys], out=offs[1:])
        blm = pa.BloomFilter(est_elements=1000, false_positive_rate=0.01)
        ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
        dblob = torch.from_numpy(blob).cuda() if blob.size else torch.zeros(0, dtype=torch.uint8, device="cuda")
        blm.add_many((dblob, torch.from_numpy(offs).cuda()))
        ob.add_varlen(raw_keys)
        assert np.array_equal(_table(blm), ob.bloom), raw_keys
        assert bool(blm.check_many((dblob, torch.from_numpy(offs).cuda())).all())


@pytest.mark.parametrize("part", [False, True], indirect=True)
@pytest.mark.parametrize("L", [1, 3, 5, 13, 17, 31, 33])
def test_fixed_odd_lengths_vs_oracle(pa, oracle, part, L):
    """fixed-length keys that are not whole dwords share the windowed walk (KeysFixed<false>)"""
    rng = np.random.default_rng(L)
    n = 40_000
    keys = rng.integers(0, 256, size=(n, L), dtype=np.uint8)
    blm = pa.BloomFilter(est_elements=300_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    blm.add_many(torch.from_numpy(keys[: n // 2]).cuda())
    ob.add_keys(keys[: n // 2])
    assert np.array_equal(_table(blm), ob.bloom)
    assert np.array_equal(blm.check_many(torch.from_numpy(keys).cuda()).cpu().numpy().astype(np.uint8), ob.check_keys(keys))


@pytest.mark.parametrize("part", [False, True], indirect=True)
def test_code_point_pairs_vs_oracle(pa, oracle, part):
    """str keys with code points above 255 as a (code points, offsets) pair: whole code points are XORed in (hashes.py:98)"""
    rng = np.random.default_rng(9)
    n = 3_000
    lens = rng.integers(0, 23, size=n)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    cps = rng.integers(1, 0x2FFF, size=int(offs[-1]), dtype=np.uint32)
    cps[(cps >= 0xD800) & (cps <= 0xDFFF)] = 0x20AC  # no surrogates
    words = ["".join(map(chr, cps[offs[i]:offs[i + 1]])) for i in range(n)]
    blm = pa.BloomFilter(est_elements=50_000, false_positive_rate=0.01)
    ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
    dpair = (torch.from_numpy(cps.view(np.int32)).cuda(), torch.from_numpy(offs).cuda())
    blm.add_many((dpair[0], dpair[1][: n // 2 + 1]))
    ob.add_hashes(np.array([oracle.default_fnv_1a(w, blm.number_hashes) for w in words[: n // 2]], dtype=np.uint64))
    assert np.array_equal(_table(blm), ob.bloom)
    got = blm.check_many(dpair).cpu().numpy()
    assert np.array_equal(got, np.asarray(blm.check_many(words)))
    assert bool(got[: n // 2].all())


@pytest.mark.parametrize("part", [False, True], indirect=True)
def test_counters_take_ragged_pairs(pa, oracle, part):
    rng = np.random.default_rng(21)
    n = 60_000
    blob, offs = _ragged(rng, n, 1, 40)
    keys = _keys(blob, offs)
    pair = (torch.from_numpy(blob).cuda(), torch.from_numpy(offs).cuda())
    cms = pa.CountMinSketch(width=2**16, depth=5)
    ref = pa.CountMinSketch(width=2**16, depth=5)
    w = rng.integers(1, 9, size=n).astype(np.int32)
    cms.add_many(pair, torch.from_numpy(w).cuda())
    ref.add_many(keys, w)
    assert torch.equal(cms.table_tensor, ref.table_tensor)
    assert np.array_equal(cms.check_many(pair).cpu().numpy(), np.asarray(ref.check_many(keys)))
    hs = np.array([oracle.default_fnv_1a(k, 5) for k in keys[:500]], dtype=np.uint64)
    assert np.array_equal(np.asarray(ref.check_alt_many(hs)), cms.check_many(pair)[:500].cpu().numpy())
    cbf = pa.CountingBloomFilter(est_elements=100_000, false_positive_rate=0.01)
    cbf.add_many(pair)
    cbf.remove_many((pair[0], pair[1][: n // 3 + 1]))
    oc = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    hk = np.array([oracle.default_fnv_1a(k, cbf.number_hashes) for k in keys[:2000]], dtype=np.uint64)
    got = cbf.check_many(pair).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, np.asarray(cbf.check_many(keys)).view(np.uint32))
    assert np.array_equal(np.asarray(cbf.check_alt_many(hk)).view(np.uint32), got[:2000])
    del oc


def test_pair_validation(pa):
    blob = torch.zeros(10, dtype=torch.uint8, device="cuda")
    with pytest.raises(TypeError):
        pa.BloomFilter(est_elements=100, false_positive_rate=0.1).add_many((blob, torch.zeros(3, dtype=torch.int32, device="cuda")))
    with pytest.raises(TypeError):
        pa.BloomFilter(est_elements=100, false_positive_rate=0.1).add_many((blob, torch.zeros(3, dtype=torch.int64)))
    with pytest.raises(ValueError):
        pa.BloomFilter(est_elements=100, false_positive_rate=0.1).add_many((np.zeros(4, dtype=np.uint8), np.array([0, 3, 9], dtype=np.int64)))
    with pytest.raises(ValueError):
        pa.BloomFilter(est_elements=100, false_positive_rate=0.1).add_many((np.zeros(4, dtype=np.uint8), np.array([0, 3, 2], dtype=np.int64)))


@pytest.mark.parametrize("part", [False, True], indirect=True)
@pytest.mark.parametrize("est,fpr", [(300_000, 0.01), (28005615 // 16, 0.01), (200_000, 0.00001), (300_000, 0.1), (300_000, 0.02)])  # Barrett / power of two / k = 17 / k = 3, 6: the round-up kernel
def test_eight_byte_keys_fast_layout_vs_oracle(pa, oracle, part, est, fpr):
    """64-bit ids: the 8-byte fast layout (KeysFixed8: one dwordx2 per lane, exact k) when the batch is 8-byte aligned, the generic dword
    walk when it is not -- same tables, same answers"""
    rng = np.random.default_rng(8)
    n = 80_000
    keys = rng.integers(0, 256, size=(n, 8), dtype=np.uint8)
    flat = torch.zeros(n * 8 + 4, dtype=torch.uint8, device="cuda")
    flat[4:].copy_(torch.from_numpy(keys).cuda().reshape(-1))
    off4 = flat[4:].view(n, 8)  # 4 bytes off an 8-byte boundary
    for dk in (torch.from_numpy(keys).cuda(), off4):
        blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
        ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
        blm.add_many(dk[: n // 2])
        ob.add_keys(keys[: n // 2])
        assert np.array_equal(_table(blm), ob.bloom)
        assert np.array_equal(blm.check_many(dk).cpu().numpy().astype(np.uint8), ob.check_keys(keys))
    cms = pa.CountMinSketch(width=2**16, depth=5)
    ref = pa.CountMinSketch(width=2**16, depth=5)
    w = rng.integers(1, 9, size=n).astype(np.int32)
    cms.add_many(torch.from_numpy(keys).cuda(), torch.from_numpy(w).cuda())
    ref.add_many(off4, torch.from_numpy(w).cuda())
    assert torch.equal(cms.table_tensor, ref.table_tensor)
    oc = oracle.OracleCMS(2**16, 5)
    oc.add_keys(keys, w)
    assert np.array_equal(np.frombuffer(bytes(cms._bins), dtype=np.int32), oc.bins)
    assert np.array_equal(cms.check_many(torch.from_numpy(keys).cuda()).cpu().numpy(), oc.check_keys(keys).astype(np.int32))


def test_batch_order_option_gives_the_same_table(pa, oracle):
    """option ragged_sort = 0 (pass 1 takes ragged keys in batch order instead of sorting each tile by length): same table, same answers"""
    from pyprobables_amd import _native as N

    rng = np.random.default_rng(77)
    n = 300_000
    blob, offs = _ragged(rng, n, 0, 60)
    keys = _keys(blob, offs)
    pair = (torch.from_numpy(blob).cuda(), torch.from_numpy(offs).cuda())
    old = [N.get_option(k) for k in ("ragged_sort", "partition_min_keys")]
    try:
        N.set_option("partition_min_keys", 1)
        tables = []
        for srt in (1, 0):
            N.set_option("ragged_sort", srt)
            blm = pa.BloomFilter(est_elements=2_000_000, false_positive_rate=0.01)
            blm.add_many((pair[0], pair[1][: n // 2 + 1]))
            tables.append((_table(blm).copy(), blm.check_many(pair).cpu().numpy()))
        ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
        ob.add_varlen(keys[: n // 2])
        for tab, got in tables:
            assert np.array_equal(tab, ob.bloom)
            assert np.array_equal(got.astype(np.uint8), ob.check_varlen(keys))
    finally:
        N.set_option("ragged_sort", old[0])
        N.set_option("partition_min_keys", old[1])


@pytest.mark.parametrize("part", [False, True], indirect=True)
@pytest.mark.parametrize("est,fpr", [(300_000, 0.01), (28005615 // 16, 0.01), (300_000, 0.1)])
def test_thirty_two_byte_keys_fast_layout_vs_oracle(pa, oracle, part, est, fpr):
    """digest-sized keys: the 32-byte fast layout (KeysFixed32: two dwordx4 per lane) when the batch is 16-byte aligned, the generic dword walk
    when it is not"""
    rng = np.random.default_rng(32)
    n = 70_000
    keys = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    flat = torch.zeros(n * 32 + 4, dtype=torch.uint8, device="cuda")
    flat[4:].copy_(torch.from_numpy(keys).cuda().reshape(-1))
    for dk in (torch.from_numpy(keys).cuda(), flat[4:].view(n, 32)):
        blm = pa.BloomFilter(est_elements=est, false_positive_rate=fpr)
        ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
        blm.add_many(dk[: n // 2])
        ob.add_keys(keys[: n // 2])
        assert np.array_equal(_table(blm), ob.bloom)
        assert np.array_equal(blm.check_many(dk).cpu().numpy().astype(np.uint8), ob.check_keys(keys))
