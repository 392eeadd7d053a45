"""ExpandingBloomFilter / RotatingBloomFilter on the HIP engine: exact sequential semantics for ordered batches.

Checked against (a) the reference's own KATs, (b) fixtures produced by the real reference
(tests/golden/golden_stack.json) and (c) the C oracle on seeded streams -- byte-for-byte on the export image."""

import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from _util import unpackbits

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
GS = json.loads((Path(__file__).parent / "golden" / "golden_stack.json").read_text())


@pytest.fixture(scope="module")
def pa():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pyprobables_amd

    return pyprobables_amd


def _stream(oracle, n, pool, salt):
    return [int(oracle.splitmix64(salt * 1000003 + j) % pool) for j in range(n)]


def _state(blm):
    raw = bytes(blm)
    return {"filters": len(blm._blooms), "counts": [b.elements_added for b in blm._blooms], "elements_added": blm.elements_added,
            "sha256": hashlib.sha256(raw).hexdigest(), "nbytes": len(raw)}


def _make(pa, g):
    if "max_queue_size" in g["kw"]:
        return pa.RotatingBloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"], max_queue_size=g["kw"]["max_queue_size"])
    return pa.ExpandingBloomFilter(est_elements=g["est_elements"], false_positive_rate=g["fpr"])


def test_reference_kats(pa):
    blm = pa.ExpandingBloomFilter(est_elements=25, false_positive_rate=0.05)
    assert hashlib.md5(bytes(blm)).hexdigest() == "eb5769ae9babdf7b37d6ce64d58812bc"  # expandingbloom_test.py:104
    assert (blm.expansions, blm.false_positive_rate, blm.estimated_elements, blm.elements_added) == (0, 0.05, 25, 0)
    blm = pa.ExpandingBloomFilter(est_elements=10, false_positive_rate=0.05)
    for i in range(100):
        blm.add(f"{i}", True)
    assert blm.expansions == 9  # :33-38
    blm = pa.ExpandingBloomFilter(est_elements=10, false_positive_rate=0.05)
    for i in range(120):
        blm.add(f"{i}")
    assert (blm.expansions, blm.elements_added) == (8, 120)  # :47-54
    blm = pa.ExpandingBloomFilter(est_elements=30, false_positive_rate=0.05)
    for i in range(100):
        blm.add(f"{i}")
    blm.add("this is a test")
    blm.add("this is another test")
    assert blm.expansions > 1 and blm.elements_added == 102  # :56-69
    assert blm.check("this is a test") and "this is another test" in blm
    assert not blm.check("this is yet another test!") and "this is not another test" not in blm
    blm = pa.ExpandingBloomFilter(est_elements=25, false_positive_rate=0.05)
    for _ in range(3):
        blm.push()
    assert (blm.expansions, blm.elements_added) == (3, 0)  # :85-97


def test_frombytes_and_file_roundtrip(pa, tmp_path):
    blm = pa.ExpandingBloomFilter(est_elements=25, false_positive_rate=0.05)
    for i in range(105):
        blm.add(str(i))
    raw = bytes(blm)
    assert raw.hex() == GS["kat_frombytes"]["hex"]
    blm2 = pa.ExpandingBloomFilter.frombytes(raw)
    assert (blm2.expansions, blm2.false_positive_rate, blm2.estimated_elements, blm2.elements_added) == (3, 0.05000000074505806, 25, 105)
    assert bytes(blm2) == raw
    assert all(blm2.check(str(i)) for i in range(105))
    path = tmp_path / "stack.ebf"
    blm.export(path)
    blm3 = pa.ExpandingBloomFilter(filepath=path)
    assert bytes(blm3) == raw and blm3.expansions == 3
    rbf = pa.RotatingBloomFilter.frombytes(raw, max_queue_size=5)
    assert rbf.current_queue_size == 4 and rbf.max_queue_size == 5 and bytes(rbf) == raw


@pytest.mark.parametrize("mode", ["one_batch", "ragged_batches", "per_key"])
@pytest.mark.parametrize("name", ["ebf_small", "ebf_force", "ebf_highfpr", "rbf_small", "rbf_highfpr"])
def test_golden_string_streams(pa, oracle, name, mode):
    g = GS[name]
    if mode == "per_key" and g["n"] > 500:
        pytest.skip("per-key loop only on the small streams")
    blm = _make(pa, g)
    seq = _stream(oracle, g["n"], g["pool"], g["salt"])
    rng = np.random.default_rng(7)
    done = 0
    for upto in sorted(int(x) for x in g["snapshots"]):
        part = [f"k{i}" for i in seq[done:upto]]
        if mode == "one_batch":
            blm.add_many(part, force=g["force"])
        elif mode == "per_key":
            for key in part:
                blm.add(key, g["force"])
        else:
            p = 0
            while p < len(part):
                step = int(rng.integers(1, 97))
                blm.add_many(part[p:p + step], force=g["force"])
                p += step
        done = upto
        assert _state(blm) == g["snapshots"][str(upto)]
    assert bytes(blm).hex() == g["hex"]
    got = blm.check_many([f"k{i}" for i in g["probes"]])
    assert np.array_equal(np.asarray(got, dtype=np.uint8), unpackbits(g["membership_bits"], len(g["probes"])))


@pytest.mark.parametrize("name", ["ebf_synth16", "rbf_synth16"])
def test_golden_device_keys(pa, oracle, name):
    g = GS[name]
    blm = _make(pa, g)
    seq = np.asarray(_stream(oracle, g["n"], g["pool"], g["salt"]))
    pool = oracle.gen_keys16(0, g["pool"] + 2000)
    keys = torch.from_numpy(np.ascontiguousarray(pool[seq])).cuda()
    blm.add_many(keys)  # 30 000 ordered keys, one call, device resident
    assert _state(blm) == g["final"]
    probes = np.arange(0, g["probe_stop"], g["probe_step"])
    got = blm.check_many(torch.from_numpy(np.ascontiguousarray(pool[probes])).cuda()).cpu().numpy()
    assert np.array_equal(got.astype(np.uint8), unpackbits(g["membership_bits"], len(probes)))


def test_push_pop_log(pa):
    g = GS["rbf_push_pop"]
    rbf = pa.RotatingBloomFilter(est_elements=10, false_positive_rate=0.05, max_queue_size=3)
    for step, rec in enumerate(g["log"]):
        if rec["op"] == "add":
            rbf.add_many([f"s{step}-{i}" for i in range(7)])
        elif rec["op"] == "push":
            rbf.push()
        else:
            rbf.pop()
        assert (rbf.current_queue_size, [b.elements_added for b in rbf._blooms], rbf.elements_added) == (rec["queue"], rec["counts"], rec["elements_added"])
    assert bytes(rbf).hex() == g["hex"]
    one = pa.RotatingBloomFilter(est_elements=10, false_positive_rate=0.05)
    with pytest.raises(pa.RotatingBloomFilterError, match="unusable system"):
        one.pop()


@pytest.mark.parametrize("seed", range(10))
def test_differential_vs_oracle(pa, oracle, seed):
    rng = np.random.default_rng(4000 + seed)
    est = int(rng.choice([40, 300, 2500, 20_000]))
    fpr = float(rng.choice([0.4, 0.2, 0.05, 0.01]))
    queue = int(rng.choice([0, 0, 2, 5]))
    pool = int(est * rng.choice([0.8, 3, 12]))
    n = int(min(est * rng.choice([2, 9]), 120_000))
    if queue:
        blm = pa.RotatingBloomFilter(est_elements=est, false_positive_rate=fpr, max_queue_size=queue)
    else:
        blm = pa.ExpandingBloomFilter(est_elements=est, false_positive_rate=fpr)
    st = oracle.OracleStack(est, fpr, queue=queue, max_filters=256)
    keys16 = oracle.gen_keys16(seed * 1_000_000, pool)
    p = 0
    while p < n:
        step = int(rng.choice([1, 13, 700, 9000, 60_000]))
        step = min(step, n - p)
        sel = rng.integers(0, pool, size=step)
        sel[step // 2:] = sel[: step - step // 2]  # repeats inside the batch
        force = bool(rng.integers(0, 8) == 0)
        batch = np.ascontiguousarray(keys16[sel])
        if rng.integers(0, 2):
            blm.add_many(torch.from_numpy(batch).cuda(), force=force)
        else:
            blm.add_many(batch, force=force)
        st.add_keys([bytes(r) for r in batch], force=force)
        p += step
        assert blm.elements_added == st.els_added
        assert [b.elements_added for b in blm._blooms] == [int(c) for c in st.counts[: st.nfilters]]
    assert bytes(blm) == st.export_bytes()
    probe = oracle.gen_keys16(seed * 1_000_000, pool + 500)
    assert np.array_equal(blm.check_many(probe).astype(np.uint8), st.check_keys([bytes(r) for r in probe]))
    print(f"seed {seed}: est={est} fpr={fpr} queue={queue} n={n} filters={st.nfilters}")


def test_multi_window_batch(pa, oracle):
    """one add_many larger than the hashing window (2^22 keys): the windows must chain like one ordered stream"""
    n, pool = 5_000_000, 3_000_000
    est, fpr = 1_200_000, 0.02
    rng = np.random.default_rng(12)
    sel = rng.integers(0, pool, size=n)
    keys16 = oracle.gen_keys16(9_000_000, pool)
    batch = np.ascontiguousarray(keys16[sel])
    blm = pa.ExpandingBloomFilter(est_elements=est, false_positive_rate=fpr)
    blm.add_many(torch.from_numpy(batch).cuda())
    st = oracle.OracleStack(est, fpr, max_filters=16)
    blob = np.ascontiguousarray(batch).reshape(-1)
    offs = np.arange(0, 16 * (n + 1), 16, dtype=np.uint64)
    import ctypes as C

    rc = oracle.lib().psk_o_stack_add_varlen(st.stack.ctypes.data, st.filter_bytes, st.max_filters, C.byref(st._n), st.counts.ctypes.data,
                                             st.m, st.k, st.est, st.queue, blob.ctypes.data, offs.ctypes.data, n, 0, C.byref(st._added))
    assert rc == 0
    assert [b.elements_added for b in blm._blooms] == [int(c) for c in st.counts[: st.nfilters]]
    assert bytes(blm) == st.export_bytes()


def test_custom_hash_function_route(pa):
    """a user callable runs per key on the host and enters through the pre-hashed layout: same stack as the fused family
    when it computes the same hashes (expandingbloom_test.py:40-45 style), md5 family through the digest kernel"""
    def my_hash(key, depth=1):
        return pa.default_fnv_1a(key, depth)

    keys = [f"{i}" for i in range(400)] + ["é", ""]
    a = pa.ExpandingBloomFilter(est_elements=30, false_positive_rate=0.05)
    b = pa.ExpandingBloomFilter(est_elements=30, false_positive_rate=0.05, hash_function=my_hash)
    a.add_many(keys)
    for i in range(0, len(keys), 37):
        b.add_many(keys[i:i + 37])
    assert bytes(a) == bytes(b) and a.expansions == b.expansions > 3
    probes = [f"p{i}" for i in range(200)] + keys[:50]
    assert list(a.check_many(probes)) == list(b.check_many(probes)) and b.check("17") and b.hash_function is my_hash
    b.add_alt(my_hash("brand new", b._k))
    assert b.check_alt(my_hash("brand new", b._k)) and b.elements_added == len(keys) + 1
    m = pa.ExpandingBloomFilter(est_elements=30, false_positive_rate=0.05, hash_function=pa.default_md5)
    m.add_many(keys)
    one = pa.ExpandingBloomFilter(est_elements=30, false_positive_rate=0.05, hash_function=pa.default_md5)
    for key in keys:
        one.add(key)
    assert bytes(m) == bytes(one) and all(m.check_many(keys))


def test_stack_of_2p26_bit_filters_with_bounded_scratch(pa, oracle):
    """filters of m ~ 2^26 bits (est 7 M at 1 %): the ordered insert resolves its chunks through a bounded hash map (32 MiB) instead of a
    uint32 per filter bit (round 3: 256 MiB here, 1 GiB at m = 2^28) -- same stack as the sequential oracle, repeats and one growth included"""
    est, fpr = 7_000_000, 0.01
    blm = pa.ExpandingBloomFilter(est_elements=est, false_positive_rate=fpr)
    assert abs(blm._blooms[-1].number_bits - 2**26) < 2**16  # (67 095 409 bits per filter)
    n, pool = 12_000_000, 12_000_000
    rng = np.random.default_rng(21)
    sel = rng.integers(0, pool, size=n)
    keys16 = oracle.gen_keys16(77_000_000, pool)
    batch = np.ascontiguousarray(keys16[sel])
    for w0 in range(0, n, 3_000_000):  # four ordered calls; the last one crosses the growth
        blm.add_many(torch.from_numpy(batch[w0:w0 + 3_000_000]).cuda())
    scratch = sum(t.numel() * t.element_size() for name, t in blm._scratch.items() if name.startswith("slots"))
    assert 0 < scratch <= 64 << 20
    st = oracle.OracleStack(est, fpr, max_filters=4)
    blob = batch.reshape(-1)
    offs = np.arange(0, 16 * (n + 1), 16, dtype=np.uint64)
    import ctypes as C

    rc = oracle.lib().psk_o_stack_add_varlen(st.stack.ctypes.data, st.filter_bytes, st.max_filters, C.byref(st._n), st.counts.ctypes.data,
                                             st.m, st.k, st.est, st.queue, blob.ctypes.data, offs.ctypes.data, n, 0, C.byref(st._added))
    assert rc == 0 and st.nfilters == 2
    assert [b.elements_added for b in blm._blooms] == [int(c) for c in st.counts[: st.nfilters]]
    assert bytes(blm) == st.export_bytes()
