"""Two handles driven CONCURRENTLY by two host threads on one device (include/psk.h "Threading": a handle is externally
synchronised, different handles are independent; errors and per-sketch options are thread-local / per handle).

Thread A owns a BloomFilter + its own HIP stream and runs insert / lookup rounds (bloom.py:234-272) with its own per-sketch
options; thread B owns a CountingBloomFilter + another stream and runs BASELINE cfg 4's add / remove stream through the update
windows (countingbloom.py:135-208) with different per-sketch options.  Both go on at the same time, many rounds, and every table
and every answer is compared with the sequential oracle; a third case provokes errors on both threads and checks that each
thread reads its OWN message from psk_last_error."""

import threading

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


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _run_threads(fns):
    errs = []

    def wrap(f):
        def g():
            try:
                f()
            except BaseException as e:  # noqa: BLE001
                import traceback

                errs.append(traceback.format_exc() + repr(e))
        return g

    ts = [threading.Thread(target=wrap(f)) for f in fns]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not any(t.is_alive() for t in ts), "a worker thread hangs"
    assert not errs, "\n".join(errs)


def test_two_handles_two_threads_two_streams_bit_exact(pa, oracle):
    from pyprobables_amd import _native as N

    rounds = 6
    n = 400_000
    bkeys = [oracle.gen_keys16(1000 + r, n) for r in range(rounds)]
    fresh = oracle.gen_keys16(77_000_000, n)
    B, nb = 150_000, 8
    ckeys = oracle.gen_keys16(5, B * nb)

    # the oracle's answers, computed up front on this thread
    ob = oracle.OracleBloom(2**28, 7)
    want_tables, want_hits, want_fresh = [], [], []
    for r in range(rounds):
        ob.add_keys(bkeys[r])
        want_tables.append(ob.bloom.copy())
        want_hits.append(ob.check_keys(bkeys[r]))
        want_fresh.append(ob.check_keys(fresh))
    start = threading.Barrier(2)
    got = {}

    def bloom_worker():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
            blm.set_engine_option("partition_min_keys", 1)       # everything through the partitioned kernels
            blm.set_engine_option("bloom_lookup", 3)             # this sketch: tile-flag lookups
            dk = [_dev(k) for k in bkeys]
            df = _dev(fresh)
            start.wait()
            out = []
            for r in range(rounds):
                blm.add_many(dk[r])
                hits = blm.check_many(dk[r])
                fr = blm.check_many(df)
                out.append((np.frombuffer(bytes(blm.bloom), dtype=np.uint8).copy(), hits.cpu().numpy().astype(np.uint8), fr.cpu().numpy().astype(np.uint8)))
            assert blm.get_engine_option("bloom_lookup") == 3 and blm.get_engine_option("partition_min_keys") == 1
            got["bloom"] = out

    def cbf_worker():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            cbf = pa.CountingBloomFilter(est_elements=7_100_000, false_positive_rate=0.01)  # ~6.8e7 counters: 4-bit slice images, update windows
            cbf.set_engine_option("update_window", 1)
            cbf.set_engine_option("remove_exact", 1)
            cbf.set_engine_option("partition_min_keys", 65536)   # differs from thread A's
            dk = _dev(ckeys)
            start.wait()
            out = []
            for rep in range(rounds):
                for b in range(nb):
                    cbf.add_many(dk[b * B:(b + 1) * B])
                    if b >= 1:
                        cbf.remove_many(dk[(b - 1) * B:(b - 1) * B + B // 2])
                out.append((cbf.table_tensor.cpu().numpy().view(np.uint32)[: cbf.number_bits].copy(), cbf.elements_added))
            assert cbf.get_engine_option("partition_min_keys") == 65536
            got["cbf"] = (out, cbf.number_bits)

    _run_threads([bloom_worker, cbf_worker])
    for r in range(rounds):
        tab, hits, fr = got["bloom"][r]
        assert np.array_equal(tab, want_tables[r]), f"bloom table, round {r}"
        assert np.array_equal(hits, want_hits[r]) and np.array_equal(fr, want_fresh[r]), f"bloom answers, round {r}"
    out, m = got["cbf"]
    oc = oracle.OracleCBF(m, 7)
    for rep in range(rounds):
        for b in range(nb):
            oc.update_keys(ckeys[b * B:(b + 1) * B])
            if b >= 1:
                oc.update_keys(ckeys[(b - 1) * B:(b - 1) * B + B // 2], -np.ones(B // 2, dtype=np.int64))
        assert np.array_equal(out[rep][0], oc.bloom), f"cbf table, repetition {rep}"
        assert out[rep][1] == int(oc.els_added)
    # the process-wide defaults were not touched by the per-sketch overrides
    assert N.get_option("bloom_lookup") == 2


def test_last_error_is_per_thread(pa):
    """psk_last_error is thread-local: two threads provoke DIFFERENT errors at the same time, each reads its own message"""
    import ctypes as C

    from pyprobables_amd import _native as N

    L = N.lib()
    seen = {}
    gate = threading.Barrier(2)

    def worker(tag, bad_call, needle):
        def run():
            for _ in range(200):
                gate.wait()
                rc = bad_call()
                assert rc != 0
                msg = N.last_error()
                assert needle in msg, (tag, msg)
            seen[tag] = True
        return run

    h = C.c_void_p()
    a = worker("create", lambda: L.psk_bloom_create(0, 7, 0, None, C.byref(h)), "table dimensions must be > 0")
    b = worker("option", lambda: L.psk_set_option(b"no_such_option_xyz", 1), "unknown option no_such_option_xyz")
    _run_threads([a, b])
    assert seen == {"create": True, "option": True}
