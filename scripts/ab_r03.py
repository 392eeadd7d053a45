"""round-3 A/Bs on one MI355X: Bloom lookup schemes by the share of absent keys; CBF 1 GiB table: nibble-delta updates and image layouts,
validated removes, nontemporal table loads of the lookups' pass 2"""
import sys, torch
sys.path.insert(0, "/root/repo")
import pyprobables_amd as pa
from pyprobables_amd import _native as N
def gen(n, start):
    t = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    N.check(N.lib().psk_gen_keys16(t.data_ptr(), start, n, 0x5EED, 0, torch.cuda.current_stream().cuda_stream or None))
    return t
def tl(fn, iters=5, warm=3):
    for _ in range(warm):
        fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
n = 10_000_000
keys, fresh = gen(n, 0), gen(n, 10 * n)
mixed = torch.cat([keys[: n // 2], fresh[: n // 2]])
q25 = torch.cat([keys[: 3 * n // 4], fresh[: n // 4]])
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
blm.add_many(keys)
# (round 3 also ran this loop with mode 3 = the two-stage cascade; the cascade was removed after it lost everywhere: DESIGN.md 3.3)
for mode, name in ((0, "keyed"), (1, "return trip"), (2, "auto")):
    N.set_option("bloom_lookup", mode)
    print(f"bloom check {name:12s}: all-hit {tl(lambda: blm.check_many(keys)):7.1f} us  25%-fresh {tl(lambda: blm.check_many(q25)):7.1f}  half-fresh {tl(lambda: blm.check_many(mixed)):7.1f}  all-fresh {tl(lambda: blm.check_many(fresh)):7.1f}", flush=True)
N.set_option("bloom_lookup", 2)
cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
for lay in (0, 1):
    N.set_option("nibble_update_layout", lay)
    for upd in (1, 0):
        N.set_option("update_nibble_slices", upd)
        cbf.clear()
        t_add = tl(lambda: cbf.add_many(keys), 4, 2)
        def addrem():
            cbf.add_many(keys); cbf.remove_many(keys)
        t_ar = tl(addrem, 3, 1)
        print(f"cbf 1GiB layout {lay} nibble-updates {upd}: add {t_add:7.1f} us, add+validated remove {t_ar:7.1f} us -> remove {t_ar - t_add:7.1f}", flush=True)
N.set_option("update_nibble_slices", 1)
N.set_option("nibble_update_layout", 1)
for nt in (0, 1):
    N.set_option("nibble_nt_loads", nt)
    cbf.clear()
    t_add = tl(lambda: cbf.add_many(keys), 4, 2)
    print(f"nt loads {nt}: cbf check 10M: {tl(lambda: cbf.check_many(keys)):7.1f} us   add {t_add:7.1f} us")
N.set_option("nibble_nt_loads", 0)
