"""round-5 A/B on one MI355X: Bloom lookup schemes (0 keyed probes, 1 return trip, 3 tile flags, 4 lazy gathers, 2 automatic) by the share of
absent keys, m = 2^28 (cfg 2; 10 M keys inserted = 23 % of the bits set, and 28 M = the design load, 52 %) and m = 2^31 (cfg 5's geometry);
microseconds per call of 10 M (2^25) keys"""
import sys
import torch
sys.path.insert(0, "/root/repo")
import pyprobables_amd as pa
from pyprobables_amd import _native as N


def gen(n, start):
    t = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    N.check(N.lib().psk_gen_keys16(t.data_ptr(), start, n, 0x5EED, 0, torch.cuda.current_stream().cuda_stream or None))
    return t


def tl(fn, iters=8, warm=4):
    for _ in range(warm):
        fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for est, n, extra in ((28005615, 10_000_000, 0), (28005615, 10_000_000, 18_000_000), (224044920, 1 << 25, 0)):
    keys, fresh = gen(n, 0), gen(n, 10 * n)
    one = keys.clone(); one[n // 2] = fresh[0]
    p1e4 = keys.clone(); p1e4[:: 10_000] = fresh[: (n + 9999) // 10_000]
    p1e2 = keys.clone(); p1e2[:: 100] = fresh[: (n + 99) // 100]
    q25 = torch.cat([keys[: 3 * n // 4], fresh[: n - 3 * n // 4]])
    q90 = fresh.clone(); q90[:: 10] = keys[: (n + 9) // 10]
    q75 = fresh.clone(); q75[:: 4] = keys[: (n + 3) // 4]
    blm = pa.BloomFilter(est_elements=est, false_positive_rate=0.01)
    blm.add_many(keys)
    if extra:
        blm.add_many(gen(extra, 100 * n))
    print(f"m = {blm.number_bits}, {n} keys per call, {n + extra} keys inserted", flush=True)
    for mode, name in ((0, "keyed"), (1, "return trip"), (3, "tile flags"), (4, "lazy gathers"), (2, "auto")):
        N.set_option("bloom_lookup", mode)
        cases = [("all-hit", keys), ("1 absent", one), ("1e-4 absent", p1e4), ("1e-2 absent", p1e2), ("25% absent", q25), ("75% absent", q75), ("90% absent", q90), ("all absent", fresh)]
        print(f"  {name:12s}: " + "  ".join(f"{c} {tl(lambda: blm.check_many(k)):7.1f}" for c, k in cases), flush=True)
    N.set_option("bloom_lookup", 2)
    del blm, keys, fresh, one, p1e4, p1e2, q25, q75, q90
    torch.cuda.empty_cache()
