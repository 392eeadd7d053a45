"""round 3: from which table size on do the 4-bit slice images beat the 32-bit / 16-bit slices?  CountingBloomFilter tables of
2^23 .. 2^27 counters, 10 M keys: lookups, unit adds, validated removes with the nibble paths on (thresholds lowered) and off"""
import sys
import torch
sys.path.insert(0, "/root/repo")
import pyprobables_amd as pa
from pyprobables_amd import _native as N

def gen(n, start):
    t = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    N.check(N.lib().psk_gen_keys16(t.data_ptr(), start, n, 0x5EED, 0, torch.cuda.current_stream().cuda_stream or None))
    return t

def tl(fn, iters=4, warm=2):
    for _ in range(warm):
        fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

n = 10_000_000
keys = gen(n, 0)
for est in (875_000, 1_750_000, 3_500_701, 7_000_000, 10_000_000):
    for nib in (1, 0):
        N.set_option("nibble_min_lg_lookup", 20 if nib else 40)
        N.set_option("nibble_min_lg_update", 20 if nib else 40)
        cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01)
        cbf.add_many(keys)
        chk = tl(lambda: cbf.check_many(keys))
        cbf.clear()
        add = tl(lambda: cbf.add_many(keys))
        def addrem():
            cbf.add_many(keys); cbf.remove_many(keys)
        ar = tl(addrem, 3, 1)
        print(f"m = {cbf.number_bits:10d} (2^{cbf.number_bits.bit_length() - 1}+) nibble {'on ' if nib else 'off'}: check {chk:7.1f} us  add {add:7.1f}  remove {ar - add:7.1f}", flush=True)
        del cbf
