"""A/B (round 3): pass 3 of the counter lookups as 1024-thread (two tiles in flight per CU) or 512-thread workgroups (four)"""
import sys
import torch
sys.path.insert(0, "/root/repo")
import pyprobables_amd as pa
from pyprobables_amd import _native as N

def gen(n, start):
    t = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    N.check(N.lib().psk_gen_keys16(t.data_ptr(), start, n, 0x5EED, 0, torch.cuda.current_stream().cuda_stream or None))
    return t

def tl(fn, iters=8, warm=3):
    for _ in range(warm):
        fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

n = 10_000_000
keys = gen(n, 0)
cms = pa.CountMinSketch(width=2**20, depth=5)
cms.add_many(keys)
cbf = pa.CountingBloomFilter(est_elements=3_400_000, false_positive_rate=0.01)  # ~2^25 counters, 32-bit slices
cbf.add_many(keys)
for t in (1024, 512, 1024, 512):
    N.set_option("lookup_collect_threads", t)
    print(f"collect threads {t}: cms check {tl(lambda: cms.check_many(keys)):7.1f} us   cbf(2^25) check {tl(lambda: cbf.check_many(keys)):7.1f} us", flush=True)
