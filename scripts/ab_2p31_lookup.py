import sys
sys.path.insert(0, "scripts")
from _common import gen_keys, timed_loop
import torch
import pyprobables_amd as pa
from pyprobables_amd import _native as N
n = 1 << 25
keys = gen_keys(n)
s = pa.BloomFilter(est_elements=224044920, false_positive_rate=0.01)
for _ in range(30):
    s.add_many(keys)
torch.cuda.synchronize()
for rep in range(2):
    for wgs in (0, 512, 768, 1024):
        N.set_option("scatter_workgroups", wgs)
        c = timed_loop(lambda: s.check_many(keys), 6)
        print(f"2^31 lookup wgs {wgs:4d}: {n / c / 1e3:8.0f} M keys/s")
N.set_option("scatter_workgroups", 0)
