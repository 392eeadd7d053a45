import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scripts')
from _common import gen_keys, timed_loop, use_knobs_build
use_knobs_build()
import torch
import pyprobables_amd as pa
from pyprobables_amd import _native as N
n = 10_000_000
keys = gen_keys(n, 0, 0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
for _ in range(1500): blm.add_many(keys)
torch.cuda.synchronize()
for rep in range(2):
    for dbg in (0, 16):
        N.set_option("part_debug", dbg)
        ms = timed_loop(lambda: blm.add_many(keys), 20, warm=3)
        print(f"dbg={dbg} insert {ms*1e3:8.1f} us")
N.set_option("part_debug", 0)
