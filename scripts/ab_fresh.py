import os, sys
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
from _common import gen_keys, timed_loop
import torch
import pyprobables_amd as pa
n = 10_000_000
keys = gen_keys(n); fresh = gen_keys(n, 10 * n)
mixed = torch.cat([keys[: n // 2], fresh[: n // 2]])
f = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01)
f.add_many(keys)
for name, b in (("hit", keys), ("fresh", fresh), ("mixed", mixed)):
    t = timed_loop(lambda: f.check_many(b), 10, warm=3)
    print(os.path.basename(os.environ.get("PSK_LIB_PATH", "default")), name, f"{t*1e3:.1f} us {n/t/1e3:.0f} M/s")
