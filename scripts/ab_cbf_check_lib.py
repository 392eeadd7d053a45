"""Same-box A/B of two engine builds on the CountingBloomFilter lookup into the 1 GiB table of BASELINE cfg 4 (10 M keys, table
changed between lookups / unchanged): HIP-event time of check_many.  Usage: see scripts/ab_cms_check_lib.py (PSK_LIB_PATH picks the library)."""
import sys
sys.path.insert(0, sys.argv[1])
import torch
import pyprobables_amd as pa
n = 10_000_000
keys = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda")
cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
cbf.add_many(keys)
small = keys[:1000].clone()
for _ in range(2): cbf.check_many(keys)
torch.cuda.synchronize()
for rep in range(3):
    t = 0.0
    for _ in range(8):
        cbf.add_many(small)            # the table changes: the 4-bit images are rebuilt by the next lookup
        cbf.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); cbf.check_many(keys); b.record(); torch.cuda.synchronize()
        t += a.elapsed_time(b)
    print(f"cbf check (table changed): {t / 8 * 1e3:.1f} us per 10 M keys")
