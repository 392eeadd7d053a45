"""round 3: 16 x 1 M-key unit add batches into the 1 GiB CountingBloomFilter table, automatic write-combining on / off"""
import sys
import torch
sys.path.insert(0, "/root/repo")
import pyprobables_amd as pa
from pyprobables_amd import _native as N

def gen(n, start):
    t = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    N.check(N.lib().psk_gen_keys16(t.data_ptr(), start, n, 0x5EED, 0, torch.cuda.current_stream().cuda_stream or None))
    return t

keys = gen(16_000_000, 0)
cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01)
def run(nb, B):
    cbf.clear()
    for b in range(nb):
        cbf.add_many(keys[b * B:(b + 1) * B])
    cbf.synchronize()
for auto in (1, 0, 1, 0):
    N.set_option("auto_combine", auto)
    for nb, B in ((16, 1_000_000), (64, 250_000), (4, 1_000_000)):
        run(nb, B); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): run(nb, B)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        print(f"auto_combine {auto}: {nb} x {B} keys (clear + adds + flush): {ms:7.3f} ms -> {nb * B / ms / 1e3:8.0f} M adds/s", flush=True)
