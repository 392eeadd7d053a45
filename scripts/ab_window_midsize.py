"""Update windows on tables between 2^24 and 2^28 counters (fewer slices than BASELINE cfg 4's 1024): G ops/s of cfg 4's stream shape
(batch b adds B keys, then removes the first half of batch b - 1) with the window on / off.   usage: ab_window_midsize.py <est_elements> [B] [batches]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import pyprobables_amd as pa
from pyprobables_amd import _native as N

est = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000; nb = int(sys.argv[3]) if len(sys.argv) > 3 else 20
keys = torch.randint(0, 256, (nb * B, 16), dtype=torch.uint8, device="cuda")
for win in (1, 0, 1, 0):
    N.set_option("update_window", win)
    cbf = pa.CountingBloomFilter(est_elements=est, false_positive_rate=0.01)
    def step():
        cbf.clear()
        for b in range(nb):
            cbf.add_many(keys[b * B:(b + 1) * B])
            if b: cbf.remove_many(keys[(b - 1) * B:(b - 1) * B + B // 2])
        cbf.synchronize()
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    ops = nb * B + (nb - 1) * (B // 2)
    print(f"m={cbf.number_bits} window={win}: {dt*1e3:8.2f} ms per step of {ops/1e6:.1f} M ops = {ops/dt/1e9:6.2f} G ops/s  folds={N.get_option('update_window_folds')} replays={N.get_option('update_window_replays')}", flush=True)
    del cbf
