"""Same-box A/B of two engine builds on the CountMinSketch lookup (10 M keys, 2^20 x 5): HIP-event time of check_many.
usage (on the GPU box, one gpurun call for BOTH variants -- boxes differ by up to 25 % on the latency-bound kernels):
    PSK_LIB_PATH=$PWD/ab/variant/libpsk_hip.so python scripts/ab_cms_check_lib.py $PWD
    python scripts/ab_cms_check_lib.py $PWD                      # the in-tree library
argv[1]: directory that holds the pyprobables_amd package to import (the repo root, or a checkout of another commit)."""
import sys, time
sys.path.insert(0, sys.argv[1])
import torch
import pyprobables_amd as pa
print("package:", pa.__file__)
n = 10_000_000
keys = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda")
w = torch.randint(1, 8, (n,), dtype=torch.int32, device="cuda")
import os
W, D = int(os.environ.get("CMS_W", 2**20)), int(os.environ.get("CMS_D", 5))  # geometry (default: BASELINE cfg 3)
cms = pa.CountMinSketch(width=W, depth=D)
cms.add_many(keys, w)
for _ in range(3): cms.check_many(keys)
torch.cuda.synchronize()
for rep in range(3):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): cms.check_many(keys)
    b.record(); torch.cuda.synchronize()
    print(f"cms check: {a.elapsed_time(b) / 20 * 1e3:.1f} us per 10 M keys")
