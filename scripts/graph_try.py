#!/usr/bin/env python3
"""Does a HIP graph of one bench step (clear + insert + lookup) shave the inter-kernel gaps?"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _common import gen_keys, gen_weights, timed_loop  # noqa: E402
import torch

import bench
import pyprobables_amd as pa

n = 10_000_000
keys = gen_keys(n, 0, 0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
res = {}


def step():
    blm.clear()
    blm.add_many(keys)
    res["r"] = blm.check_many(keys)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 50
print(f"eager  {eager*1e3:.4f} ms/step")

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 50
print(f"graph  {graph*1e3:.4f} ms/step  all found: {bool(res['r'].all().item())}")
