import sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
import pyprobables_amd as pa
n = 10_000_000
keys = gen_keys(n, 0, 0)
blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=0)
def step():
    blm.clear(); blm.add_many(keys); return blm.check_many(keys)
for _ in range(20): step()
torch.cuda.synchronize(); torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
evs[0].record()
for i in range(30):
    step(); evs[i+1].record()
torch.cuda.synchronize()
print("per-step ms:", " ".join(f"{evs[i].elapsed_time(evs[i+1]):.3f}" for i in range(30)))
