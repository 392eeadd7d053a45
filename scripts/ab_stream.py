"""round 3: cost of enqueueing inserts on a non-default stream (ab_overlap.py showed two-stream inserts 3.5x slower than serial)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pyprobables_amd as pa
from oracle import oracle

n = 10_000_000
keys = torch.from_numpy(oracle.gen_keys16(0, n)).cuda()
A = pa.BloomFilter(est_elements=28_005_615, false_positive_rate=0.01)
B = pa.BloomFilter(est_elements=28_005_615, false_positive_rate=0.01)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def timed(label, body, reps=10):
    for _ in range(2):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        body()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:48s} {(t2 - t0) / reps * 1e6:8.1f} us per iteration (host enqueue {(t1 - t0) / reps * 1e6:7.1f} us)", flush=True)

def on(s, f, op):
    with torch.cuda.stream(s):
        getattr(f, op)(keys)

timed("A.add default stream", lambda: A.add_many(keys))
timed("A.add on s1", lambda: on(s1, A, "add_many"))
timed("A.add on s1, B.add on s1", lambda: (on(s1, A, "add_many"), on(s1, B, "add_many")))
timed("A.add on s1, B.add on s2", lambda: (on(s1, A, "add_many"), on(s2, B, "add_many")))
timed("A.add on s1, B.check on s2", lambda: (on(s1, A, "add_many"), on(s2, B, "check_many")))
timed("A.check on s1, B.check on s2", lambda: (on(s1, A, "check_many"), on(s2, B, "check_many")))
timed("A.add default, B.add default", lambda: (A.add_many(keys), B.add_many(keys)))
