"""round 3 experiment: do pass 1 (VALU-bound) and pass 2 (HBM / LDS-bound) of two independent filters overlap when they are enqueued
on two streams?  (the ceiling of an in-engine two-round pipeline)  slice_bias -1: 64 KiB slice images, so that an apply
workgroup and a scatter workgroup fit one CU's LDS together."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pyprobables_amd as pa
from pyprobables_amd import _native as N
from oracle import oracle

n = 10_000_000
keys = torch.from_numpy(oracle.gen_keys16(0, n)).cuda()
keys2 = torch.from_numpy(oracle.gen_keys16(n, n)).cuda()
for bias in (0, -1):
    N.set_option("slice_bias", bias)
    A = pa.BloomFilter(est_elements=28_005_615, false_positive_rate=0.01)  # ~2^28 bits, k = 7
    B = pa.BloomFilter(est_elements=28_005_615, false_positive_rate=0.01)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for op in ("add", "check", "add|check"):
        def run(f, kk, which):
            if which == "add":
                f.add_many(kk)
            else:
                f.check_many(kk)
        a_op, b_op = (op.split("|") * 2)[:2]
        for _ in range(2):
            run(A, keys, a_op); run(B, keys2, b_op)
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            run(A, keys, a_op); run(B, keys2, b_op)
        torch.cuda.synchronize()
        serial = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            with torch.cuda.stream(s1):
                run(A, keys, a_op)
            with torch.cuda.stream(s2):
                run(B, keys2, b_op)
        torch.cuda.synchronize()
        conc = (time.perf_counter() - t0) / reps
        # offset start: stream 2 begins half an operation later (a lookup of 1/4 of the keys first)
        print(f"bias {bias} m {A.number_bits} {op:10s}: serial {serial*1e6:7.1f} us per pair, two streams {conc*1e6:7.1f} us  ({serial/conc:.2f}x)", flush=True)
