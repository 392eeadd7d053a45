"""how the all-cores CPU leg of bench.py scales with threads on the GPU box (is the container CPU-limited?)"""
import os, sys
sys.path.insert(0, "oracle")
import oracle
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.__class__.__name__)
print("loadavg", open("/proc/loadavg").read().strip())
m, k = 2**28, 7
for t in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    n = 2_000_000 * min(t, 32)
    ob = oracle.OracleBloom(m, k)
    ob.insert_check_mt_shared(0, n, t)
    print(f"shared table, {t:3d} threads: {2 * n / ob.mt_seconds / 1e6:8.1f} Mkeys/s ({ob.mt_seconds:.2f} s)", flush=True)
