"""Where a value-returning per-key call spends its time.
GPU box:  python scripts/latency_probe.py
Three layers per operation: the Python mirror method, the bare C-ABI call through ctypes with prebuilt arguments, and (Bloom only) an
empty-batch call (n = 0: argument checks only, no launch) as the ctypes floor."""
import ctypes as C, time, sys
import numpy as np
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyprobables_amd import BloomFilter, CountingBloomFilter, CountMinSketch, _native as N
from pyprobables_amd.keys import pack_keys


def loop(f, n=3000):
    for _ in range(200):
        f()
    t = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t) / n * 1e6


blm = BloomFilter(est_elements=1_000_000, false_positive_rate=0.01)
blm.add_many([f"k{i}" for i in range(1000)])
key = sys.argv[1] if len(sys.argv) > 1 else "k17"
print(f"key = {key!r} ({len(key)} characters)")
sys.path.insert(0, str(ROOT / "oracle"))
import pymirror  # the interpreted per-key loop of the reference, restated (oracle/pymirror.py): the same host's figure to compare with

mb = pymirror.MirrorBloom(blm.number_bits, blm.number_hashes)
for i in range(1000):
    mb.add(f"k{i}")
mb.add(key)
blm.add(key)
print(f"interpreted per-key loop (mirror) {loop(lambda: mb.check(key)):7.2f} us   (bloom.py:252-272 restated, this host)")
print(f"BloomFilter.check(key)            {loop(lambda: blm.check(key)):7.2f} us   (python mirror)")
print(f"key in blm                        {loop(lambda: key in blm):7.2f} us")
b = pack_keys([key])
out = np.zeros(1, np.uint8)
L = N.lib()
args = (blm._tab.handle, *b.args(), b.where, out.ctypes.data, blm._tab.stream)
print(f"psk_bloom_check via ctypes        {loop(lambda: L.psk_bloom_check(*args)):7.2f} us   (prebuilt arguments)")
N.set_option("host_poll_us", 0)
print(f"  ... host_poll_us = 0            {loop(lambda: L.psk_bloom_check(*args)):7.2f} us   (stream wait instead of the mailbox)")
N.set_option("host_poll_us", 200)
print(f"pack_keys([key])                  {loop(lambda: pack_keys([key])):7.2f} us")
print(f"_tab.out_buffer                   {loop(lambda: blm._tab.out_buffer(b, 1, np.uint8, None)):7.2f} us")
cms = CountMinSketch(width=1 << 20, depth=5)
print(f"CountMinSketch.add(key)           {loop(lambda: cms.add(key)):7.2f} us")
print(f"CountMinSketch.check(key)         {loop(lambda: cms.check(key)):7.2f} us")
cbf = CountingBloomFilter(est_elements=1_000_000, false_positive_rate=0.01)
print(f"CountingBloomFilter.add(key)      {loop(lambda: cbf.add(key)):7.2f} us")
print(f"CountingBloomFilter.check(key)    {loop(lambda: cbf.check(key)):7.2f} us")
print(f"CountingBloomFilter.remove(key)   {loop(lambda: (cbf.add(key), cbf.remove(key))) / 2:7.2f} us   (half of an add + remove pair)")
from pyprobables_amd import ExpandingBloomFilter

ebf = ExpandingBloomFilter(est_elements=1_000_000, false_positive_rate=0.01)
ctr = [0]


def fresh_add():
    ctr[0] += 1
    ebf.add(f"fresh-{ctr[0]}")


print(f"ExpandingBloomFilter.add(fresh)   {loop(fresh_add):7.2f} us   (one filter: a check, then an insert)")
print(f"ExpandingBloomFilter.add(seen)    {loop(lambda: ebf.add('fresh-1')):7.2f} us")
print(f"ExpandingBloomFilter.check(key)   {loop(lambda: ebf.check('fresh-1')):7.2f} us")
