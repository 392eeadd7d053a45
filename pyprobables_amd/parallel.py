"""Multi-GPU: one process per GPU, key stream partitioned by rank, full-size replica per GPU, one merge.

Inserts commute (OR for Bloom, wrap-free SUM for the counters), so rank r inserts its contiguous slice
of the key stream into its own replica with NO communication; a single collective then makes every
replica equal to the table one filter fed the whole stream would hold.

* Bloom:  allreduce(OR).  RCCL has no bitwise-OR reduction (rccl.h ncclRedOp_t: sum/prod/max/min/avg),
  so it is composed: ``all_to_all`` of the R bit-range slices (every GPU sends slice j straight to its
  owner j over its own xGMI link -- all 7 links busy at once, no ring), the HIP ``psk_or_reduce_slices``
  kernel on the R received slices, then ``all_gather`` of the reduced slices.
* CMS / CBF: ``all_reduce(SUM)`` on the int32 / uint32 table; exact whenever no *global* counter reaches
  a rail (the same condition under which the single-GPU fast path is exact); the engine's wrap-free bound
  is then re-derived from the merged table.

``torch.distributed`` (backend "nccl" == RCCL on ROCm) is plumbing here; the reduce kernel is ours.
"""

from __future__ import annotations

import os

from . import _native as N

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous key range [lo, hi) of ``rank`` (the remainder goes to the first ranks)"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def hip_or_reduce(dst, src, nslices: int, slice_words: int) -> None:
    """dst[w] = OR_j src[j*slice_words + w] on the GPU (the engine's kernel; device tensors only)"""
    if not (dst.is_cuda and src.is_cuda):
        raise RuntimeError("hip_or_reduce needs device tensors (there is no CPU fallback)")
    stream = torch.cuda.current_stream(dst.device).cuda_stream or None
    N.check(N.lib().psk_or_reduce_slices(dst.data_ptr(), src.data_ptr(), nslices, slice_words, dst.device.index, stream))


def allreduce_or_(table, group=None, or_reduce=hip_or_reduce):
    """in-place bitwise-OR all-reduce of a 1-D int32 tensor (all_to_all -> OR kernel -> all_gather)"""
    world = dist.get_world_size(group)
    if world == 1 and not os.environ.get("PSK_FORCE_MERGE_PATH"):  # (the env knob lets a 1-GPU test drive RCCL)
        return table
    n = table.numel()
    slice_words = -(-n // world)
    slice_words = (slice_words + 3) & ~3  # 16-byte slices for the uint4 kernel
    padded = slice_words * world
    work = table
    if padded != n:
        work = torch.zeros(padded, dtype=table.dtype, device=table.device)
        work[:n].copy_(table)
    recv = torch.empty_like(work)
    dist.all_to_all_single(recv, work, group=group)            # slice j of every rank lands on rank j
    mine = torch.empty(slice_words, dtype=table.dtype, device=table.device)
    or_reduce(mine, recv, world, slice_words)                  # OR of the R partial slices
    dist.all_gather_into_tensor(work, mine, group=group)       # every rank gets the full merged table
    if padded != n:
        table.copy_(work[:n])
    return table


def _sum_int(value: int, device, group=None) -> int:
    t = torch.tensor([value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def merge_bloom(blm, group=None, or_reduce=hip_or_reduce, sync_elements: bool = True) -> None:
    """make every rank's BloomFilter the filter of the union of all ranks' inserts.

    ``elements_added`` becomes the SUM over ranks (one filter fed the whole stream counts every add,
    bloom.py:250) -- not ``union()``'s estimate (bloom.py:427).  That scalar all-reduce ends in a host
    read; pass ``sync_elements=False`` inside a throughput loop and fix the counter up afterwards."""
    allreduce_or_(blm.table_tensor, group, or_reduce)
    if sync_elements:
        blm.elements_added = _sum_int(blm.elements_added, blm.table_tensor.device, group)


def merge_counters(sk, group=None) -> None:
    """CountMinSketch / CountingBloomFilter: SUM all-reduce of the counter table + elements_added"""
    els = sk.elements_added  # folds the device-side tallies
    t = sk.table_tensor
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    if t.is_cuda:
        N.check(N.lib().psk_rescan_bound(sk._tab.handle, sk._tab.stream))
    total = _sum_int(els, t.device, group)
    sk._els_added = total


class MergeHandle:
    """a merge running on its own HIP stream; ``wait()`` makes torch's current stream wait for it"""

    def __init__(self, stream, device):
        self._stream, self._device = stream, device

    def wait(self) -> None:
        if self._stream is not None:
            torch.cuda.current_stream(self._device).wait_stream(self._stream)
            self._stream = None


_merge_streams: dict = {}


def merge_bloom_async(blm, group=None, or_reduce=hip_or_reduce) -> MergeHandle:
    """``merge_bloom`` (table only, no host sync) on a side stream, so that table-independent work -- typically pass 1 of
    the next lookup, ``BloomFilter.check_many_begin`` -- overlaps the collective.  Everything already enqueued on the
    current stream (the inserts) is ordered before the merge; call ``wait()`` before anything reads the merged table."""
    t = blm.table_tensor
    if not t.is_cuda:  # gloo / CPU tensors (tests): nothing to overlap
        allreduce_or_(t, group, or_reduce)
        return MergeHandle(None, None)
    dev = t.device
    side = _merge_streams.get(dev.index)
    if side is None:
        side = _merge_streams[dev.index] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        allreduce_or_(t, group, or_reduce)
    return MergeHandle(side, dev)
