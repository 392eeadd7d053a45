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
import weakref

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


_merge_bufs: dict = {}


def _merge_buffers(table, world: int, slice_words: int, lane: str = "sync"):
    """(work | None, recv, mine) for one (table, world, lane): allocated once, reused by every merge of that table -- the
    collective sits inside timed loops and two table-sized allocations per call are not free.  ``lane`` keeps the
    buffers of ``merge_bloom_async`` (side stream) apart from those of the synchronous merge, so that a synchronous
    merge issued before an async handle's ``wait()`` cannot race on them.  The entry is dropped when the table tensor
    is freed (the buffers are about 2x the table)."""
    key = (table.data_ptr(), table.numel(), str(table.dtype), world, str(table.device), lane)
    bufs = _merge_bufs.get(key)
    if bufs is None:
        padded = slice_words * world
        work = None if padded == table.numel() else torch.zeros(padded, dtype=table.dtype, device=table.device)
        recv = torch.empty(padded, dtype=table.dtype, device=table.device)
        mine = torch.empty(slice_words, dtype=table.dtype, device=table.device)
        if len(_merge_bufs) >= 8:  # views of tables come and go (tests): keep the cache small
            _merge_bufs.pop(next(iter(_merge_bufs)))
        bufs = _merge_bufs[key] = (work, recv, mine)
        try:
            weakref.finalize(table, _merge_bufs.pop, key, None)
        except TypeError:  # pragma: no cover  (an object that cannot be weakly referenced: the size cap above still holds)
            pass
    return bufs


def release_merge_buffers() -> None:
    """drop the cached exchange buffers (one table-sized buffer per merged table)"""
    _merge_bufs.clear()


def allreduce_or_(table, group=None, or_reduce=hip_or_reduce, lane: str = "sync"):
    """in-place bitwise-OR all-reduce of a 1-D int32 tensor (all_to_all -> OR kernel -> all_gather)"""
    world = dist.get_world_size(group)
    if world == 1 and not os.environ.get("PSK_FORCE_MERGE_PATH"):  # (the env knob lets a 1-GPU test drive RCCL)
        return table
    n = table.numel()
    slice_words = -(-n // world)
    slice_words = (slice_words + 3) & ~3  # 16-byte slices for the uint4 kernel
    work, recv, mine = _merge_buffers(table, world, slice_words, lane)
    src = table
    if work is not None:  # n is not a multiple of 4 * world words: exchange a zero-padded copy
        work[:n].copy_(table)
        src = work
    dist.all_to_all_single(recv, src, group=group)             # slice j of every rank lands on rank j
    or_reduce(mine, recv, world, slice_words)                  # OR of the R partial slices
    dist.all_gather_into_tensor(src, mine, group=group)        # every rank gets the full merged table
    if work is not None:
        table.copy_(work[:n])
    return table


def _sum_int(value: int, device, group=None) -> int:
    t = torch.tensor([value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def _sum_wide(value: int, device, group=None) -> int:
    """exact SUM of python ints that may sit near the 64-bit rails: (high, low) 32-bit limbs, summed separately"""
    t = torch.tensor([value >> 32, value & 0xFFFFFFFF], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    hi, lo = (int(x) for x in t.tolist())
    return (hi << 32) + lo


def merge_bloom(blm, group=None, or_reduce=hip_or_reduce, sync_elements: bool = True) -> None:
    """make every rank's BloomFilter the filter of the union of all ranks' inserts.

    ``elements_added`` becomes the SUM over ranks (one filter fed the whole stream counts every add,
    bloom.py:250) -- not ``union()``'s estimate (bloom.py:427).  That scalar all-reduce ends in a host
    read; pass ``sync_elements=False`` inside a throughput loop and fix the counter up afterwards."""
    allreduce_or_(blm.table_tensor, group, or_reduce)
    if sync_elements:
        blm.elements_added = _sum_int(blm.elements_added, blm.table_tensor.device, group)


_I32_MAX, _I32_MIN, _U32_MAX = 2**31 - 1, -(2**31), 2**32 - 1
_I64_MAX, _I64_MIN, _U64_MAX = 2**63 - 1, -(2**63), 2**64 - 1


def _abs_bound(sk, t, unsigned: bool) -> int:
    """upper bound on |any counter| of this rank's replica: the engine's device-resident PSK_CTR_ABS_BOUND when the
    sketch has a handle, else (CPU tensors in the gloo tests) the exact maximum"""
    tab = getattr(sk, "_tab", None)
    if tab is not None and getattr(tab, "handle", None):
        return int(tab.counters()[N.CTR_ABS_BOUND])
    if t.numel() == 0:
        return 0
    w = t.to(torch.int64)
    return int(((w & 0xFFFFFFFF) if unsigned else w.abs()).max().item())


def merge_counters(sk, group=None, unsigned: bool | None = None) -> None:
    """CountMinSketch / CountingBloomFilter: SUM all-reduce of the counter table + elements_added.

    A 32-bit ``all_reduce(SUM)`` wraps silently, so the ranks first agree on the SUM of their per-replica bounds on
    |counter|.  Below the rail the plain 32-bit reduction is exact.  Otherwise the tables are widened to int64, summed,
    and clamped to the rails the way ``join`` does (countminsketch.py:380-391: INT32_MAX / INT32_MIN; the CBF add
    clamps at 2^32-1, countingbloom.py:149-151) -- for add-only streams exactly the table ONE sketch fed every rank's
    updates would hold, since its saturating adds stop at the same rail."""
    if unsigned is None:
        unsigned = getattr(type(sk), "_KIND", "") == "cbf"
    els = sk.elements_added  # folds the device-side tallies
    t = sk.table_tensor
    bound_sum = _sum_int(min(_abs_bound(sk, t, unsigned), 1 << 40), t.device, group)
    if bound_sum <= (_U32_MAX if unsigned else _I32_MAX):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)   # wrap-free: no global counter can reach a rail
    else:
        wide = t.to(torch.int64)
        if unsigned:
            wide &= 0xFFFFFFFF                                   # the int32 tensor holds uint32 bit patterns
        dist.all_reduce(wide, op=dist.ReduceOp.SUM, group=group)
        if unsigned:
            wide.clamp_(max=_U32_MAX)
            wide[wide > _I32_MAX] -= 1 << 32                     # back to the int32 bit pattern
        else:
            wide.clamp_(min=_I32_MIN, max=_I32_MAX)
        t.copy_(wide.to(torch.int32))
    if t.is_cuda:
        N.check(N.lib().psk_rescan_bound(sk._tab.handle, sk._tab.stream))
    # elements_added: SUM over ranks, then the reference's scalar clamp (countminsketch.py:285-287 / countingbloom.py:154)
    total = _sum_wide(els, t.device, group)
    sk._els_added = max(min(total, _U64_MAX if unsigned else _I64_MAX), 0 if unsigned else _I64_MIN)


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
        allreduce_or_(t, group, or_reduce, lane="async")
    return MergeHandle(side, dev)
