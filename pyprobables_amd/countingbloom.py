"""CountingBloomFilter: uint32 counters in HBM, add / remove / check as HIP kernels.

Drop-in for the hot path of ``probables.CountingBloomFilter`` (``probables/blooms/countingbloom.py``).

Two execution modes
  * single-key ``add`` / ``remove`` (and ``update_ordered``): the reference semantics executed literally,
    in order, on the GPU -- exact for any stream, including the per-op return value;
  * ``add_many`` / ``remove_many``: unordered batches with k atomics per key.  Bit-exact with the
    reference for well-formed streams (no counter saturates; every remove targets a key with at least
    ``num_els`` live inserts), which makes the result independent of the order inside the batch.
    Anything else is reported through :meth:`batch_diagnostics`.
"""

from __future__ import annotations

import struct

import numpy as np

from . import _native as N
from ._base import weights_arg
from .bloom import BloomFilter, _torch_dtype
from .exceptions import SimilarityError
from .hashes import HashResultsT, KeyT
from .keys import KeyBatch, pack_hashes

_U32_MAX = 2**32 - 1
_U64_MAX = 2**64 - 1


class CountingBloomFilter(BloomFilter):
    """Counting Bloom filter on the GPU (constructor identical to the reference, countingbloom.py:48-55)."""

    _TYPE = "counting"
    _KIND = "cbf"
    _ELEM = struct.Struct("I")
    _MISMATCH = "The parameter second must be of type CountingBloomFilter"

    def __init__(self, est_elements=None, false_positive_rate=None, filepath=None, hex_string=None, hash_function=None,
                 device=None, combine_updates: bool = False, borrow_keys: bool = False):
        """Small unit-weight ``add_many`` batches into big tables (more than 2^26 counters) are write-combined by the engine
        on its own: adds commute, so each batch is hashed and partitioned when it is handed over and the probes are folded
        into the table together with their successors -- exact, no opt-in (``psk_set_option("auto_combine", 0)`` turns it
        off).  Reads and removes always see every update handed over.

        ``combine_updates`` (extra, off by default) also defers ``remove_many``: folding a big table costs a pass over the
        WHOLE table whatever the batch size, so streams that interleave small add and remove batches (BASELINE config 4:
        1M-key batches into 1 GiB) only run fast when the removes wait as well.  They are then plain decrements applied
        after the window's adds -- exact for well-formed streams (every remove targets a key with enough live inserts), the
        contract of the unordered batch ops; a remove of an absent key is tallied in ``batch_diagnostics()['violations']``
        instead of being a no-op (countingbloom.py:200-201).

        ``combine_updates="borrow"``: as above, and device batches of 16-byte keys are not even copied -- the sketch keeps a
        reference to the key tensor and hashes it where it lies at the flush (``PSK_DEVICE_BORROWED``).  The caller must not
        OVERWRITE such a tensor before the next read of the sketch (``check*``, ``elements_added``, ``flush()`` ...).

        ``borrow_keys=True`` (extra, off by default) is the same promise for the DEFAULT path: small ``add_many`` / ``remove_many`` batches
        of 16-byte keys into big tables wait in the engine's update window (exact for any stream: the flush proves every deferred remove
        or replays the window in order) as copies of their keys -- 22 % of BASELINE config 4's step.  With the promise that a device
        key tensor handed over is not overwritten until the sketch is next read, the window keeps a reference instead and hashes the
        batches where they lie.  Results are the same bit for bit; only the copy goes."""
        self._combine = bool(combine_updates)
        self._borrow = combine_updates == "borrow"
        self._borrow_window = bool(borrow_keys)
        self._borrowed: list = []
        super().__init__(est_elements, false_positive_rate, filepath, hex_string, hash_function, device)

    @classmethod
    def frombytes(cls, b, hash_function=None, device=None):
        inst = super().frombytes(b, hash_function, device)
        inst._combine = False
        inst._borrow = False
        inst._borrow_window = False
        inst._borrowed = []
        return inst

    def _release_borrowed(self, keep_last: int = 0) -> None:
        """let go of lent key tensors whose batches the engine has hashed (all but the newest ``keep_last``).  The kernels that read them
        were enqueued on the sketch's stream -- torch's CURRENT stream of the device (``_base.py`` ``stream``) --, so every tensor is
        first recorded on that stream: the caching allocator then keeps its memory until those kernels have run, even when the tensor
        was allocated on another stream.  (The sketch must be driven from the stream the key tensors are ready on; see ``borrow_keys``.)"""
        lent = getattr(self, "_borrowed", None)
        if not lent:
            return
        gone = lent[: len(lent) - keep_last] if keep_last else lent[:]
        try:
            import torch

            cur = torch.cuda.current_stream(self._tab.device) if self._tab is not None else None
            for keep in gone:
                for t in keep if isinstance(keep, (list, tuple)) else (keep,):
                    if cur is not None and isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(cur)
        except ImportError:  # pragma: no cover
            pass
        del lent[: len(gone)]

    @staticmethod
    def _insufficient_msg() -> str:
        return "Insufecient parameters to set up the Counting Bloom Filter"  # (sic) countingbloom.py:73

    def _flush(self) -> None:
        super()._flush()
        if self._tab is not None:  # write-combined updates (automatic for small add batches into big tables) reach the table
            self._tab.flush()
            self._release_borrowed()  # (the flush has hashed the borrowed key tensors: they may go)

    def _table_len(self, n_bits: int) -> int:
        return int(n_bits)  # one uint32 per position (countingbloom.py:77)

    # -------------------------------------------------------------- elements_added bookkeeping
    # adds/removes with device-side weights are tallied on the device; fold them in lazily
    def _fold_counters(self) -> None:
        if self._tab is None or not getattr(self, "_dirty", False):
            return
        c = self._tab.counters()  # (flushes the engine's write-combined updates)
        self._release_borrowed()
        self._els_added = min(self._els_added + c[N.CTR_ADDED], _U64_MAX) - c[N.CTR_REMOVED]
        self._diag = [a + b for a, b in zip(getattr(self, "_diag", [0, 0]), (c[N.CTR_VIOLATIONS], c[N.CTR_SATURATED]))]
        self._tab.reset_counters()
        self._dirty = False

    @property
    def elements_added(self) -> int:
        self._fold_counters()
        return self._els_added

    @elements_added.setter
    def elements_added(self, val: int):
        self._fold_counters()
        self._els_added = val

    def batch_diagnostics(self) -> dict:
        """order-dependence report of the unordered batches so far: ``violations`` (partial / underflowing
        removes) and ``saturated`` (counter updates that hit 2^32-1)"""
        self._dirty = True
        self._fold_counters()
        v, s = getattr(self, "_diag", [0, 0])
        return {"violations": v, "saturated": s}

    def clear(self) -> None:
        super().clear()
        self._dirty, self._diag = False, [0, 0]
        self._release_borrowed()

    # -------------------------------------------------------------- single-key ops (ordered kernel)
    def _ordered(self, b: KeyBatch, num_els, opmode: int) -> np.ndarray:
        if b.where != N.HOST:
            raise ValueError("ordered updates take host batches")
        w = np.ascontiguousarray(np.broadcast_to(np.asarray(num_els, dtype=np.int64), (b.n,)))
        out = np.empty(b.n, dtype=np.uint32)
        N.check(N.lib().psk_cbf_update_ordered(self._tab.handle, *b.args(), w.ctypes.data if b.n else None, opmode,
                                               b.where, out.ctypes.data if b.n else None, self._tab.stream))
        self._dirty = True
        return out

    def _ordered_one(self, key, num_els, opmode: int):
        """one ordered update of one key through the preallocated words (``_base.OneKey``); None: take the general path"""
        raw = self._one_key(key)
        if raw is None or type(num_els) is not int or not 0 <= num_els < 1 << 62:
            return None
        t = self._tab
        one = t.one
        one.w[0] = num_els
        N.check(N.lib().psk_cbf_update_ordered(t.handle, N.KEYS_FIXED, raw or None, None, 1, len(raw), one.w_addr, opmode, N.HOST,
                                               one.o_addr, t.stream))
        self._dirty = True
        return int(one.o_u32[0])

    def add(self, key: KeyT, num_els: int = 1) -> int:
        """countingbloom.py:125-133"""
        res = self._ordered_one(key, num_els, N.OP_ADD)
        return res if res is not None else int(self._ordered(self._batch(key), num_els, N.OP_ADD)[0])

    def add_alt(self, hashes: HashResultsT, num_els: int = 1) -> int:
        """countingbloom.py:135-155"""
        return int(self._ordered(pack_hashes(hashes, self._number_hashes), num_els, N.OP_ADD)[0])

    def remove(self, key: KeyT, num_els: int = 1) -> int:
        """countingbloom.py:176-184"""
        res = self._ordered_one(key, num_els, N.OP_REMOVE)
        return res if res is not None else int(self._ordered(self._batch(key), num_els, N.OP_REMOVE)[0])

    def remove_alt(self, hashes: HashResultsT, num_els: int = 1) -> int:
        """countingbloom.py:186-208"""
        return int(self._ordered(pack_hashes(hashes, self._number_hashes), num_els, N.OP_REMOVE)[0])

    def update_ordered(self, keys, signed_num_els) -> np.ndarray:
        """execute a mixed stream strictly in order on the device: ``w >= 0`` adds ``w``, ``w < 0`` removes
        ``-w``; returns every op's reference return value (uint32[n])"""
        return self._ordered(self._batch(keys), signed_num_els, N.OP_SIGNED)

    def check(self, key: KeyT) -> int:  # type: ignore[override]
        """countingbloom.py:157-164"""
        raw = self._one_key(key)
        if raw is None:
            return int(self._check_batch(self._batch(key))[0])
        t = self._tab
        one = t.one
        N.check(N.lib().psk_cbf_check(t.handle, N.KEYS_FIXED, raw or None, None, 1, len(raw), N.HOST, one.o_addr, t.stream))
        self._release_borrowed()
        return int(one.o_u32[0])

    def check_alt(self, hashes: HashResultsT) -> int:  # type: ignore[override]
        """countingbloom.py:166-174 (min over ALL supplied hashes)"""
        return int(self._check_batch(pack_hashes(hashes, 1))[0])

    # -------------------------------------------------------------- batch ops (unordered kernels)
    def _check_batch(self, b: KeyBatch):
        addr, fin = self._tab.out_buffer(b, b.n, np.uint32, _torch_dtype("int32"))
        N.check(N.lib().psk_cbf_check(self._tab.handle, *b.args(), b.where, addr, self._tab.stream))
        self._release_borrowed()  # (the engine applied what was waiting before it looked anything up)
        return fin()

    def _update_batch(self, remove: bool, b: KeyBatch, num_els) -> None:
        keep: list = []
        w_addr, _ = weights_arg(num_els, b.n, np.uint32, b.where, keep, 0, _U32_MAX, self._tab.device)
        if getattr(self, "_combine", False):  # write-combined: collected on the device, applied per 2^26 keys
            where = b.where
            if getattr(self, "_borrow", False) and where == N.DEVICE and w_addr is None:
                where = N.DEVICE_BORROWED        # the engine keeps the POINTER: hold the buffers until the next flush
                # Drop our references in step with the engine's own flushes (it flushes a list at 4096 batches or `combine_keys`
                # keys), BEFORE this batch is handed over: a flush after the append would release the very batch the engine is about
                # to be given a pointer to.
                self._borrowed_keys = getattr(self, "_borrowed_keys", 0) + b.n
                if len(self._borrowed) >= 4000 or self._borrowed_keys > N.get_option("combine_keys"):
                    self._flush()
                    self._borrowed_keys = b.n
                self._borrowed.append(b.keep)
            N.check(N.lib().psk_cbf_update_combined(self._tab.handle, *b.args(), w_addr, int(remove), where, self._tab.stream))
        else:
            fn = N.lib().psk_cbf_remove if remove else N.lib().psk_cbf_add
            where = b.where
            lend = (getattr(self, "_borrow_window", False) and where == N.DEVICE and w_addr is None and b.layout == N.KEYS_FIXED and b.key_len == 16
                    and b.data % 16 == 0 and b.n)
            if lend:
                where = N.DEVICE_BORROWED  # the engine may keep the POINTER (update window): hold the tensor until the window is applied
                self._borrowed.append(b.keep)
            N.check(fn(self._tab.handle, *b.args(), w_addr, where, self._tab.stream))
            if len(self._borrowed) >= 128 or (self._borrowed and not lend):
                # the window holds its latest `pending` batches, ours among them: everything older has been hashed and may go
                # (asked every 128 lent batches, not every call: the question is a ctypes round trip on a path that is enqueue-bound)
                pending = self._tab.get_option("window_pending_batches")
                if pending < len(self._borrowed):
                    self._release_borrowed(keep_last=pending)
        self._dirty = True

    def _add_batch(self, b: KeyBatch, num_els=None) -> None:  # type: ignore[override]
        self._update_batch(False, b, num_els)

    def add_many(self, keys, num_els=None) -> None:  # type: ignore[override]
        """``num_els``: None (=1), an int, or one count per key"""
        self._add_batch(self._batch(keys), num_els)

    def add_alt_many(self, hashes, num_els=None) -> None:  # type: ignore[override]
        self._add_batch(pack_hashes(hashes, self._number_hashes), num_els)

    def remove_many(self, keys, num_els=None) -> None:
        self._update_batch(True, self._batch(keys), num_els)

    def remove_alt_many(self, hashes, num_els=None) -> None:
        self._update_batch(True, pack_hashes(hashes, self._number_hashes), num_els)

    def check_many(self, keys):
        """uint32[n] numpy (host input) / int32-bits torch tensor (device input): min counter per key"""
        return self._check_batch(self._batch(keys))

    def check_alt_many(self, hashes):
        return self._check_batch(pack_hashes(hashes, 1))

    def check_many_bits(self, keys):
        raise NotImplementedError("ballot bitmaps are a BloomFilter feature")

    # -------------------------------------------------------------- statistics / algebra
    def _cnt_number_bits_set(self) -> int:
        return self._tab.nonzero()  # countingbloom.py:302-304

    def __str__(self) -> str:
        """countingbloom.py:99-123 (host statistics over a snapshot of the counters)"""
        tab = self._tab.read().view(np.uint32)
        total = int(tab.astype(np.uint64).sum())
        largest = int(tab.max()) if tab.size else 0
        largest_idx = int(tab.argmax()) if tab.size else 0
        fullness = total / self.number_bits
        return (
            "CountingBloom:\n"
            f"\tbits: {self.number_bits}\n"
            f"\testimated elements: {self.estimated_elements}\n"
            f"\tnumber hashes: {self.number_hashes}\n"
            f"\tmax false positive rate: {self.false_positive_rate:.6f}\n"
            f"\telements added: {self.elements_added}\n"
            f"\tcurrent false positive rate: {self.current_false_positive_rate():.6f}\n"
            "\tis on disk: no\n"
            f"\tindex fullness: {fullness:.6}\n"
            f"\tmax index usage: {largest}\n"
            f"\tmax index id: {largest_idx}\n"
            f"\tcalculated elements: {total // self.number_hashes}\n"
        )

    def _require_similar(self, second, msg="Counting Bloom Filters are not similar enough to calculate similarity"):
        if not isinstance(second, CountingBloomFilter):
            raise TypeError(self._MISMATCH)
        if self._verify_bloom_similarity(second) is False:
            raise SimilarityError(msg)
        if second._tab.device != self._tab.device:
            raise ValueError("set operations need both filters on the same device")

    def union(self, second):
        """element-wise sum (countingbloom.py:271-300)"""
        import ctypes as C  # noqa: PLC0415

        self._require_similar(second)
        res = CountingBloomFilter(self.estimated_elements, self.false_positive_rate, hash_function=self.hash_function,
                                  device=self._tab.device)
        L, t = N.lib(), res._tab
        ov = C.c_uint64(0)
        N.check(L.psk_table_add_u32(t.ptr, self._tab.ptr, self.number_bits, C.byref(ov), t.device, t.stream))
        N.check(L.psk_table_add_u32(t.ptr, second._tab.ptr, self.number_bits, C.byref(ov), t.device, t.stream))
        if ov.value:  # the reference's array('I') store raises here
            raise OverflowError("unsigned int is greater than maximum")
        res.elements_added = res.estimate_elements()
        return res

    def intersection(self, second):
        """countingbloom.py:210-240: sum where both are non-zero (one streaming kernel)"""
        import ctypes as C  # noqa: PLC0415

        self._require_similar(second)
        res = CountingBloomFilter(self.estimated_elements, self.false_positive_rate, hash_function=self.hash_function,
                                  device=self._tab.device)
        t, ov = res._tab, C.c_uint64(0)
        N.check(N.lib().psk_cbf_intersect(t.ptr, self._tab.ptr, second._tab.ptr, self.number_bits, C.byref(ov), t.device, t.stream))
        if ov.value:  # the reference's array('I') store raises here
            raise OverflowError("unsigned int is greater than maximum")
        res.elements_added = res.estimate_elements()
        return res

    def jaccard_index(self, second) -> float:
        """countingbloom.py:242-269 on the sets of non-zero positions (one streaming kernel)"""
        import ctypes as C  # noqa: PLC0415

        self._require_similar(second)
        out = (C.c_uint64 * 2)()
        t = self._tab
        N.check(N.lib().psk_cbf_jaccard_counts(t.ptr, second._tab.ptr, self.number_bits, out, t.device, t.stream))
        if out[0] == 0:
            return 1.0
        return out[1] / out[0]
