"""ExpandingBloomFilter / RotatingBloomFilter on the GPU engine.

Drop-in for ``probables/blooms/expandingbloom.py``.  The reference's insert is conditional on a lookup
("add the key to the newest filter unless ANY filter of the stack already reports it", :149-170), which makes
a stream order dependent: whether key i is inserted depends on every earlier insert.  ``add_many`` keeps that
semantics *exactly* for a whole ordered batch:

* the batch is hashed once (all filters share m, k and the hash family): ``psk_bloom_indices``;
* it is cut into chunks that cannot cross a growth / rotation boundary (a chunk holds at most as many
  not-yet-present keys as the newest filter has room for), so inside a chunk every older filter is constant;
* ``psk_idx_resolve_ordered`` decides, in parallel, which keys of the chunk the sequential loop would insert
  (a key is inserted exactly when it is the first key of the chunk to touch one of its clear bits: proof in
  ``csrc/psk_index_ops.hip``);
* ``psk_idx_insert`` ORs the winners into the newest filter.

Growth is lazy exactly as in the reference: a full newest filter is only replaced when a key really has to be
inserted.  The single-key ``add`` / ``check`` methods are the same code with a batch of one.
"""

from __future__ import annotations

import ctypes as C
import math
import struct
from pathlib import Path

import numpy as np

from . import _native as N
from .bloom import BloomFilter, _existing_file
from .exceptions import RotatingBloomFilterError
from .hashes import HashFuncT, HashResultsT, KeyT, default_fnv_1a
from .keys import KeyBatch, pack_hashes

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

_WINDOW = 1 << 22  # keys hashed / resolved per round (bounds the index scratch: window * k * 4 bytes)


class ExpandingBloomFilter:
    """Bloom filter that grows by stacking filters (constructor identical to expandingbloom.py:46-52)."""

    _FOOTER = struct.Struct("QQQf")  # size, est_elements, elements_added, fpr   (expandingbloom.py:70)
    _COUNT = struct.Struct("Q")      # per-filter elements_added                 (expandingbloom.py:71)

    def __init__(self, est_elements: int | None = None, false_positive_rate: float | None = None,
                 filepath: str | Path | None = None, hash_function: HashFuncT | None = None, device=None):
        self._blooms: list[BloomFilter] = []
        self._fpr = false_positive_rate if false_positive_rate is not None else 0.0
        self._est_elements = est_elements if est_elements is not None else 100
        self._hash_func: HashFuncT = hash_function if hash_function is not None else default_fnv_1a
        self._added_elements = 0
        self._device = device
        self._scratch: dict = {}
        self.last_batch_stats = {"chunks": 0}  # diagnostics
        if _existing_file(filepath):
            self._load(Path(filepath).expanduser().resolve().read_bytes())
        else:
            self._add_bloom_filter()

    # ------------------------------------------------------------------ construction helpers
    def _new_filter(self) -> BloomFilter:
        return BloomFilter(est_elements=self._est_elements, false_positive_rate=self._fpr,
                           hash_function=self._hash_func, device=self._device)

    def _add_bloom_filter(self) -> None:
        """expandingbloom.py:172-179"""
        self._blooms.append(self._new_filter())

    @classmethod
    def frombytes(cls, b, hash_function: HashFuncT | None = None, device=None) -> "ExpandingBloomFilter":
        """expandingbloom.py:73-87"""
        size, est_els, added_els, fpr = cls._parse_footer(b)
        blm = cls(est_elements=est_els, false_positive_rate=fpr, hash_function=hash_function, device=device)
        blm._parse_blooms(b, size)
        blm._added_elements = added_els
        return blm

    # ------------------------------------------------------------------ properties (expandingbloom.py:100-126)
    @property
    def expansions(self) -> int:
        return len(self._blooms) - 1

    @property
    def false_positive_rate(self) -> float:
        return self._fpr

    @property
    def estimated_elements(self) -> int:
        return self._est_elements

    @property
    def elements_added(self) -> int:
        return self._added_elements

    @property
    def hash_function(self) -> HashFuncT:
        return self._hash_func

    def __contains__(self, key: KeyT) -> bool:
        return self.check(key)

    def __bytes__(self) -> bytes:
        return b"".join(self._export_parts())

    def push(self) -> None:
        """expandingbloom.py:128-130"""
        self._add_bloom_filter()

    # ------------------------------------------------------------------ device plumbing
    @property
    def _k(self) -> int:
        return self._blooms[-1].number_hashes

    def _dev(self) -> int:
        return self._blooms[-1].device

    def _stream(self):
        return self._blooms[-1]._tab.stream

    def _buf(self, name: str, count: int, dtype, fill=None):
        """device scratch that only ever grows (torch owns the memory; the kernels are ours)"""
        t = self._scratch.get(name)
        if t is None or t.numel() < count:
            t = torch.empty(max(count, 1), dtype=dtype, device=f"cuda:{self._dev()}")
            if fill is not None:
                t.fill_(fill)
            self._scratch[name] = t
        return t

    def _indices(self, b: KeyBatch):
        """the k bit positions of every key of the batch (bloom.py:247), hashed once for the whole stack"""
        last = self._blooms[-1]
        if last.number_bits >= 1 << 32:  # (bit position 2^32 - 1 is the resolution map's "empty" marker)
            raise NotImplementedError("stacked filters need m <= 2^32 bits per filter on this engine")
        idx = self._buf("idx", b.n * self._k, torch.int32)
        N.check(N.lib().psk_bloom_indices(last._tab.handle, *b.args(), b.where, idx.data_ptr(), self._stream()))
        return idx

    def _present(self, idx, start: int, count: int):
        """uint8[count]: is key start+i reported by ANY filter (expandingbloom.py:147)"""
        out = self._buf("present", count, torch.uint8)
        k, L = self._k, N.lib()
        for j, f in enumerate(self._blooms):
            N.check(L.psk_idx_test(f.table_tensor.data_ptr(), idx.data_ptr() + 4 * start * k, count, k, out.data_ptr(),
                                   1 if j else 0, self._dev(), self._stream()))
        return out[:count]

    def _insert(self, blm: BloomFilter, idx, start: int, count: int, flag) -> None:
        N.check(N.lib().psk_idx_insert(blm.table_tensor.data_ptr(), idx.data_ptr() + 4 * start * self._k,
                                       flag.data_ptr() if flag is not None else None, count, self._k, self._dev(),
                                       self._stream()))

    _SUB = 1 << 18  # keys per resolution step: the scratch is a hash map over the clear bits ONE step touches (see _resolve_insert)

    def _resolve_insert(self, blm: BloomFilter, idx, start: int, count: int, present) -> int:
        """insert, into ``blm``, exactly the keys of the ordered chunk the reference loop would insert; -> how many.
        The chunk is walked in steps of ``_SUB`` keys -- resolve (which of the step's keys find a clear bit first), insert, next: each
        step sees the filter as the previous one left it, which is the sequential semantics.  Scratch: a 2^lg-slot map of
        (bit position, first candidate) sized for one step -- 32 MiB for k = 7 -- where round 3 kept a uint32 per filter BIT (1 GiB for
        m = 2^28)."""
        k = self._k
        lg = max(12, (2 * min(count, self._SUB) * k - 1).bit_length())
        # ONE map, sized for a full step (the kernel takes lg as a parameter and leaves the slots it used reset): chunk sizes that vary
        # used to leave a map per size alive
        lg_max = max(12, (2 * self._SUB * k - 1).bit_length())
        slots = self._buf("slots", 2 << lg_max, torch.int32, fill=-1)
        flag = self._buf("flag", min(count, self._SUB), torch.uint8)
        cnt = self._buf("count", 2, torch.int64)
        cnt.zero_()
        L, tab = N.lib(), blm.table_tensor.data_ptr()
        for s in range(0, count, self._SUB):
            n = min(self._SUB, count - s)
            N.check(L.psk_idx_resolve_ordered_hashed(tab, idx.data_ptr() + 4 * (start + s) * k, (present.data_ptr() + s) if present is not None else None,
                                                    n, k, slots.data_ptr(), lg, flag.data_ptr(), cnt.data_ptr(), self._dev(), self._stream()))
            N.check(L.psk_idx_insert(tab, idx.data_ptr() + 4 * (start + s) * k, flag.data_ptr(), n, k, self._dev(), self._stream()))
        inserted, full = (int(x) for x in cnt.tolist())
        if full:  # (a map of twice the step's probes cannot fill: a guard, not a path)
            raise RuntimeError("stacked filter: the resolution map overflowed")
        return inserted

    # ------------------------------------------------------------------ growth policy
    def _room(self, blm: BloomFilter):
        """inserts the newest filter still takes before the next insert replaces it (expandingbloom.py:181-184:
        ``elements_added >= est_elements`` grows)"""
        return max(0, math.ceil(self._est_elements) - blm.elements_added)

    def _grow(self) -> None:
        self._add_bloom_filter()

    # ------------------------------------------------------------------ the ordered batch
    def _add_batch(self, b: KeyBatch, force: bool) -> None:
        n = b.n
        if n == 0:
            return
        idx = self._indices(b)
        k, s = self._k, 0
        while s < n:
            last = self._blooms[-1]
            room = self._room(last)
            rem = n - s
            if force:  # expandingbloom.py:168: every key is inserted; the stack only grows at count boundaries
                if room == 0:
                    self._grow()
                    continue
                take = rem if room is None else min(room, rem)
                self._insert(last, idx, s, take, None)
                last._els_added += take
                self._added_elements += take
                s += take
                continue
            if rem == 1:  # single-key call: one lookup, at most one insert, one host sync
                if int(self._present(idx, s, 1)[0].item()):
                    self._added_elements += 1
                    return
                if room == 0:
                    self._grow()  # (a rotation may drop the oldest filter: the key is still absent afterwards)
                    last = self._blooms[-1]
                self._insert(last, idx, s, 1, None)
                last._els_added += 1
                self._added_elements += 1
                return
            # Look at a window of the remaining keys: the chunk ends at the room-th candidate anyway, so testing far
            # beyond it against every filter would be wasted work (a window of repeats just yields a smaller chunk).
            win = rem if room is None else min(rem, room + room // 16 + 65536)
            present = self._present(idx, s, win)
            cand = present == 0
            ncand = int(cand.sum().item())
            if ncand == 0:  # the whole window is already reported: counted, not inserted
                self._added_elements += win
                s += win
                continue
            if room == 0:
                # the newest filter is full: it is replaced when the first key that really has to be inserted comes
                j = int(torch.argmax(cand.to(torch.uint8)).item())
                self._added_elements += j
                s += j
                self._grow()
                continue
            if room is None or ncand <= room:
                e = win
            else:  # end the chunk right after the room-th candidate: no growth can happen inside it
                csum = torch.cumsum(cand, 0)
                e = int(torch.searchsorted(csum, torch.tensor([room], device=csum.device, dtype=csum.dtype)).item()) + 1
            inserted = self._resolve_insert(last, idx, s, e, present)
            last._els_added += inserted
            self._added_elements += e
            self.last_batch_stats["chunks"] += 1
            s += e

    def _batches(self, keys, prehashed: bool):
        """cut an arbitrary key container into windows of device-ready batches"""
        first = self._blooms[0]
        if prehashed:
            yield pack_hashes(keys, self._k)
            return
        if isinstance(keys, (str, bytes, bytearray, memoryview)):
            keys = [keys]
        n = len(keys)
        if n <= _WINDOW:
            yield first._batch(keys)
            return
        for w0 in range(0, n, _WINDOW):
            yield first._batch(keys[w0:w0 + _WINDOW])

    # ------------------------------------------------------------------ public API (expandingbloom.py:132-170)
    def add_many(self, keys, force: bool = False) -> None:
        """ordered batch insert with the reference's per-key semantics (see the module docstring)"""
        self.last_batch_stats = {"chunks": 0}
        for b in self._batches(keys, False):
            self._blooms[0]._tab.check_batch(b)
            self._add_batch(b, force)

    def add_alt_many(self, hashes, force: bool = False) -> None:
        self.last_batch_stats = {"chunks": 0}
        for b in self._batches(hashes, True):
            self._add_batch(b, force)

    # Single-key calls (the reference's interface: expandingbloom.py:132-158) on a shallow stack go filter by filter through the engine's
    # one-key path -- one launch and a polled completion mailbox each (NOTES.md 3.5) -- instead of the batch machinery (hash once, test every
    # filter, read the verdict back: three launches and two stream waits for one key); deeper stacks hash once for all filters.
    _ONE_KEY_FILTERS = 4

    def _one(self, key):
        """the bytes of ONE key the engine hashes itself (``BloomFilter._one_key``), or None: take the batch path"""
        return self._blooms[-1]._one_key(key) if len(self._blooms) <= self._ONE_KEY_FILTERS else None

    def _reported(self, raw: bytes) -> bool:
        """expandingbloom.py:140-147 for one key: does ANY filter report it"""
        L = N.lib()
        for f in self._blooms:
            t = f._tab
            N.check(L.psk_bloom_check(t.handle, N.KEYS_FIXED, raw or None, None, 1, len(raw), N.HOST, t.one.o_addr, t.stream))
            if t.one.o_u8[0]:
                return True
        return False

    def add(self, key: KeyT, force: bool = False) -> None:
        """expandingbloom.py:149-158"""
        raw = self._one(key)
        if raw is None:
            self.add_many([key], force)
            return
        self.last_batch_stats = {"chunks": 0}
        if not force and self._reported(raw):  # :154: counted, not inserted
            self._added_elements += 1
            return
        last = self._blooms[-1]
        if self._room(last) == 0:
            self._grow()  # (a rotation may drop the oldest filter: the key is still absent afterwards)
            last = self._blooms[-1]
        t = last._tab
        N.check(N.lib().psk_bloom_add(t.handle, N.KEYS_FIXED, raw or None, None, 1, len(raw), N.HOST, t.stream))
        last._els_added += 1
        self._added_elements += 1

    def add_alt(self, hashes: HashResultsT, force: bool = False) -> None:
        """expandingbloom.py:160-170"""
        self.add_alt_many(hashes, force)

    def _check_batch(self, b: KeyBatch):
        if b.n == 0:
            return np.zeros(0, dtype=np.bool_) if b.where != N.DEVICE else torch.zeros(0, dtype=torch.bool, device=f"cuda:{self._dev()}")
        idx = self._indices(b)
        res = self._present(idx, 0, b.n).clone()
        if b.where == N.DEVICE:
            return res.view(torch.bool)
        return res.cpu().numpy().view(np.bool_)

    def check_many(self, keys):
        """membership of every key in ANY filter of the stack: numpy bool[n] (host keys) / torch bool[n] (device keys)"""
        outs = [self._check_batch(b) for b in self._batches(keys, False)]
        if len(outs) == 1:
            return outs[0]
        return torch.cat(outs) if torch.is_tensor(outs[0]) else np.concatenate(outs)

    def check_alt_many(self, hashes):
        return self._check_batch(pack_hashes(hashes, self._k))

    def check(self, key: KeyT) -> bool:
        """expandingbloom.py:132-138"""
        raw = self._one(key)
        return self._reported(raw) if raw is not None else bool(self.check_many([key])[0])

    def check_alt(self, hashes: HashResultsT) -> bool:
        """expandingbloom.py:140-147"""
        return bool(self.check_alt_many(hashes)[0])

    # ------------------------------------------------------------------ byte format (expandingbloom.py:186-262)
    def _export_parts(self):
        for blm in self._blooms:
            yield self._COUNT.pack(blm.elements_added)
            yield blm._table_bytes()
        yield self._FOOTER.pack(len(self._blooms), self.estimated_elements, self.elements_added, self.false_positive_rate)

    def export(self, file) -> None:
        """per filter: uint64 elements_added + the raw bit array; then the ``QQQf`` footer"""
        if hasattr(file, "write"):
            for part in self._export_parts():
                file.write(part)
        else:
            with open(Path(file).expanduser().resolve(), "wb") as fp:
                self.export(fp)

    @classmethod
    def _parse_footer(cls, b):
        raw = bytes(b[-cls._FOOTER.size:])
        size, est_els, els_added, fpr = cls._FOOTER.unpack(raw)
        return int(size), int(est_els), int(els_added), float(fpr)

    def _parse_blooms(self, b, size: int) -> None:
        self._blooms = []
        self._scratch = {}
        start = 0
        for _ in range(size):
            blm = self._new_filter()
            end = start + self._COUNT.size + blm.bloom_length
            blm._els_added = int(self._COUNT.unpack(bytes(b[start:start + self._COUNT.size]))[0])
            blm._tab.write(bytes(b[start + self._COUNT.size:end]))
            self._blooms.append(blm)
            start = end

    def _load(self, blob: bytes) -> None:
        size, est_els, els_added, fpr = self._parse_footer(blob)
        self._added_elements = els_added
        self._fpr = fpr
        self._est_elements = est_els
        self._parse_blooms(blob, size)


class RotatingBloomFilter(ExpandingBloomFilter):
    """Stack of at most ``max_queue_size`` filters; the oldest is dropped when a new one is needed
    (expandingbloom.py:265-361)."""

    def __init__(self, est_elements: int | None = None, false_positive_rate: float | None = None, max_queue_size: int = 10,
                 filepath: str | Path | None = None, hash_function: HashFuncT | None = None, device=None):
        super().__init__(est_elements=est_elements, false_positive_rate=false_positive_rate, filepath=filepath,
                         hash_function=hash_function, device=device)
        self._queue_size = max_queue_size

    @classmethod
    def frombytes(cls, b, max_queue_size: int, hash_function: HashFuncT | None = None, device=None) -> "RotatingBloomFilter":
        """expandingbloom.py:299-318"""
        size, est_els, added_els, fpr = cls._parse_footer(b)
        blm = cls(est_elements=est_els, false_positive_rate=fpr, max_queue_size=max_queue_size, hash_function=hash_function,
                  device=device)
        blm._parse_blooms(b, size)
        blm._added_elements = added_els
        return blm

    @property
    def max_queue_size(self) -> int:
        return self._queue_size

    @property
    def current_queue_size(self) -> int:
        return len(self._blooms)

    def _room(self, blm: BloomFilter):
        """expandingbloom.py:346-347: rotates only when ``elements_added == estimated_elements`` (a filter loaded
        with a larger count never rotates again: unbounded room)"""
        c, est = blm.elements_added, blm.estimated_elements
        if c == est:
            return 0
        if c < est:
            return math.ceil(est) - c if float(est).is_integer() else None
        return None

    def _grow(self) -> None:
        self._rotate(force=False, ready=True)

    def _rotate(self, force: bool, ready: bool) -> None:
        """expandingbloom.py:343-358"""
        no_need_to_pop = self.current_queue_size < self._queue_size
        if force or ready:
            if not no_need_to_pop:
                self._blooms.pop(0)
            self._add_bloom_filter()

    def pop(self) -> None:
        """expandingbloom.py:333-341"""
        if self.current_queue_size == 1:
            raise RotatingBloomFilterError("Popping a Bloom Filter will result in an unusable system!")
        self._blooms.pop(0)

    def push(self) -> None:
        """expandingbloom.py:343-345"""
        self._rotate(force=True, ready=False)
