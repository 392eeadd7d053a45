"""BloomFilter with the bit array resident in MI355X HBM and add/check as fused HIP kernels.

Drop-in for the hot path of the reference ``probables.BloomFilter`` (``probables/blooms/bloom.py``):
same constructor keywords, ``add`` / ``check`` / ``in``, ``add_alt`` / ``check_alt``, ``hashes`` and the
``hash_function`` plugin, the same export byte formats -- plus the batch methods the reference lacks
(``add_many`` / ``check_many``), which are the point of the exercise.
"""

from __future__ import annotations

import math
import struct
from array import array
from binascii import hexlify, unhexlify
from io import IOBase
from mmap import mmap
from numbers import Number
from pathlib import Path
from textwrap import wrap

import numpy as np

from . import _native as N
from ._base import DeviceTable
from .exceptions import InitializationError, SimilarityError
from .hashes import HashFuncT, HashResultsT, KeyT, default_fnv_1a, device_digest, is_fused_fnv
from .keys import KeyBatch, digest_batch, one_key_bytes, pack_hashes, pack_keys

_LN2_SQUARED = 0.4804530139182   # bloom.py:477 (the literal the reference and its C sibling use)
_LN2 = 0.6931471805599453        # bloom.py:478

_FOOTER = struct.Struct("QQf")      # est_elements, elements_added, fpr  (bloom.py:108)
_FOOTER_BE = struct.Struct(">QQf")  # hex export is big-endian (bloom.py:109)


def _is_hex(text) -> bool:
    return text is not None and all(c in "0123456789abcdefABCDEF" for c in text)


def _existing_file(path) -> bool:
    return path is not None and Path(path).exists()


class BloomFilter:
    """Bloom filter on the GPU.

    Args (identical to the reference, bloom.py:69-76):
        est_elements, false_positive_rate, filepath, hex_string, hash_function
    Extra:
        device: HIP device index (default: torch's current device)

    Initialisation order, as in the reference (bloom.py:47-51): file, then hex string, then parameters.
    """

    _TYPE = "regular"
    _KIND = "bloom"
    _ELEM = struct.Struct("B")
    _MISMATCH = "The parameter second must be of type BloomFilter or a BloomFilterOnDisk"

    def __init__(self, est_elements=None, false_positive_rate=None, filepath=None, hex_string=None,
                 hash_function: HashFuncT | None = None, device=None):
        self._dev_arg = device
        self._els_added = 0
        self._pending: list = []
        self._tab: DeviceTable | None = None
        if _existing_file(filepath):
            self._load(Path(filepath).expanduser().resolve().read_bytes(), hash_function)
        elif _is_hex(hex_string):
            self._load_hex(hex_string, hash_function)
        else:
            if est_elements is None or false_positive_rate is None:
                raise InitializationError(self._insufficient_msg())
            fpr, n_hashes, n_bits = self._get_optimized_params(est_elements, false_positive_rate)
            self._configure(est_elements, fpr, n_hashes, n_bits, hash_function)

    @staticmethod
    def _insufficient_msg() -> str:
        return "Insufecient parameters to set up the Bloom Filter"  # (sic) bloom.py:101

    # ------------------------------------------------------------------ sizing (bloom.py:463-502)
    @classmethod
    def _get_optimized_params(cls, estimated_elements, false_positive_rate):
        """(n, p) -> (p as float32, k, m).  Host arithmetic, once per filter."""
        if not (isinstance(estimated_elements, Number) and estimated_elements > 0):
            raise InitializationError("Bloom: estimated elements must be greater than 0")
        if not (isinstance(false_positive_rate, Number) and 0.0 <= false_positive_rate < 1.0):
            raise InitializationError("Bloom: false positive rate must be between 0.0 and 1.0")
        # the reference rounds p through a C float "to mimic the c version" (bloom.py:474-475)
        p32 = struct.unpack("f", struct.pack("f", float(false_positive_rate)))[0]
        n_bits = math.ceil((-estimated_elements * math.log(p32)) / _LN2_SQUARED)
        n_hashes = int(round(_LN2 * n_bits / estimated_elements))
        if n_hashes == 0:
            raise InitializationError("Bloom: Number hashes is zero; unusable parameters provided")
        return p32, n_hashes, n_bits

    def _table_len(self, n_bits: int) -> int:
        return math.ceil(n_bits / 8.0)  # bloom.py:495 with 8 bits per element

    def _configure(self, est_els, fpr, n_hashes, n_bits, hash_func):
        self._est_elements = est_els
        self._fpr = fpr
        self._number_hashes = int(n_hashes)
        self._num_bits = int(n_bits)
        self._bloom_length = self._table_len(n_bits)
        self._hash_func = default_fnv_1a if hash_func is None else hash_func
        self._is_fused = is_fused_fnv(hash_func)          # decided once: ours, None, or the reference's own default_fnv_1a
        self._digest = device_digest(self._hash_func)      # default_md5 / default_sha256 (ours or the reference's)
        self._els_added = 0
        self._tab = DeviceTable(self._KIND, self._num_bits, self._number_hashes, self._dev_arg)

    # ------------------------------------------------------------------ properties (bloom.py:141-214)
    @property
    def false_positive_rate(self) -> float:
        return self._fpr

    @property
    def estimated_elements(self) -> int:
        return self._est_elements

    @property
    def number_hashes(self) -> int:
        return self._number_hashes

    @property
    def number_bits(self) -> int:
        return self._num_bits

    @property
    def elements_added(self) -> int:
        return self._els_added

    @elements_added.setter
    def elements_added(self, val: int):
        self._els_added = val

    @property
    def is_on_disk(self) -> bool:
        return False

    @property
    def bloom_length(self) -> int:
        return self._bloom_length

    @property
    def bloom(self) -> array:
        """host SNAPSHOT of the table as the reference's ``array`` type (the live table is in HBM)"""
        self._flush()
        return array(self._ELEM.format, self._tab.read().tobytes())

    @property
    def hash_function(self) -> HashFuncT:
        return self._hash_func

    @property
    def device(self) -> int:
        return self._tab.device

    @property
    def table_tensor(self):
        """the torch int32 tensor backing the table (padded to 16 B); what the multi-GPU merge reduces.  Handing it out tells the
        engine that the table may be written from outside, now or later (``psk_table_info``): anything it derived from the table
        is dropped, and nothing of the kind is kept again until :meth:`table_released` says the holder is done."""
        self._flush()
        _ = self._tab.exposed_ptr
        return self._tab.tensor

    def table_released(self) -> None:
        """the caller no longer writes through a tensor obtained from :attr:`table_tensor` (it takes the property again before any
        later write): the engine may keep state derived from the table again -- e.g. the 4-bit images of repeated
        CountingBloomFilter lookups"""
        self._flush()
        self._tab.written()

    def set_engine_option(self, name: str, value) -> None:
        """override an engine tunable for THIS sketch (``psk_sketch_set_option``: "partition_min_keys", "cbf_lookup_shadow",
        "auto_combine", "update_window", "update_window_keys", "scratch_budget_bytes", "remove_exact", "bloom_lookup"); ``None`` =
        follow the process-wide default (``_native.set_option``) again"""
        self._tab.set_option(name, value)

    def get_engine_option(self, name: str) -> int:
        return self._tab.get_option(name)

    def scratch_bytes(self) -> dict:
        """device memory the engine holds for this sketch besides its table (``psk_scratch_bytes``): ``total``, of which
        ``waiting_updates`` (update window / write-combining lists) and ``kept_images`` (4-bit slice images of an unchanged table)"""
        return self._tab.scratch_bytes()

    def release_scratch(self) -> None:
        """apply what is waiting and free the engine's scratch for this sketch (it regrows on demand)"""
        self._flush()
        self._tab.release_scratch()

    @property
    def _fused(self) -> bool:
        """True when the kernel computes the hashes itself (default FNV-1a family)"""
        return self._is_fused

    # ------------------------------------------------------------------ working set (bloom.py:216-272)
    def clear(self) -> None:
        self._pending = []
        self._els_added = 0
        self._tab.clear()

    # Per-key ``add`` calls (the reference's only insert API, e.g. ``for w in words: blm.add(w)``) are write-combined
    # on the host and reach the GPU as ONE batch: a Bloom insert returns nothing and commutes with every other
    # insert, so deferring it is unobservable.  Everything that reads the table flushes first.
    _PENDING_LIMIT = 1 << 16

    def _flush(self) -> None:
        if self._pending:
            b = self._batch(self._pending)
            N.check(N.lib().psk_bloom_add(self._tab.handle, *b.args(), b.where, self._tab.stream))
            self._pending = []  # only once the engine has taken them: a failing flush must not lose counted keys

    def hashes(self, key: KeyT, depth: int | None = None) -> HashResultsT:
        """the plugin call site (bloom.py:223-232)"""
        return self._hash_func(key, self._number_hashes if depth is None else depth)

    def _batch(self, keys) -> KeyBatch:
        """keys -> device-ready batch; a custom hash_function is evaluated here, on the host, per key"""
        if self._fused:
            b = pack_keys(keys)
        elif self._digest is not None:  # default_md5 / default_sha256: digest chains on the GPU
            b = digest_batch(keys, self._digest, self._number_hashes, self._tab.device, self._tab.stream)
        else:
            if isinstance(keys, (str, bytes, bytearray, memoryview)):
                keys = [keys]
            b = pack_hashes([self._hash_func(k, self._number_hashes) for k in keys], self._number_hashes) \
                if len(keys) else pack_hashes(np.zeros((0, self._number_hashes), dtype=np.uint64), self._number_hashes)
        self._tab.check_batch(b)
        return b

    def _add_batch(self, b: KeyBatch) -> None:
        N.check(N.lib().psk_bloom_add(self._tab.handle, *b.args(), b.where, self._tab.stream))
        self._els_added += b.n  # bloom.py:250, once per key

    def _check_batch(self, b: KeyBatch):
        self._flush()
        addr, fin = self._tab.out_buffer(b, b.n, np.uint8, _torch_dtype("uint8"))
        N.check(N.lib().psk_bloom_check(self._tab.handle, *b.args(), b.where, addr, self._tab.stream))
        res = fin()
        return res.view(np.bool_) if isinstance(res, np.ndarray) else res.view(_torch_dtype("bool"))

    def add(self, key: KeyT) -> None:
        """bloom.py:234-239; write-combined on the host (see ``_flush``)"""
        if not isinstance(key, (str, bytes, bytearray, memoryview)):
            raise TypeError(f"keys must be str or bytes-like, got {type(key).__name__}")
        self._pending.append(bytes(key) if isinstance(key, (bytearray, memoryview)) else key)
        self._els_added += 1  # bloom.py:250
        if len(self._pending) >= self._PENDING_LIMIT:
            self._flush()

    def add_alt(self, hashes: HashResultsT) -> None:
        """bloom.py:241-250: insert the element represented by its hashes"""
        self._add_batch(pack_hashes(hashes, self._number_hashes))

    def _one_key(self, key):
        """a single key for the value-returning per-key calls: its bytes when the engine hashes it itself (``_base.OneKey``), else None"""
        return one_key_bytes(key) if self._is_fused else None

    def check(self, key: KeyT) -> bool:
        """bloom.py:252-259"""
        raw = self._one_key(key)
        if raw is None:
            return bool(self._check_batch(self._batch(key))[0])
        if self._pending:
            self._flush()
        t = self._tab
        one = t.one
        N.check(N.lib().psk_bloom_check(t.handle, N.KEYS_FIXED, raw or None, None, 1, len(raw), N.HOST, one.o_addr, t.stream))
        return bool(one.o_u8[0])

    def check_alt(self, hashes: HashResultsT) -> bool:
        """bloom.py:261-272"""
        return bool(self._check_batch(pack_hashes(hashes, self._number_hashes))[0])

    def __contains__(self, key: KeyT) -> bool:
        return self.check(key)

    # ------------------------------------------------------------------ batch API (new)
    def add_many(self, keys) -> None:
        """insert a whole batch with ONE kernel launch.  ``keys``: sequence of str/bytes, an (n, L) uint8
        numpy array, or an (n, L) uint8 torch tensor (CUDA tensors are consumed in place, asynchronously)."""
        self._add_batch(self._batch(keys))

    def check_many(self, keys):
        """membership of every key: numpy bool[n] (host input) or torch bool[n] on the device (device input)"""
        return self._check_batch(self._batch(keys))

    def check_many_begin(self, keys) -> None:
        """first half of a split lookup: hash + partition a device-resident batch WITHOUT reading the table (so it can
        run while a multi-GPU merge of the table is in flight, see ``parallel.merge_bloom_async``).  Finish with
        :meth:`check_many_finish`; the keys must stay alive and unchanged until then."""
        self._flush()
        if getattr(self, "_split", None) is not None:
            raise RuntimeError("a split lookup is already pending on this filter")
        b = self._batch(keys)
        if b.where == N.DEVICE:
            N.check(N.lib().psk_bloom_check_begin(self._tab.handle, *b.args(), self._tab.stream))
            self._split = ("engine", b)
        else:  # host batches gain nothing from the split: looked up at finish time
            self._split = ("late", b)

    def check_many_finish(self):
        """second half: membership of the batch given to :meth:`check_many_begin`, against the table as it is NOW"""
        if getattr(self, "_split", None) is None:
            raise RuntimeError("check_many_finish() without a pending check_many_begin()")
        kind, b = self._split
        self._split = None
        if kind == "late":
            return self._check_batch(b)
        self._flush()  # add() calls made since begin: the answer is against the table as it is NOW
        out = torch_mod().empty(b.n, dtype=_torch_dtype("uint8"), device=f"cuda:{self._tab.device}")
        N.check(N.lib().psk_bloom_check_finish(self._tab.handle, out.data_ptr(), self._tab.stream))
        return out.view(_torch_dtype("bool"))

    def add_alt_many(self, hashes) -> None:
        """pre-hashed batch: (n, >=k) uint64"""
        self._add_batch(pack_hashes(hashes, self._number_hashes))

    def check_alt_many(self, hashes):
        return self._check_batch(pack_hashes(hashes, self._number_hashes))

    def check_many_bits(self, keys):
        """membership as a ballot bitmap (bit i&63 of word i>>6) plus the number of hits"""
        self._flush()
        b = self._batch(keys)
        nwords = (b.n + 63) // 64
        if b.where == N.DEVICE:
            import torch  # noqa: PLC0415

            bits = torch.zeros(nwords, dtype=torch.int64, device=f"cuda:{self._tab.device}")
            hits = torch.zeros(1, dtype=torch.int64, device=f"cuda:{self._tab.device}")
            N.check(N.lib().psk_bloom_check_bits(self._tab.handle, *b.args(), b.where, bits.data_ptr(), hits.data_ptr(), self._tab.stream))
            return bits, hits
        bits = np.zeros(nwords, dtype=np.uint64)
        hits = np.zeros(1, dtype=np.uint64)
        N.check(N.lib().psk_bloom_check_bits(self._tab.handle, *b.args(), b.where, bits.ctypes.data if nwords else None,
                                             hits.ctypes.data, self._tab.stream))
        return bits, int(hits[0])

    def synchronize(self) -> None:
        self._flush()
        self._tab.synchronize()

    # ------------------------------------------------------------------ export / import (bloom.py:274-338, 504-550)
    def _footer(self, st: struct.Struct) -> bytes:
        return st.pack(self.estimated_elements, self.elements_added, self.false_positive_rate)

    def _table_bytes(self) -> bytes:
        self._flush()
        return self._tab.read().tobytes()

    def __bytes__(self) -> bytes:
        return self._table_bytes() + self._footer(_FOOTER)

    def export_hex(self) -> str:
        return str(hexlify(self._table_bytes()) + hexlify(self._footer(_FOOTER_BE)), "utf-8")

    def export(self, file) -> None:
        """raw table + ``QQf`` footer, byte-compatible with the reference and the author's C library"""
        if isinstance(file, (IOBase, mmap)):
            file.write(bytes(self))
        else:
            Path(file).expanduser().resolve().write_bytes(bytes(self))

    def export_c_header(self, filename) -> None:
        """bloom.py:306-322"""
        body = ("  " + line for line in wrap(", ".join(f"0x{e:02x}" for e in bytearray.fromhex(self.export_hex())), 80))
        what = "standard BloomFilter" if self._TYPE in ("regular", "regular-on-disk") else "CountingBloomFilter"
        with open(filename, "w", encoding="utf-8") as fh:
            print(f"/* BloomFilter Export of a {what} */", file=fh)
            print("#include <inttypes.h>", file=fh)
            print("const uint64_t estimated_elements = ", self.estimated_elements, ";", sep="", file=fh)
            print("const uint64_t elements_added = ", self.elements_added, ";", sep="", file=fh)
            print("const float false_positive_rate = ", self.false_positive_rate, ";", sep="", file=fh)
            print("const uint64_t number_bits = ", self.number_bits, ";", sep="", file=fh)
            print("const unsigned int number_hashes = ", self.number_hashes, ";", sep="", file=fh)
            print("const unsigned char bloom[] = {", *body, "};", sep="\n", file=fh)

    def export_size(self) -> int:
        return self.bloom_length * self._ELEM.size + _FOOTER.size

    @classmethod
    def frombytes(cls, b, hash_function: HashFuncT | None = None, device=None):
        inst = cls.__new__(cls)
        inst._dev_arg = device
        inst._els_added = 0
        inst._pending = []
        inst._tab = None
        inst._load(bytes(b), hash_function)
        return inst

    @classmethod
    def _parse_footer(cls, st: struct.Struct, raw: bytes):
        est, added, fpr = st.unpack_from(bytes(raw))
        fpr, n_hashes, n_bits = cls._get_optimized_params(est, float(fpr))
        return int(est), int(added), float(fpr), int(n_hashes), int(n_bits)

    def _load(self, blob: bytes, hash_function=None) -> None:
        est, added, fpr, n_hashes, n_bits = self._parse_footer(_FOOTER, blob[-_FOOTER.size:])
        self._configure(est, fpr, n_hashes, n_bits, hash_function)
        self._tab.write(blob[: self._ELEM.size * self.bloom_length])
        self._els_added = added

    def _load_hex(self, hex_string: str, hash_function=None) -> None:
        cut = _FOOTER_BE.size * 2
        est, added, fpr, n_hashes, n_bits = self._parse_footer(_FOOTER_BE, unhexlify(hex_string[-cut:]))
        self._configure(est, fpr, n_hashes, n_bits, hash_function)
        self._tab.write(unhexlify(hex_string[:-cut]))
        self._els_added = added

    # ------------------------------------------------------------------ statistics (bloom.py:117-133, 340-369)
    def _cnt_number_bits_set(self) -> int:
        self._flush()
        return self._tab.popcount()  # device popcount kernel

    def estimate_elements(self) -> int:
        setbits = self._cnt_number_bits_set()
        if setbits >= self.number_bits:
            return -1
        log_n = math.log(1 - (float(setbits) / float(self.number_bits)))
        return int(-1 * (float(self.number_bits) / float(self.number_hashes)) * log_n)

    def current_false_positive_rate(self) -> float:
        dbl = (self.number_hashes * -1 * self.elements_added) / self.number_bits
        return math.pow((1 - math.exp(dbl)), self.number_hashes)

    def __str__(self) -> str:
        return (
            "BloomFilter:\n"
            f"\tbits: {self.number_bits}\n"
            f"\testimated elements: {self.estimated_elements}\n"
            f"\tnumber hashes: {self.number_hashes}\n"
            f"\tmax false positive rate: {self.false_positive_rate:.6f}\n"
            f"\tbloom length (8 bits): {self.bloom_length}\n"
            f"\telements added: {self.elements_added}\n"
            f"\testimated elements added: {self.estimate_elements()}\n"
            f"\tcurrent false positive rate: {self.current_false_positive_rate():.6f}\n"
            f"\texport size (bytes): {self.export_size()}\n"
            f"\tnumber bits set: {self._cnt_number_bits_set()}\n"
            "\tis on disk: no\n"
        )

    # ------------------------------------------------------------------ set algebra (bloom.py:371-460)
    def _verify_bloom_similarity(self, second) -> bool:
        return not (
            self.number_hashes != second.number_hashes
            or self.number_bits != second.number_bits
            or self.hashes("test") != second.hashes("test")
        )

    def _require_similar(self, second, msg="Bloom Filters are not similar"):
        if not isinstance(second, BloomFilter) or type(second)._KIND != self._KIND:
            raise TypeError(self._MISMATCH)
        if self._verify_bloom_similarity(second) is False:
            raise SimilarityError(msg)
        if second._tab.device != self._tab.device:
            raise ValueError("set operations need both filters on the same device")

    def _combine(self, second, fn_name: str):
        self._flush()
        second._flush()
        res = type(self)(self.estimated_elements, self.false_positive_rate, hash_function=self.hash_function,
                         device=self._tab.device)
        L, t = N.lib(), res._tab
        N.check(L.psk_table_or(t.ptr, self._tab.ptr, t.nwords, t.device, t.stream))  # res = self (res starts empty)
        N.check(getattr(L, fn_name)(t.ptr, second._tab.ptr, t.nwords, t.device, t.stream))
        res.elements_added = res.estimate_elements()
        return res

    def union(self, second):
        """bytewise OR as one streaming kernel (bloom.py:401-428)"""
        self._require_similar(second)
        return self._combine(second, "psk_table_or")

    def intersection(self, second):
        """bytewise AND (bloom.py:371-399)"""
        self._require_similar(second)
        return self._combine(second, "psk_table_and")

    def jaccard_index(self, second) -> float:
        """popcount(AND) / popcount(OR) (bloom.py:430-460)"""
        self._require_similar(second)
        cu = self._combine(second, "psk_table_or")._cnt_number_bits_set()
        if cu == 0:
            return 1.0
        return self._combine(second, "psk_table_and")._cnt_number_bits_set() / cu


def torch_mod():
    import torch  # noqa: PLC0415

    return torch


def _torch_dtype(name: str):
    import torch  # noqa: PLC0415

    return getattr(torch, name)
