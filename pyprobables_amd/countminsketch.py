"""CountMinSketch (+ Count-Mean / Count-Mean-Min query variants) with the width x depth int32 table in
HBM and add / remove / check as fused HIP kernels.

Drop-in for the hot path of ``probables.CountMinSketch`` (``probables/countminsketch/countminsketch.py``).
Single-key ``add`` / ``remove`` run the reference semantics literally (ordered kernel: exact return value,
exact int32 / int64 clamps); ``add_many`` / ``remove_many`` are unordered atomic batches whose final table is
bit-exact whenever it does not depend on the order (same-sign weights, or no bin touching a rail).
"""

from __future__ import annotations

import math
import struct
from io import BytesIO, IOBase
from mmap import mmap
from numbers import Number
from pathlib import Path

import numpy as np

from . import _native as N
from ._base import DeviceTable, weights_arg
from .bloom import _existing_file, _torch_dtype
from .exceptions import CountMinSketchError, InitializationError
from .hashes import HashFuncT, HashResultsT, KeyT, default_fnv_1a, device_digest, is_fused_fnv
from .keys import KeyBatch, digest_batch, one_key_bytes, pack_hashes, pack_keys

_I32_MAX, _I32_MIN = 2**31 - 1, -(2**31)
_I64_MAX, _I64_MIN = 2**63 - 1, -(2**63)
_FOOTER = struct.Struct("IIq")  # width, depth, elements_added (countminsketch.py:122)
_QUERIES = {"min": N.Q_MIN, "mean": N.Q_MEAN, "mean-min": N.Q_MEANMIN}


class CountMinSketch:
    """Count-Min sketch on the GPU.

    Args (identical to the reference, countminsketch.py:59-67):
        width, depth, confidence, error_rate, filepath, hash_function
    Extra: device.  Initialisation order: file, then width/depth, then confidence/error_rate."""

    _DEFAULT_QUERY = "min"

    def __init__(self, width=None, depth=None, confidence=None, error_rate=None, filepath=None,
                 hash_function: HashFuncT | None = None, device=None):
        self._dev_arg = device
        self._els_added = 0
        self._dirty = False
        self._query = self._DEFAULT_QUERY
        self._tab: DeviceTable | None = None
        self._hash_function = default_fnv_1a if hash_function is None else hash_function
        self._is_fused = is_fused_fnv(hash_function)
        self._digest = device_digest(self._hash_function)
        if filepath is not None and _existing_file(filepath):
            self._parse_bytes(Path(filepath).expanduser().resolve().read_bytes())
            return
        if width is not None and depth is not None:
            if not (isinstance(width, Number) and width > 0 and isinstance(depth, Number) and depth > 0):
                raise InitializationError("CountMinSketch: width and depth must be greater than 0")
            self._width, self._depth = int(width), int(depth)
            self._confidence = 1 - (1 / math.pow(2, self._depth))
            self._error_rate = 2 / self._width
        elif confidence is not None and error_rate is not None:
            if not (isinstance(confidence, Number) and confidence > 0 and isinstance(error_rate, Number) and error_rate > 0):
                raise InitializationError("CountMinSketch: width and depth must be greater than 0")
            self._confidence, self._error_rate = confidence, error_rate
            self._width = math.ceil(2 / error_rate)                                   # countminsketch.py:102
            self._depth = math.ceil((-1 * math.log(1 - confidence)) / 0.6931471805599453)  # :103-104
        else:
            raise InitializationError(
                "Must provide one of the following to initialize the Count-Min Sketch:\n"
                "    A file to load,\n"
                "    The width and depth,\n"
                "    OR confidence and error rate"
            )
        self._tab = DeviceTable("cms", self._width, self._depth, self._dev_arg)

    # ------------------------------------------------------------------ properties
    @property
    def width(self) -> int:
        return self._width

    @property
    def depth(self) -> int:
        return self._depth

    @property
    def confidence(self) -> float:
        return self._confidence

    @property
    def error_rate(self) -> float:
        return self._error_rate

    def _fold_counters(self) -> None:
        if self._tab is None or not self._dirty:
            return
        c = self._tab.counters()
        e = self._els_added + c[N.CTR_ADDED] - c[N.CTR_REMOVED]
        self._els_added = max(min(e, _I64_MAX), _I64_MIN)  # countminsketch.py:285-287, 317-319 (per batch)
        self._saturated = getattr(self, "_saturated", 0) + c[N.CTR_SATURATED]
        self._tab.reset_counters()
        self._dirty = False

    @property
    def elements_added(self) -> int:
        self._fold_counters()
        return self._els_added

    def batch_diagnostics(self) -> dict:
        self._dirty = True
        self._fold_counters()
        return {"saturated": getattr(self, "_saturated", 0)}

    @property
    def query_type(self) -> str:
        return self._query

    @query_type.setter
    def query_type(self, val):
        """'min' | 'mean' | 'mean-min'; anything else means 'min' (countminsketch.py:223-238)"""
        val = val.lower() if isinstance(val, str) else None
        self._query = val if val in ("mean", "mean-min") else "min"

    @property
    def hash_function(self) -> HashFuncT:
        return self._hash_function

    @property
    def device(self) -> int:
        return self._tab.device

    @property
    def table_tensor(self):
        return self._tab.tensor

    def set_engine_option(self, name: str, value) -> None:
        """override an engine tunable for THIS sketch (``psk_sketch_set_option``); ``None`` = follow the process-wide default again"""
        self._tab.set_option(name, value)

    def get_engine_option(self, name: str) -> int:
        return self._tab.get_option(name)

    def scratch_bytes(self) -> dict:
        """device memory the engine holds for this sketch besides its table (``psk_scratch_bytes``)"""
        return self._tab.scratch_bytes()

    def release_scratch(self) -> None:
        self._tab.release_scratch()

    @property
    def _bins(self):
        """host SNAPSHOT of the bins (int32, row-major by depth)"""
        from array import array  # noqa: PLC0415

        return array("i", self._tab.read().tobytes())

    @property
    def _fused(self) -> bool:
        return self._is_fused

    # ------------------------------------------------------------------ dunder / io
    def __str__(self) -> str:
        return (
            "Count-Min Sketch:\n"
            f"\tWidth: {self.width}\n"
            f"\tDepth: {self.depth}\n"
            f"\tConfidence: {self.confidence}\n"
            f"\tError Rate: {self.error_rate}\n"
            f"\tElements Added: {self.elements_added}"
        )

    def __contains__(self, key: KeyT) -> bool:
        return self.check(key) != 0

    def __bytes__(self) -> bytes:
        with BytesIO() as f:
            self.export(f)
            return f.getvalue()

    def export(self, file) -> None:
        """bins + ``IIq`` footer (countminsketch.py:342-354)"""
        blob = self._tab.read().tobytes() + _FOOTER.pack(self.width, self.depth, self.elements_added)
        if isinstance(file, (IOBase, mmap)):
            file.write(blob)
        else:
            Path(file).expanduser().resolve().write_bytes(blob)

    @classmethod
    def frombytes(cls, b, hash_function: HashFuncT | None = None, device=None):
        width, depth, _ = _FOOTER.unpack_from(bytes(b[-_FOOTER.size:]))
        inst = cls(width=width, depth=depth, hash_function=hash_function, device=device)
        inst._parse_bytes(bytes(b))
        return inst

    def _parse_bytes(self, blob: bytes) -> None:
        """countminsketch.py:417-427"""
        width, depth, added = _FOOTER.unpack_from(blob[-_FOOTER.size:])
        self._width, self._depth = width, depth
        self._confidence = 1 - (1 / math.pow(2, depth))
        self._error_rate = 2 / width
        if self._tab is None or (self._tab.m, self._tab.k) != (width, depth):
            self._tab = DeviceTable("cms", width, depth, self._dev_arg)
        self._tab.write(blob[: 4 * width * depth])
        self._els_added, self._dirty = added, False

    def clear(self) -> None:
        self._els_added, self._dirty = 0, False
        self._tab.clear()

    def hashes(self, key: KeyT, depth: int | None = None) -> HashResultsT:
        """the plugin call site (countminsketch.py:246-255)"""
        return self._hash_function(key, self.depth if depth is None else depth)

    # ------------------------------------------------------------------ batches
    def _batch(self, keys) -> KeyBatch:
        if self._fused:
            b = pack_keys(keys)
        elif self._digest is not None:  # default_md5 / default_sha256: digest chains on the GPU
            b = digest_batch(keys, self._digest, self._depth, self._tab.device, self._tab.stream)
        else:
            if isinstance(keys, (str, bytes, bytearray, memoryview)):
                keys = [keys]
            b = pack_hashes([self._hash_function(k, self._depth) for k in keys], self._depth) if len(keys) \
                else pack_hashes(np.zeros((0, self._depth), dtype=np.uint64), self._depth)
        self._tab.check_batch(b)
        return b

    def _alt(self, hashes) -> KeyBatch:
        b = pack_hashes(hashes, self._depth)
        if b.n and b.key_len != self._depth:
            # countminsketch.py:275 enumerates ALL supplied hashes; more than `depth` runs off the table there
            raise IndexError("array index out of range")
        return b

    def _ordered(self, b: KeyBatch, num_els, opmode: int) -> np.ndarray:
        if b.where != N.HOST:
            raise ValueError("ordered updates take host batches")
        w = np.ascontiguousarray(np.broadcast_to(np.asarray(num_els, dtype=np.int64), (b.n,)))
        out = np.empty(b.n + 1, dtype=np.int64)  # n return values + elements_added after the batch
        els_in = self.elements_added
        N.check(N.lib().psk_cms_update_ordered(self._tab.handle, *b.args(), w.ctypes.data if b.n else None, opmode,
                                               _QUERIES[self._query], els_in, b.where, out.ctypes.data, self._tab.stream))
        self._els_added = int(out[b.n])
        return out[: b.n]

    def _ordered_one(self, key, num_els, opmode: int):
        """one ordered update of one key through the preallocated words (``_base.OneKey``); None: take the general path"""
        raw = one_key_bytes(key) if self._is_fused else None
        if raw is None or type(num_els) is not int or not -(1 << 62) < num_els < 1 << 62:
            return None
        t = self._tab
        one = t.one
        one.w[0] = num_els
        N.check(N.lib().psk_cms_update_ordered(t.handle, N.KEYS_FIXED, raw or None, None, 1, len(raw), one.w_addr, opmode,
                                               _QUERIES[self._query], self.elements_added, N.HOST, one.o_addr, t.stream))
        self._els_added = int(one.o[1])
        return int(one.o[0])

    def add(self, key: KeyT, num_els: int = 1) -> int:
        """countminsketch.py:257-265"""
        res = self._ordered_one(key, num_els, N.OP_ADD)
        return res if res is not None else int(self._ordered(self._batch(key), num_els, N.OP_ADD)[0])

    def add_alt(self, hashes: HashResultsT, num_els: int = 1) -> int:
        """countminsketch.py:267-288"""
        return int(self._ordered(self._alt(hashes), num_els, N.OP_ADD)[0])

    def remove(self, key: KeyT, num_els: int = 1) -> int:
        """countminsketch.py:290-298"""
        res = self._ordered_one(key, num_els, N.OP_REMOVE)
        return res if res is not None else int(self._ordered(self._batch(key), num_els, N.OP_REMOVE)[0])

    def remove_alt(self, hashes: HashResultsT, num_els: int = 1) -> int:
        """countminsketch.py:300-321"""
        return int(self._ordered(self._alt(hashes), num_els, N.OP_REMOVE)[0])

    def update_ordered(self, keys, signed_num_els) -> np.ndarray:
        """strictly ordered mixed stream on the device: ``w >= 0`` adds, ``w < 0`` removes ``-w``; returns
        every op's reference return value (int64[n])"""
        return self._ordered(self._batch(keys), signed_num_els, N.OP_SIGNED)

    def _check_batch(self, b: KeyBatch):
        L = N.lib()
        if self._query == "mean-min":
            addr, fin = self._tab.out_buffer(b, b.n, np.int64, _torch_dtype("int64"))
            N.check(L.psk_cms_check_meanmin(self._tab.handle, *b.args(), b.where, self.elements_added, addr, self._tab.stream))
        else:
            addr, fin = self._tab.out_buffer(b, b.n, np.int32, _torch_dtype("int32"))
            N.check(L.psk_cms_check(self._tab.handle, *b.args(), b.where, _QUERIES[self._query], addr, self._tab.stream))
        return fin()

    def check(self, key: KeyT) -> int:
        """countminsketch.py:323-330"""
        raw = one_key_bytes(key) if self._is_fused else None
        if raw is None:
            return int(self._check_batch(self._batch(key))[0])
        t = self._tab
        one = t.one
        if self._query == "mean-min":
            N.check(N.lib().psk_cms_check_meanmin(t.handle, N.KEYS_FIXED, raw or None, None, 1, len(raw), N.HOST, self.elements_added,
                                                  one.o_addr, t.stream))
            return int(one.o[0])
        N.check(N.lib().psk_cms_check(t.handle, N.KEYS_FIXED, raw or None, None, 1, len(raw), N.HOST, _QUERIES[self._query], one.o_addr,
                                      t.stream))
        return int(one.o_i32[0])

    def check_alt(self, hashes: HashResultsT) -> int:
        """countminsketch.py:332-340"""
        return int(self._check_batch(self._alt(hashes))[0])

    def _update_batch(self, fn, b: KeyBatch, num_els) -> None:
        keep: list = []
        w_addr, _ = weights_arg(num_els, b.n, np.int32, b.where, keep, _I32_MIN, _I32_MAX, self._tab.device)
        N.check(fn(self._tab.handle, *b.args(), w_addr, b.where, self._tab.stream))
        self._dirty = True

    def add_many(self, keys, num_els=None) -> None:
        """ONE kernel launch for the whole batch; ``num_els``: None (=1), an int, or one int32 per key
        (numpy, or a CUDA tensor next to CUDA keys)"""
        self._update_batch(N.lib().psk_cms_add, self._batch(keys), num_els)

    def remove_many(self, keys, num_els=None) -> None:
        self._update_batch(N.lib().psk_cms_remove, self._batch(keys), num_els)

    def add_alt_many(self, hashes, num_els=None) -> None:
        self._update_batch(N.lib().psk_cms_add, self._alt(hashes), num_els)

    def remove_alt_many(self, hashes, num_els=None) -> None:
        self._update_batch(N.lib().psk_cms_remove, self._alt(hashes), num_els)

    def check_many(self, keys):
        """estimated count per key under the current ``query_type``"""
        return self._check_batch(self._batch(keys))

    def check_alt_many(self, hashes):
        return self._check_batch(self._alt(hashes))

    def synchronize(self) -> None:
        self._tab.synchronize()

    # ------------------------------------------------------------------ join (countminsketch.py:356-399)
    def join(self, second: "CountMinSketch") -> None:
        if not isinstance(second, CountMinSketch):
            raise TypeError(f"Unable to merge a count-min sketch with {type(second)}")
        if self.width != second.width or self.depth != second.depth or self.hashes("test") != second.hashes("test"):
            raise CountMinSketchError("Unable to merge as the count-min sketches are mismatched")
        if second._tab.device != self._tab.device:
            raise ValueError("join needs both sketches on the same device")
        t = self._tab
        N.check(N.lib().psk_table_add_sat_i32(t.ptr, second._tab.ptr, self.width * self.depth, t.device, t.stream))
        e = self.elements_added + second.elements_added
        self._els_added = max(min(e, _I64_MAX), _I64_MIN)
        N.check(N.lib().psk_rescan_bound(t.handle, t.stream))  # the joined table may sit on a rail


class CountMeanSketch(CountMinSketch):
    """default query 'mean' (countminsketch.py:456-491)"""

    _DEFAULT_QUERY = "mean"


class CountMeanMinSketch(CountMinSketch):
    """default query 'mean-min' (countminsketch.py:494-529)"""

    _DEFAULT_QUERY = "mean-min"
