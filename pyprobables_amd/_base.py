"""Device-side plumbing shared by the three sketches: one C-ABI handle + the HBM-resident table.

The table lives in a torch tensor (so it can be handed to RCCL through ``torch.distributed`` without a
copy) and is *borrowed* by the engine (``ext_table`` of ``psk_*_create``).  All work is enqueued on
torch's current HIP stream for that device.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .exceptions import NativeLibraryError
from .keys import KeyBatch

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _resolve_device(device) -> int:
    if device is None:
        if torch is not None and torch.cuda.is_available():
            return torch.cuda.current_device()
        return 0
    if torch is not None and isinstance(device, torch.device):
        return device.index or 0
    if isinstance(device, str):
        return int(device.split(":")[1]) if ":" in device else 0
    return int(device)


_raw_stream = getattr(getattr(torch, "_C", None), "_cuda_getCurrentRawStream", None) if torch is not None else None


class OneKey:
    """Argument and result words of the value-returning single-key calls (``key in blm``, ``cms.add(key)`` ...), allocated once per
    sketch: the reference's whole interface is per key (bloom.py:252, countminsketch.py:257, countingbloom.py:125), and a call that is one
    kernel launch + one wait should not spend more time building numpy arrays around its 16 bytes than on the device.  A handle is used
    by one thread at a time (include/psk.h), so is this."""

    __slots__ = ("w", "w_addr", "o", "o_addr", "o_u8", "o_i32", "o_u32")

    def __init__(self):
        self.w = np.zeros(1, dtype=np.int64)       # num_els of an ordered update
        self.w_addr = self.w.ctypes.data
        self.o = np.zeros(2, dtype=np.int64)       # the result (+ elements_added behind a CountMinSketch update)
        self.o_addr = self.o.ctypes.data
        self.o_u8, self.o_i32, self.o_u32 = self.o.view(np.uint8), self.o.view(np.int32), self.o.view(np.uint32)


class DeviceTable:
    """owns the psk handle and the table tensor"""

    def __init__(self, kind: str, m: int, k: int, device=None):
        L = N.lib()  # raises NativeLibraryError when the engine is missing
        self.kind, self.m, self.k = kind, int(m), int(k)
        self.device = _resolve_device(device)
        self.one = OneKey()
        self._cuda = torch is not None and torch.cuda.is_available()
        if N.device_count() == 0:
            raise NativeLibraryError(
                "no HIP device available: the sketch table lives in GPU memory and there is no CPU fallback"
            )
        if kind == "bloom":
            padded, create = L.psk_bloom_table_bytes(self.m), L.psk_bloom_create
            self.logical_bytes = (self.m + 7) // 8
        elif kind == "cbf":
            padded, create = L.psk_cbf_table_bytes(self.m), L.psk_cbf_create
            self.logical_bytes = 4 * self.m
        else:
            padded, create = L.psk_cms_table_bytes(self.m, self.k), L.psk_cms_create
            self.logical_bytes = 4 * self.m * self.k
        self.padded_bytes = int(padded)
        self.tensor = None
        ext = None
        if torch is not None and torch.cuda.is_available():
            self.tensor = torch.zeros(self.padded_bytes // 4, dtype=torch.int32, device=f"cuda:{self.device}")
            torch.cuda.current_stream(self.device).synchronize()
            ext = self.tensor.data_ptr()
        h = C.c_void_p()
        N.check(create(self.m, self.k, self.device, ext, C.byref(h)))
        self.handle = h
        if ext is not None:
            # the tensor is this object's own: nothing writes it behind the engine's back without saying so (`ptr` / `exposed_ptr`)
            N.check(L.psk_sketch_set_option(h, b"table_private", 1))

    # -- lifetime
    def close(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                N.lib().psk_destroy(h)
            except Exception:  # interpreter shutdown
                pass

    def __del__(self):
        self.close()

    # -- helpers
    @property
    def stream(self):
        if self._cuda:  # torch's CURRENT stream of the sketch's device, looked up per call (callers switch streams)
            if _raw_stream is not None:
                return _raw_stream(self.device) or None
            return torch.cuda.current_stream(self.device).cuda_stream or None
        return None

    def flush(self):
        """apply write-combined updates that are still waiting in the engine (CBF ``combine_updates``); a no-op otherwise"""
        N.check(N.lib().psk_flush(self.handle, self.stream))

    @property
    def ptr(self) -> int:
        """raw device pointer of the table for THIS package's own table kernels (pending write-combined updates are applied
        first; the engine is told that the table may change NOW -- not that somebody keeps the pointer)"""
        self.flush()
        if self.tensor is None:
            return self.exposed_ptr
        N.check(N.lib().psk_table_info(self.handle, None, None, None))
        return self.tensor.data_ptr()

    @property
    def exposed_ptr(self) -> int:
        """the pointer as handed to code outside this package: from here on the engine keeps nothing derived from the table
        (psk_table_info) until `written()` says the holder is done"""
        self.flush()
        p = C.c_void_p()
        N.check(N.lib().psk_table_info(self.handle, C.byref(p), None, None))
        return p.value

    def written(self):
        """the holder of a pointer / tensor handed out earlier has written what it wanted to (and takes the pointer again before it
        writes any more): bounds are rescanned, derived state may be kept again (psk_rescan_bound)"""
        N.check(N.lib().psk_rescan_bound(self.handle, self.stream))

    def set_option(self, name: str, value: int):
        """override an engine option for this sketch only (psk_sketch_set_option); value None: follow the process default"""
        N.check(N.lib().psk_sketch_set_option(self.handle, name.encode(), -(2**63) if value is None else int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int64(0)
        N.check(N.lib().psk_sketch_get_option(self.handle, name.encode(), C.byref(v)))
        return v.value

    @property
    def nwords(self) -> int:
        return self.padded_bytes // 4

    def check_batch(self, b: KeyBatch):
        if b.where == N.DEVICE and b.device is not None and b.device != self.device:
            raise ValueError(f"key batch lives on cuda:{b.device}, the sketch on cuda:{self.device}")

    def clear(self):
        N.check(N.lib().psk_clear(self.handle, self.stream))

    def read(self) -> np.ndarray:
        """the table in the reference's byte layout (uint8[logical_bytes])"""
        out = np.empty(self.logical_bytes, dtype=np.uint8)
        if out.size:
            N.check(N.lib().psk_read_table(self.handle, out.ctypes.data, out.size, self.stream))
        return out

    def write(self, raw: bytes | np.ndarray):
        a = np.frombuffer(bytes(raw), dtype=np.uint8) if not isinstance(raw, np.ndarray) else np.ascontiguousarray(raw).view(np.uint8)
        if a.size != self.logical_bytes:
            raise ValueError(f"table image has {a.size} bytes, expected {self.logical_bytes}")
        N.check(N.lib().psk_write_table(self.handle, a.ctypes.data, a.size, self.stream))

    def counters(self) -> list[int]:
        buf = (C.c_int64 * N.CTR_COUNT)()
        N.check(N.lib().psk_get_counters(self.handle, buf, self.stream))
        return list(buf)

    def reset_counters(self):
        N.check(N.lib().psk_reset_counters(self.handle, self.stream))

    def release_scratch(self):
        """free the engine's staging / partition buffers for this table (they regrow on the next large batch)"""
        N.check(N.lib().psk_release_scratch(self.handle))

    def scratch_bytes(self) -> dict:
        """device memory the engine holds for this table besides the table itself, in bytes: `total`, of which `waiting_updates` (update
        window / write-combining lists) and `kept_images` (4-bit slice images of an unchanged CountingBloomFilter table)"""
        out = (C.c_uint64 * 3)()
        N.check(N.lib().psk_scratch_bytes(self.handle, out))
        return {"total": int(out[0]), "waiting_updates": int(out[1]), "kept_images": int(out[2])}

    def synchronize(self):
        N.check(N.lib().psk_synchronize(self.handle, self.stream))

    def popcount(self) -> int:
        out = C.c_uint64(0)
        N.check(N.lib().psk_table_popcount(self.ptr, self.nwords, C.byref(out), self.device, self.stream))
        return out.value

    def nonzero(self) -> int:
        out = C.c_uint64(0)
        N.check(N.lib().psk_table_nonzero_u32(self.ptr, self.nwords, C.byref(out), self.device, self.stream))
        return out.value

    # -- output buffers matching where the batch lives
    def out_buffer(self, b: KeyBatch, n_items: int, np_dtype, torch_dtype):
        """-> (address, finalize() -> result)"""
        if b.where == N.DEVICE:
            t = torch.empty(n_items, dtype=torch_dtype, device=f"cuda:{self.device}")
            return t.data_ptr(), (lambda: t)
        a = np.empty(n_items, dtype=np_dtype)
        return (a.ctypes.data if a.size else 0), (lambda: a)


def weights_arg(w, n: int, np_dtype, where: int, keep: list, lo: int, hi: int, device: int | None = None):
    """per-key weights -> (address or None, host_sum or None).  Scalars are broadcast on the host.
    ``device``: the sketch's HIP device -- host weights that accompany a device batch are uploaded THERE (not to
    torch's current device) and device weights must already live there."""
    if w is None:
        return None, n
    if torch is not None and isinstance(w, torch.Tensor):
        if w.is_cuda:
            if where != N.DEVICE:
                raise ValueError("device weights need a device key batch")
            if device is not None and w.device.index != device:
                raise ValueError(f"weights live on cuda:{w.device.index}, the sketch on cuda:{device}")
            want = torch.int32 if np_dtype in (np.int32, np.uint32) else torch.int64
            if w.numel() != n:
                raise ValueError("weights length differs from the number of keys")
            if w.is_floating_point() or w.dtype == torch.bool:
                raise TypeError("num_els must be integers")
            if want == torch.int32 and w.element_size() > 4 and n:
                # a 64-bit device tensor would be narrowed silently: check its range like the host path does (one small reduction)
                mn, mx = int(w.min().item()), int(w.max().item())
                if mn < lo or mx > hi:
                    raise OverflowError(f"num_els outside [{lo}, {hi}]")
            t = w.to(want).contiguous()
            keep.append(t)
            return t.data_ptr(), None
        w = w.numpy()
    if np.isscalar(w) or isinstance(w, int):
        if int(w) == 1:
            return None, n
        if not lo <= int(w) <= hi:
            raise OverflowError(f"num_els {w} outside [{lo}, {hi}]")
        a = np.full(n, int(w), dtype=np_dtype)
    else:
        src = np.asarray(w)
        if src.size != n:
            raise ValueError("weights length differs from the number of keys")
        if src.size and (int(src.min()) < lo or int(src.max()) > hi):
            raise OverflowError(f"num_els outside [{lo}, {hi}]")
        a = np.ascontiguousarray(src, dtype=np_dtype)
    total = int(a.astype(np.int64).sum()) if a.size else 0
    if where == N.DEVICE:
        target = f"cuda:{device}" if device is not None else "cuda"
        t = torch.from_numpy(a.view(np.int32 if a.dtype.itemsize == 4 else np.int64)).to(target)
        keep.append(t)
        return t.data_ptr(), total
    keep.append(a)
    return (a.ctypes.data if a.size else None), total
