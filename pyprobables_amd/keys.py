"""Packing of key batches into the four layouts of the C ABI (include/psk.h ``psk_layout``).

The reference hashes one python object at a time (``fnv_1a`` walks ``list(key)`` for bytes-likes and
``map(ord, key)`` for str, hashes.py:98).  A batch is turned into ONE contiguous buffer:

* equal-length byte keys            -> PSK_KEYS_FIXED    uint8[n][L]
* ragged byte / latin-1 str keys    -> PSK_KEYS_VARLEN8  uint8 blob + uint64 offsets[n+1]
* any str with a code point > 255   -> PSK_KEYS_VARLEN32 uint32 code points + offsets (never UTF-8:
  the reference XORs the whole code point into the state)
* ``(n, L)`` uint8 numpy arrays / torch tensors are taken as-is (torch CUDA tensors zero-copy).
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

from . import _native as N

try:  # torch is plumbing: device buffers + streams
    import torch
except Exception:  # pragma: no cover
    torch = None


@dataclass
class KeyBatch:
    layout: int
    data: int            # address (host or device)
    offsets: int         # address or 0
    n: int
    key_len: int
    where: int           # N.HOST / N.DEVICE
    device: int | None = None
    keep: list = field(default_factory=list)  # keeps the buffers alive for the duration of the call

    def args(self):
        return (self.layout, self.data or None, self.offsets or None, self.n, self.key_len)


try:  # the C packer of key lists (csrc/psk_pylist.c, built by build.py); without it the python packer below does the same job
    from . import _pylist
except Exception:  # pragma: no cover
    _pylist = None


def _np_ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def _is_key(obj) -> bool:
    return isinstance(obj, (str, bytes, bytearray, memoryview))


def _code_points(key) -> np.ndarray:
    if isinstance(key, str):
        return np.frombuffer(key.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)
    return np.frombuffer(bytes(key), dtype=np.uint8).astype(np.uint32)


def _pack_homogeneous(keys: list, n: int):
    """all-str or all-bytes lists without a per-key Python loop: ONE join, ONE encode, lengths through map(len).
    (The per-key loop packs ~2.5 M keys/s; this is what bounds ``add_many(list_of_str)``.)  None: mixed types."""
    try:
        joined = "".join(keys)
        wide_ok = True
    except TypeError:
        wide_ok = False
        try:
            joined = b"".join(keys)
        except TypeError:
            return None
        if not set(map(type, keys)) <= {bytes, bytearray, memoryview}:  # b"".join takes ANY buffer (numpy arrays, array('B')):
            return None                                                 # those are not keys -- the per-key path raises
    lens = np.fromiter(map(len, keys), dtype=np.int64, count=n)
    if wide_ok:
        try:
            blob8 = joined.encode("latin-1")  # code points <= 255 == byte values
        except UnicodeEncodeError:
            offs = np.zeros(n + 1, dtype=np.int64)  # (same dtype as `lens`: a uint64 `out` sends cumsum down its casting path, 10 x slower)
            np.cumsum(lens, out=offs[1:])
            offs = offs.view(np.uint64)
            blob = np.frombuffer(joined.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)
            if blob.size != int(offs[-1]):
                return None  # (lone surrogates / odd encodings: let the careful path decide)
            blob = np.ascontiguousarray(blob) if blob.size else np.zeros(1, dtype=np.uint32)
            return KeyBatch(N.KEYS_VARLEN32, _np_ptr(blob), _np_ptr(offs), n, 0, N.HOST, None, [blob, offs])
    else:
        blob8 = joined
    if int(lens.sum()) != len(blob8):
        return None
    first = int(lens[0])
    if int(lens.min()) == first == int(lens.max()):
        a = np.frombuffer(blob8, dtype=np.uint8).reshape(n, first) if first else np.zeros((n, 0), dtype=np.uint8)
        return KeyBatch(N.KEYS_FIXED, _np_ptr(a) if a.size else 0, 0, n, first, N.HOST, None, [a, blob8])
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    offs = offs.view(np.uint64)
    blob = np.frombuffer(blob8, dtype=np.uint8)
    return KeyBatch(N.KEYS_VARLEN8, _np_ptr(blob), _np_ptr(offs), n, 0, N.HOST, None, [blob, offs, blob8])


# device (blob, offsets) pairs are validated only on request (see _pack_ragged)
VALIDATE_DEVICE_OFFSETS = os.environ.get("PSK_VALIDATE_OFFSETS", "0") not in ("", "0")


def _is_array(x) -> bool:
    return isinstance(x, np.ndarray) or (torch is not None and isinstance(x, torch.Tensor))


def _pack_ragged(blob, offsets) -> KeyBatch:
    """a ragged batch handed over as it lies in memory: ``blob`` = the keys' elements end to end (uint8 bytes, or 4-byte code points
    for str keys -- hashes.py:98 XORs whole code points), ``offsets`` = n + 1 ascending 8-byte positions, key i = blob[offsets[i]:offsets[i+1]].
    Both on the host, or both on one device (zero-copy: the C ABI takes ``offsets`` from either side).

    CONTRACT for device pairs: the offsets ascend and stay inside the blob.  Host pairs are validated here; a device pair is handed to the
    kernels as it is (checking it would cost a reduction over the offsets and a synchronisation on every batch) -- a descending or
    out-of-range offset makes a lane walk memory far outside the blob, which can fault the GPU.  ``PSK_VALIDATE_OFFSETS=1`` in the
    environment (or ``keys.VALIDATE_DEVICE_OFFSETS = True``) turns the same two checks on for device pairs (debugging a producer)."""
    is_t = [torch is not None and isinstance(x, torch.Tensor) for x in (blob, offsets)]
    cuda = [t and x.is_cuda for t, x in zip(is_t, (blob, offsets))]
    if cuda[0] != cuda[1]:
        raise TypeError("(blob, offsets): both on the host or both on one device")
    if cuda[0]:
        if blob.device != offsets.device:
            raise TypeError("(blob, offsets): both tensors must live on the same device")
        if blob.dim() != 1 or offsets.dim() != 1 or offsets.numel() < 1:
            raise TypeError("(blob, offsets): 1-D blob and 1-D offsets of n + 1 entries")
        if offsets.element_size() != 8 or offsets.dtype.is_floating_point:
            raise TypeError("(blob, offsets): offsets must be 64-bit integers")
        if blob.dtype == torch.uint8:
            layout = N.KEYS_VARLEN8
        elif blob.element_size() == 4 and not blob.dtype.is_floating_point:
            layout = N.KEYS_VARLEN32
        else:
            raise TypeError("(blob, offsets): blob must be uint8 bytes or 4-byte code points")
        b, o = blob.contiguous(), offsets.contiguous()
        if VALIDATE_DEVICE_OFFSETS:
            oi = o.view(torch.int64)
            if oi.numel() > 1 and bool((oi[1:] < oi[:-1]).any().item()):
                raise ValueError("(blob, offsets): offsets must ascend")
            if int(oi[0].item()) < 0 or int(oi[-1].item()) > b.numel():
                raise ValueError("(blob, offsets): offsets reach past the blob")
        if b.numel() == 0:
            b = torch.zeros(1, dtype=blob.dtype, device=blob.device)  # (all keys empty: the engine still wants an address)
        return KeyBatch(layout, b.data_ptr(), o.data_ptr(), o.numel() - 1, 0, N.DEVICE, b.device.index, [b, o])
    b = blob.numpy() if is_t[0] else np.asarray(blob)
    o = offsets.numpy() if is_t[1] else np.asarray(offsets)
    if b.ndim != 1 or o.ndim != 1 or o.size < 1:
        raise TypeError("(blob, offsets): 1-D blob and 1-D offsets of n + 1 entries")
    if o.dtype.kind not in "iu" or o.dtype.itemsize != 8:
        raise TypeError("(blob, offsets): offsets must be 64-bit integers")
    if b.dtype == np.uint8:
        layout = N.KEYS_VARLEN8
    elif b.dtype.kind in "iu" and b.dtype.itemsize == 4:
        layout = N.KEYS_VARLEN32
    else:
        raise TypeError("(blob, offsets): blob must be uint8 bytes or 4-byte code points")
    o = np.ascontiguousarray(o).view(np.uint64)
    # (offsets need not start at 0: a window into a larger blob is fine)
    if o.size > 1 and (np.diff(o.view(np.int64)) < 0).any():
        raise ValueError("(blob, offsets): offsets must ascend")
    if int(o[-1]) > b.size:
        raise ValueError("(blob, offsets): offsets reach past the blob")
    if o[0] != 0:  # (the engine stages host blobs from their first byte: offsets relative to the first key)
        b = b[int(o[0]):]
        o = o - o[0]
    b = np.ascontiguousarray(b)
    if b.size == 0:
        b = np.zeros(1, dtype=b.dtype)
    return KeyBatch(layout, _np_ptr(b), _np_ptr(o), o.size - 1, 0, N.HOST, None, [b, o])


def one_key_bytes(key):
    """ONE ``str`` / ``bytes`` key as the bytes the engine hashes -- a ``str`` by code point, so code points <= 255 travel as byte values
    (hashes.py:98 of the reference: ``ord`` per character) -- or None when the key needs the general packer (wider code points,
    ``bytearray`` / ``memoryview``, anything else: ``pack_keys`` also raises the TypeError)."""
    if type(key) is str:
        try:
            return key.encode("latin-1")
        except UnicodeEncodeError:
            return None
    return key if type(key) is bytes else None


def pack_keys(keys) -> KeyBatch:
    """one key, a sequence of keys, a (n, L) uint8 array or a (n, L) uint8 torch tensor, or a ragged ``(blob, offsets)`` pair of
    arrays / tensors (host or device) -> KeyBatch"""
    if _is_key(keys):
        keys = [keys]
    if isinstance(keys, tuple) and len(keys) == 2 and _is_array(keys[0]) and _is_array(keys[1]):
        return _pack_ragged(keys[0], keys[1])
    if torch is not None and isinstance(keys, torch.Tensor):
        if keys.dtype != torch.uint8 or keys.dim() != 2:
            raise TypeError("tensor key batches must be uint8 of shape (n, key_len)")
        if keys.is_cuda:
            t = keys.contiguous()
            return KeyBatch(N.KEYS_FIXED, t.data_ptr(), 0, t.shape[0], t.shape[1], N.DEVICE, t.device.index, [t])
        keys = keys.numpy()
    if isinstance(keys, np.ndarray):
        a = keys
        if a.dtype.kind == "S":  # fixed-width byte strings: raw buffer, trailing NULs included
            a = np.ascontiguousarray(a).view(np.uint8).reshape(a.shape[0], a.dtype.itemsize)
        if a.dtype != np.uint8 or a.ndim != 2:
            raise TypeError("array key batches must be uint8 of shape (n, key_len) or dtype 'S<L>'")
        a = np.ascontiguousarray(a)
        return KeyBatch(N.KEYS_FIXED, _np_ptr(a) if a.size else 0, 0, a.shape[0], a.shape[1], N.HOST, None, [a])

    keys = list(keys)
    n = len(keys)
    if n == 0:
        return KeyBatch(N.KEYS_FIXED, 0, 0, 0, 0, N.HOST)
    if _pylist is not None:
        packed = _pylist.pack(keys)  # None: an element that is neither str nor bytes / bytearray -- the python path decides
        if packed is not None:
            layout, blob, offs, n, key_len = packed
            if layout == 0:
                a = np.frombuffer(blob, dtype=np.uint8)[: n * key_len].reshape(n, key_len)
                return KeyBatch(N.KEYS_FIXED, _np_ptr(a) if a.size else 0, 0, n, key_len, N.HOST, None, [a, blob])
            b = np.frombuffer(blob, dtype=np.uint8 if layout == 1 else np.uint32)
            o = np.frombuffer(offs, dtype=np.uint64)
            return KeyBatch(N.KEYS_VARLEN8 if layout == 1 else N.KEYS_VARLEN32, _np_ptr(b), _np_ptr(o), n, 0, N.HOST, None, [b, o, blob, offs])
    fast = _pack_homogeneous(keys, n)
    if fast is not None:
        return fast
    raw = []
    wide = False
    for k in keys:
        if isinstance(k, str):
            try:
                raw.append(k.encode("latin-1"))  # code points <= 255 == byte values
            except UnicodeEncodeError:
                wide = True
                break
        elif isinstance(k, (bytes, bytearray, memoryview)):
            raw.append(bytes(k))
        else:
            raise TypeError(f"keys must be str or bytes-like, got {type(k).__name__}")
    if wide:
        parts = [_code_points(k) for k in keys]
        offs = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum([p.size for p in parts], out=offs[1:])
        blob = np.concatenate(parts) if offs[-1] else np.zeros(1, dtype=np.uint32)
        blob = np.ascontiguousarray(blob, dtype=np.uint32)
        return KeyBatch(N.KEYS_VARLEN32, _np_ptr(blob), _np_ptr(offs), n, 0, N.HOST, None, [blob, offs])
    first = len(raw[0])
    if all(len(r) == first for r in raw):
        a = np.frombuffer(b"".join(raw), dtype=np.uint8).reshape(n, first) if first else np.zeros((n, 0), dtype=np.uint8)
        return KeyBatch(N.KEYS_FIXED, _np_ptr(a) if a.size else 0, 0, n, first, N.HOST, None, [a])
    offs = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in raw], out=offs[1:])
    blob = np.frombuffer(b"".join(raw), dtype=np.uint8)
    return KeyBatch(N.KEYS_VARLEN8, _np_ptr(blob), _np_ptr(offs), n, 0, N.HOST, None, [blob, offs])


def pack_hashes(hashes, need: int) -> KeyBatch:
    """pre-computed hashes -> PSK_KEYS_HASHES.  Accepts one list of ints (a single key, add_alt style),
    a sequence of such lists, a (n, h) uint64 array or a (n, h) torch tensor (int64/uint64 bits)."""
    if torch is not None and isinstance(hashes, torch.Tensor):
        if hashes.dim() != 2 or hashes.element_size() != 8:
            raise TypeError("hash tensors must be 2-D with 8-byte elements")
        if hashes.is_cuda:
            t = hashes.contiguous()
            if t.shape[1] < need:
                raise ValueError(f"need at least {need} hashes per key, got {t.shape[1]}")
            return KeyBatch(N.KEYS_HASHES, t.data_ptr(), 0, t.shape[0], t.shape[1], N.DEVICE, t.device.index, [t])
        hashes = hashes.numpy().view(np.uint64)
    if not isinstance(hashes, np.ndarray):
        hashes = list(hashes)
        if hashes and not isinstance(hashes[0], (list, tuple, np.ndarray)):
            hashes = [hashes]
        if len({len(h) for h in hashes}) > 1:
            raise ValueError("all keys of a pre-hashed batch must carry the same number of hashes")
        hashes = np.array([[int(x) & 0xFFFFFFFFFFFFFFFF for x in h] for h in hashes], dtype=np.uint64).reshape(len(hashes), -1)
    a = np.ascontiguousarray(hashes, dtype=np.uint64)
    if a.ndim != 2:
        raise TypeError("hash batches must be 2-D (n, hashes_per_key)")
    if a.shape[0] and a.shape[1] < need:
        raise ValueError(f"need at least {need} hashes per key, got {a.shape[1]}")
    return KeyBatch(N.KEYS_HASHES, _np_ptr(a) if a.size else 0, 0, a.shape[0], a.shape[1], N.HOST, None, [a])


# ---------------------------------------------------------------- digest families on the device
def _pack_bytes_utf8(keys) -> KeyBatch:
    """byte image of every key for the digest families: a str is UTF-8 encoded (hashes.py:34 -- unlike the FNV
    family, which walks code points)"""
    if _is_key(keys):
        keys = [keys]
    if (torch is not None and isinstance(keys, torch.Tensor)) or isinstance(keys, np.ndarray):
        return pack_keys(keys)  # raw (n, L) byte matrices
    if isinstance(keys, tuple) and len(keys) == 2 and _is_array(keys[0]) and _is_array(keys[1]):
        return pack_keys(keys)  # ragged (blob, offsets): bytes as they lie
    keys = list(keys)
    try:
        if "".join(keys).isascii():  # ASCII: UTF-8 bytes == code points, the vectorised packing applies as is
            return pack_keys(keys)
    except TypeError:
        pass
    raw = []
    for k in keys:
        if isinstance(k, str):
            raw.append(k.encode("utf-8"))
        elif isinstance(k, (bytes, bytearray, memoryview)):
            raw.append(bytes(k))
        else:
            raise TypeError(f"keys must be str or bytes-like, got {type(k).__name__}")
    return pack_keys(raw)


def digest_batch(keys, algo: int, depth: int, device: int, stream=None) -> KeyBatch:
    """keys -> PSK_KEYS_HASHES batch of the chained md5 / sha256 family (hashes.py:125-150), computed by the engine
    (``psk_digest_chain``).  Device key tensors stay on the device; host keys come back as a host hash matrix."""
    b = _pack_bytes_utf8(keys)
    if b.layout not in (N.KEYS_FIXED, N.KEYS_VARLEN8):
        raise TypeError("digest families hash bytes")
    if b.where == N.DEVICE:
        out = torch.empty((b.n, depth), dtype=torch.int64, device=f"cuda:{device}")
        N.check(N.lib().psk_digest_chain(algo, *b.args(), depth, N.DEVICE, out.data_ptr(), device, stream))
        return KeyBatch(N.KEYS_HASHES, out.data_ptr() if b.n else 0, 0, b.n, depth, N.DEVICE, device, [out, *b.keep])
    out = np.empty((b.n, depth), dtype=np.uint64)
    N.check(N.lib().psk_digest_chain(algo, *b.args(), depth, N.HOST, out.ctypes.data if out.size else None, device, stream))
    return KeyBatch(N.KEYS_HASHES, _np_ptr(out) if out.size else 0, 0, b.n, depth, N.HOST, None, [out])
