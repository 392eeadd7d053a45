"""ctypes binding of the C ABI declared in ``include/psk.h`` (``csrc/libpsk_hip.so``).

The library is built without a DT_NEEDED on the HIP runtime (``-no-hip-rt``) so that it binds to the
one HIP runtime already living in the process: PyTorch-ROCm bundles its own ``libamdhip64.so`` and two
runtimes in one process cannot share streams or allocations.  We therefore load torch's copy with
RTLD_GLOBAL first (or the system one when torch is absent) and only then the engine.

No fallback: if the engine cannot be loaded every entry point raises :class:`NativeLibraryError`.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from .exceptions import NativeLibraryError

PSK_OK, PSK_EINVAL, PSK_ENODEV, PSK_ENOMEM, PSK_EHIP, PSK_ECONTRACT = 0, -1, -2, -3, -4, -5
HOST, DEVICE, DEVICE_BORROWED = 0, 1, 2
KEYS_FIXED, KEYS_VARLEN8, KEYS_VARLEN32, KEYS_HASHES = 0, 1, 2, 3
Q_MIN, Q_MEAN, Q_MEANMIN = 0, 1, 2
OP_ADD, OP_REMOVE, OP_SIGNED = 0, 1, 2
CTR_ADDED, CTR_REMOVED, CTR_VIOLATIONS, CTR_SATURATED, CTR_ABS_BOUND, CTR_ELS_OUT, CTR_COUNT = 0, 1, 2, 3, 4, 5, 8

# PSK_LIB_PATH lets a bench A/B two builds of the engine inside one process launch each (same GPU box)
LIB_PATH = Path(os.environ.get("PSK_LIB_PATH") or (Path(__file__).resolve().parent / "csrc" / "libpsk_hip.so"))

_vp, _u64, _u32, _i64, _int = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int64, C.c_int
_KEYS = [_int, _vp, _vp, _u64, _u32]  # layout, data, offsets, n, key_len

# name -> (restype, argtypes): every symbol include/psk.h declares
PROTOTYPES = {
    "psk_last_error": (C.c_char_p, []),
    "psk_version": (_int, []),
    "psk_device_count": (_int, [C.POINTER(_int)]),
    "psk_set_option": (_int, [C.c_char_p, _i64]),
    "psk_get_option": (_int, [C.c_char_p, C.POINTER(_i64)]),
    "psk_debug_phase_profile": (_int, [_vp, _u32, _u32, C.POINTER(_u64)]),
    "psk_bloom_table_bytes": (_u64, [_u64]),
    "psk_cbf_table_bytes": (_u64, [_u64]),
    "psk_cms_table_bytes": (_u64, [_u64, _u32]),
    "psk_bloom_create": (_int, [_u64, _u32, _int, _vp, C.POINTER(_vp)]),
    "psk_cbf_create": (_int, [_u64, _u32, _int, _vp, C.POINTER(_vp)]),
    "psk_cms_create": (_int, [_u64, _u32, _int, _vp, C.POINTER(_vp)]),
    "psk_destroy": (_int, [_vp]),
    "psk_clear": (_int, [_vp, _vp]),
    "psk_synchronize": (_int, [_vp, _vp]),
    "psk_table_info": (_int, [_vp, C.POINTER(_vp), C.POINTER(_u64), C.POINTER(_u64)]),
    "psk_read_table": (_int, [_vp, _vp, _u64, _vp]),
    "psk_write_table": (_int, [_vp, _vp, _u64, _vp]),
    "psk_get_counters": (_int, [_vp, C.POINTER(_i64), _vp]),
    "psk_reset_counters": (_int, [_vp, _vp]),
    "psk_rescan_bound": (_int, [_vp, _vp]),
    "psk_sketch_set_option": (_int, [_vp, C.c_char_p, _i64]),
    "psk_sketch_get_option": (_int, [_vp, C.c_char_p, C.POINTER(_i64)]),
    "psk_bloom_add": (_int, [_vp, *_KEYS, _int, _vp]),
    "psk_bloom_check": (_int, [_vp, *_KEYS, _int, _vp, _vp]),
    "psk_bloom_check_begin": (_int, [_vp, *_KEYS, _vp]),
    "psk_bloom_check_finish": (_int, [_vp, _vp, _vp]),
    "psk_bloom_check_bits": (_int, [_vp, *_KEYS, _int, _vp, _vp, _vp]),
    "psk_cbf_add": (_int, [_vp, *_KEYS, _vp, _int, _vp]),
    "psk_cbf_remove": (_int, [_vp, *_KEYS, _vp, _int, _vp]),
    "psk_cbf_check": (_int, [_vp, *_KEYS, _int, _vp, _vp]),
    "psk_cbf_update_combined": (_int, [_vp, *_KEYS, _vp, _int, _int, _vp]),
    "psk_flush": (_int, [_vp, _vp]),
    "psk_cbf_update_ordered": (_int, [_vp, *_KEYS, _vp, _int, _int, _vp, _vp]),
    "psk_cms_add": (_int, [_vp, *_KEYS, _vp, _int, _vp]),
    "psk_cms_remove": (_int, [_vp, *_KEYS, _vp, _int, _vp]),
    "psk_cms_check": (_int, [_vp, *_KEYS, _int, _int, _vp, _vp]),
    "psk_cms_check_meanmin": (_int, [_vp, *_KEYS, _int, _i64, _vp, _vp]),
    "psk_cms_update_ordered": (_int, [_vp, *_KEYS, _vp, _int, _int, _i64, _int, _vp, _vp]),
    "psk_fnv1a_hash": (_int, [*_KEYS, _u32, _int, _vp, _int, _vp]),
    "psk_digest_chain": (_int, [_int, *_KEYS, _u32, _int, _vp, _int, _vp]),
    "psk_table_or": (_int, [_vp, _vp, _u64, _int, _vp]),
    "psk_table_and": (_int, [_vp, _vp, _u64, _int, _vp]),
    "psk_table_popcount": (_int, [_vp, _u64, C.POINTER(_u64), _int, _vp]),
    "psk_table_nonzero_u32": (_int, [_vp, _u64, C.POINTER(_u64), _int, _vp]),
    "psk_table_add_sat_i32": (_int, [_vp, _vp, _u64, _int, _vp]),
    "psk_table_add_u32": (_int, [_vp, _vp, _u64, C.POINTER(_u64), _int, _vp]),
    "psk_cbf_intersect": (_int, [_vp, _vp, _vp, _u64, C.POINTER(_u64), _int, _vp]),
    "psk_cbf_jaccard_counts": (_int, [_vp, _vp, _u64, C.POINTER(_u64), _int, _vp]),
    "psk_release_scratch": (_int, [_vp]),
    "psk_scratch_bytes": (_int, [_vp, C.POINTER(C.c_uint64)]),
    "psk_or_reduce_slices": (_int, [_vp, _vp, _u32, _u64, _int, _vp]),
    "psk_merge_or": (_int, [_vp, _vp, _vp]),
    "psk_merge_sum": (_int, [_vp, _vp, _vp]),
    "psk_bloom_indices": (_int, [_vp, _int, _vp, _vp, _u64, _u32, _int, _vp, _vp]),
    "psk_idx_test": (_int, [_vp, _vp, _u64, _u32, _vp, _int, _int, _vp]),
    "psk_idx_resolve_ordered": (_int, [_vp, _vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp, _int, _vp]),
    "psk_idx_resolve_ordered_hashed": (_int, [_vp, _vp, _vp, _u64, _u32, _vp, _u32, _vp, _vp, _int, _vp]),
    "psk_idx_insert": (_int, [_vp, _vp, _vp, _u64, _u32, _int, _vp]),
    "psk_bytes_or": (_int, [_vp, _vp, _u64, _int, _vp]),
    "psk_gen_keys16": (_int, [_vp, _u64, _u64, _u64, _int, _vp]),
    "psk_gen_weights": (_int, [_vp, _u64, _u64, _u64, _int, _vp]),
    "psk_gups": (_int, [_vp, _u64, _u64, _int, _u64, _vp, _int, _vp]),
}

_lib = None
_hip_rt = None


def _preload_hip_runtime():
    """make ONE HIP runtime globally visible before the engine resolves its hip* symbols"""
    global _hip_rt
    if _hip_rt is not None:
        return
    candidates = []
    try:
        import torch  # noqa: PLC0415  (plumbing: device memory, streams, torch.distributed)

        candidates.append(Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so")
    except Exception:  # torch absent: a plain C/C++ style deployment
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    candidates += [Path(rocm) / "lib" / "libamdhip64.so", Path("libamdhip64.so")]
    errors = []
    for cand in candidates:
        try:
            _hip_rt = C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
            return
        except OSError as ex:
            errors.append(f"{cand}: {ex}")
    raise NativeLibraryError("could not load a HIP runtime (libamdhip64.so): " + "; ".join(errors))


def lib():
    """the loaded engine (raises NativeLibraryError when it is missing -- there is no fallback)"""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NativeLibraryError(
                f"{LIB_PATH} is missing: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()' "
                "or pyprobables_amd/build.py); there is no CPU fallback"
            )
        _preload_hip_runtime()
        try:
            L = C.CDLL(str(LIB_PATH))
        except OSError as ex:
            raise NativeLibraryError(f"failed to load {LIB_PATH}: {ex}") from ex
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as ex:
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from ex
            fn.restype, fn.argtypes = res, args
        _lib = L
        for env, opt in (("PSK_PART_DEBUG", "part_debug"),):  # bench-only ablation bits, see PartGeom::dbg
            if os.environ.get(env):
                L.psk_set_option(opt.encode(), int(os.environ[env]))
        for item in filter(None, os.environ.get("PSK_OPTIONS", "").split(",")):  # e.g. PSK_OPTIONS=partition_two_level_slices=0
            name, _, value = item.partition("=")
            if L.psk_set_option(name.strip().encode(), int(value)) != 0:
                raise NativeLibraryError(f"PSK_OPTIONS: {(L.psk_last_error() or b'').decode()}")
    return _lib


def last_error() -> str:
    return (lib().psk_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> None:
    """status code -> exception"""
    if rc == PSK_OK:
        return
    msg = last_error()
    if rc == PSK_EINVAL:
        raise ValueError(msg)
    if rc == PSK_ENOMEM:
        raise MemoryError(msg)
    raise NativeLibraryError(f"psk error {rc}: {msg}")


def set_option(name: str, value: int) -> None:
    check(lib().psk_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = _i64(0)
    check(lib().psk_get_option(name.encode(), C.byref(v)))
    return v.value


def device_count() -> int:
    n = _int(0)
    rc = lib().psk_device_count(C.byref(n))
    return n.value if rc == PSK_OK else 0
