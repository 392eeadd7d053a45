"""The C-ABI library loads on a CPU-only box and exports every symbol include/psk.h declares
(no compute calls here: those need a GPU and live in test_gpu_parity.py)."""

import ctypes as C
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "psk.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psk_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from pyprobables_amd import _native as N
    from pyprobables_amd import build as B

    B.build()  # no-op when up to date; hipcc cross-compiles gfx950 without a GPU
    L = N.lib()
    names = _declared()
    assert len(names) >= 40
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/psk.h but not exported"
    # the ctypes prototype table covers exactly the declared surface
    assert sorted(N.PROTOTYPES) == names
    assert L.psk_version() >= 100


def test_library_has_no_second_hip_runtime_dependency():
    """built with -no-hip-rt: it must bind to the HIP runtime already in the process (torch's)"""
    import subprocess

    from pyprobables_amd import _native as N

    out = subprocess.run(["readelf", "-d", str(N.LIB_PATH)], capture_output=True, text=True).stdout
    assert "libamdhip64" not in out


def test_pure_size_helpers_without_gpu():
    from pyprobables_amd import _native as N

    L = N.lib()
    assert L.psk_bloom_table_bytes(63) == 16          # ceil(63/8)=8 -> padded to 16
    assert L.psk_bloom_table_bytes(2**28) == 2**25
    assert L.psk_cbf_table_bytes(63) == 256           # 252 -> 256
    assert L.psk_cms_table_bytes(1000, 5) == 20000
    assert L.psk_cms_table_bytes(2**20, 5) == 5 * 2**22


def test_fails_loudly_without_a_device():
    """no silent CPU fallback: creating a sketch without a GPU is an error"""
    import pytest

    import pyprobables_amd as pa
    from pyprobables_amd import _native as N

    if N.device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = N.lib().psk_bloom_create(1000, 4, 0, None, C.byref(h))
    assert rc == N.PSK_ENODEV and "no HIP device" in N.last_error()
    with pytest.raises(pa.NativeLibraryError):
        pa.BloomFilter(est_elements=10, false_positive_rate=0.05)
    with pytest.raises(pa.NativeLibraryError):
        pa.CountMinSketch(width=100, depth=3)
    with pytest.raises(pa.NativeLibraryError):
        pa.CountingBloomFilter(est_elements=10, false_positive_rate=0.05)


# include/psk.h's option table: the names a caller may rely on
SUPPORTED_OPTIONS = ["partition", "partition_min_keys", "partition_max_keys", "partition_cache_bytes", "scratch_budget_bytes", "pass1_bins", "bloom_lookup",
                     "cms_small_weights", "cbf_lookup_shadow", "remove_exact", "update_window", "update_window_keys", "auto_combine", "combine_keys",
                     "merge_single_rank"]
# A/B switches of experiments that were measured and dropped: bench build (-DPSK_BENCH_KNOBS=1) only
RETIRED_OPTIONS = ["update_window_shadow", "update_window_image", "update_window_nt", "nibble_lookup_pipe", "nibble_update_pipe", "nibble_update_parts",
                   "nibble_update_layout", "nibble_nt_loads", "big_table_nt", "lookup_split", "lookup_run_lanes", "lookup_collect_threads", "slice_bias",
                   "scatter_workgroups", "combine_scatter", "combine_fused_flush", "part_debug"]


def test_option_table_supported_names_work_and_retired_ones_are_rejected():
    """psk_set_option / psk_get_option are host-only: the shipped library knows every name of include/psk.h's table, gives each back what was
    set, and answers "unknown option" to the retired bench knobs"""
    from pyprobables_amd import _native as N

    L = N.lib()
    header = (ROOT / "include" / "psk.h").read_text()
    assert len(SUPPORTED_OPTIONS) <= 15
    for name in SUPPORTED_OPTIONS:
        assert f'"{name}"' in header, f"{name} missing from include/psk.h's option table"
        old = N.get_option(name)
        N.set_option(name, old)
        assert N.get_option(name) == old
    if N.LIB_PATH.name != "libpsk_hip.so":
        return  # (a bench build loaded through PSK_LIB_PATH knows the retired names by design)
    for name in RETIRED_OPTIONS:
        assert L.psk_set_option(name.encode(), 0) != 0 and "unknown option" in N.last_error(), name
        v = C.c_int64(0)
        assert L.psk_get_option(name.encode(), C.byref(v)) != 0, name
    assert L.psk_set_option(b"update_window_folds", 1) != 0 and "read-only" in N.last_error()  # counters are read-only
