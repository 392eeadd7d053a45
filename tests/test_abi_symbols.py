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
