"""Build the HIP engine in-tree: ``python -m pyprobables_amd.build`` -> ``csrc/libpsk_hip.so``.

hipcc cross-compiles gfx950 without a GPU.  ``-no-hip-rt`` leaves the hip* symbols undefined so the
library binds to the HIP runtime already loaded in the process (see ``_native.py``).
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
SOURCES = ["psk_capi.hip"]
HEADERS = ["psk_device.hpp", "../../include/psk.h"]
OUT = CSRC / "libpsk_hip.so"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    return any((CSRC / f).resolve().stat().st_mtime > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return OUT
    cmd = [
        hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-no-hip-rt",
        "-Wno-unused-value", "-o", str(OUT),
    ] + [str(CSRC / s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
