"""Build the HIP engine in-tree: ``python -m pyprobables_amd.build`` -> ``csrc/libpsk_hip.so``.

hipcc cross-compiles gfx950 without a GPU.  The translation units (C ABI + direct kernels, and one per
partitioned-path launcher family) are compiled in parallel, then linked with ``-no-hip-rt``: the hip* symbols stay
undefined so the library binds to the HIP runtime already loaded in the process (see ``_native.py``).
"""

from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
# launcher translation units built twice (power-of-two / Barrett instantiations, see psk_host.hpp PSK_TU_POW2)
VARIANT_SOURCES = [
    "psk_part_bloom_add.hip",
    "psk_part_bloom_check.hip",
    "psk_part_cbf.hip",
    "psk_part_cms.hip",
    "psk_part_cms_check.hip",
    "psk_part_cbf_check.hip",
    "psk_part_cbf_multi.hip",
    "psk_part_cbf_window.hip",
]
PLAIN_SOURCES = ["psk_capi.hip", "psk_index_ops.hip", "psk_merge.hip", "psk_part_dispatch.hip"]
# (source, object stem, extra flags); the heaviest units first so that the pool stays busy to the end
SOURCES = [(f, Path(f).stem + f"_v{v}", [f"-DPSK_TU_POW2={v}"]) for f in VARIANT_SOURCES for v in (1, 0)] + \
          [(f, Path(f).stem, []) for f in PLAIN_SOURCES]
HEADERS = ["psk_device.hpp", "psk_partition.hpp", "psk_host.hpp", "psk_part_counter.hpp", "psk_lookup.hpp", "psk_part_lookup.hpp", "psk_nibble.hpp", "psk_nibble_pipe.hpp", "psk_window.hpp", "psk_digest.hpp",
           "../../include/psk.h"]
OUT = CSRC / "libpsk_hip.so"
OBJ = CSRC / "build"
OUT_KNOBS = CSRC / "libpsk_hip_knobs.so"
OBJ_KNOBS = CSRC / "build_knobs"
# --offload-compress: every unit's gfx950 code object is zstd-compressed inside its bundle (the HIP runtime inflates it when the library is
# loaded): the ~1000 k_part_scatter instantiations are near-copies of each other, so the library that travels to every GPU box shrinks several-fold
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fvisibility-inlines-hidden", "--offload-compress"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found")


def _newest_header() -> float:
    return max((CSRC / h).resolve().stat().st_mtime for h in HEADERS)


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(path: Path, seen=None) -> set:
    """the files `path` includes with quotes, transitively (every translation unit is rebuilt only when one of ITS headers changed)"""
    seen = set() if seen is None else seen
    for inc in _INC.findall(path.read_text()):
        f = (path.parent / inc).resolve()
        if f.exists() and f not in seen:
            seen.add(f)
            _deps(f, seen)
    return seen


def needs_build() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    if _newest_header() > t or any((CSRC / f).stat().st_mtime > t for f, _, _ in SOURCES):
        return True
    # (a PSK_BUILD_ONLY build links a library that is newer than every header while the objects it skipped are stale; a tree that
    # received only the built library -- objects are not tracked -- has nothing to be stale: the library's own time stamp decided above)
    if not OBJ.exists():
        return False
    for src, stem, _ in SOURCES:
        obj = OBJ / (stem + ".o")
        if not obj.exists() or obj.stat().st_mtime < max([(CSRC / src).stat().st_mtime] + [f.stat().st_mtime for f in _deps(CSRC / src)]):
            return True
    return False


def _compile(item, force: bool, verbose: bool, objdir: Path, extra: list) -> Path:
    src, stem, flags = item
    extra = [*extra, *flags]
    obj = objdir / (stem + ".o")
    dep = max([(CSRC / src).stat().st_mtime] + [f.stat().st_mtime for f in _deps(CSRC / src)])
    # PSK_BUILD_ONLY="bloom_check,capi" (development only): recompile just the matching units, keep the other objects as they are -- for
    # experiments inside one kernel family; never for a change of anything two units share (psk_sketch, PartGeom, the launchers' signatures)
    only = [t for t in os.environ.get("PSK_BUILD_ONLY", "").split(",") if t]
    if only and obj.exists() and not any(t in stem for t in only):
        return obj
    if force or not obj.exists() or obj.stat().st_mtime < dep:
        cmd = [hipcc(), *FLAGS, *extra, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=str(CSRC))
    return obj


PYLIST_SRC = CSRC / "psk_pylist.c"
PYLIST_OUT = CSRC.parent / "_pylist.so"


def build_pylist(force: bool = False, verbose: bool = True) -> Path:
    """the host-side packer of key LISTS (csrc/psk_pylist.c, a CPython extension: gcc, no GPU code) -> pyprobables_amd/_pylist.so"""
    import sysconfig

    if not force and PYLIST_OUT.exists() and PYLIST_OUT.stat().st_mtime >= PYLIST_SRC.stat().st_mtime:
        return PYLIST_OUT
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        raise RuntimeError("gcc not found (needed for pyprobables_amd/_pylist.so)")
    cmd = [cc, "-O2", "-fPIC", "-shared", "-Wall", "-I" + sysconfig.get_paths()["include"], str(PYLIST_SRC), "-o", str(PYLIST_OUT)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return PYLIST_OUT


def build(force: bool = False, verbose: bool = True, knobs: bool = False) -> Path:
    """knobs=True: the bench-only build with the ablation / phase-profile bits compiled in (-DPSK_BENCH_KNOBS=1) ->
    csrc/libpsk_hip_knobs.so; never loaded unless PSK_LIB_PATH points at it (scripts/ablate.py, scripts/profile_sq.sh)"""
    out = OUT_KNOBS if knobs else OUT
    if not knobs:
        try:
            build_pylist(force, verbose)
        except Exception as e:  # noqa: BLE001 -- keys.py treats _pylist as optional (a Python loop packs lists without it): never fail the engine build for it
            print(f"warning: pyprobables_amd/_pylist.so not built ({e}); key lists are packed by the Python fallback", file=sys.stderr, flush=True)
    if not knobs and not force and not needs_build():
        return out
    objdir = OBJ_KNOBS if knobs else OBJ
    extra = ["-DPSK_BENCH_KNOBS=1"] if knobs else []
    objdir.mkdir(exist_ok=True)
    jobs = min(len(SOURCES), os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(lambda s: _compile(s, force, verbose, objdir, extra), SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt", "-o", str(out), *map(str, objs)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, knobs="--knobs" in sys.argv))
