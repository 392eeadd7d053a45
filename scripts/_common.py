"""helpers shared by the side-evidence scripts: device-generated streams, event-timed loops, the knobs build"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def use_knobs_build():
    """point the engine loader at csrc/libpsk_hip_knobs.so (-DPSK_BENCH_KNOBS=1: the ablation / phase-profile bits of
    PartGeom::dbg exist only there); call BEFORE importing pyprobables_amd.  Builds it when missing (hipcc, ~2 min)."""
    lib = ROOT / "pyprobables_amd" / "csrc" / "libpsk_hip_knobs.so"
    if not lib.exists():
        from pyprobables_amd import build

        build.build(knobs=True)
    os.environ["PSK_LIB_PATH"] = str(lib)


def gen_keys(n, start=0, device=0):
    import torch

    import bench
    from pyprobables_amd import _native as N

    t = torch.empty((n, 16), dtype=torch.uint8, device=f"cuda:{device}")
    N.check(N.lib().psk_gen_keys16(t.data_ptr(), start, n, bench.SEED, device, torch.cuda.current_stream(device).cuda_stream or None))
    return t


def gen_weights(n, start=0, device=0):
    import torch

    import bench
    from pyprobables_amd import _native as N

    t = torch.empty(n, dtype=torch.int32, device=f"cuda:{device}")
    N.check(N.lib().psk_gen_weights(t.data_ptr(), start, n, bench.SEED, device, torch.cuda.current_stream(device).cuda_stream or None))
    return t


def timed_loop(fn, iters, warm=2):
    import torch

    import bench

    return bench.timed_loop(torch, fn, iters, warm)
