"""The drop-in boundary is a C ABI: build examples/psk_demo.c with the system C compiler, link it against
libpsk_hip.so + the ROCm HIP runtime (no Python, no torch in the process) and run it on the GPU."""

import json
import shutil
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
# what the REAL reference held after the C programs' workloads (tests/golden/gen_golden_cdemo.py, imported pyprobables)
GOLDEN = json.loads((ROOT / "tests" / "golden" / "golden_cdemo.json").read_text())


def _printed(stdout: str) -> dict:
    return dict(x.split("=", 1) for x in stdout.splitlines() if "=" in x and " " not in x.split("=", 1)[0])


def test_plain_c_program_drives_the_engine(tmp_path):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cc = shutil.which("gcc") or shutil.which("cc")
    rocm_lib = Path("/opt/rocm/lib")
    if cc is None or not (rocm_lib / "libamdhip64.so").exists():
        pytest.skip("no C compiler / ROCm runtime")
    lib_dir = ROOT / "pyprobables_amd" / "csrc"
    exe = tmp_path / "psk_demo"
    subprocess.run([cc, "-O2", "-std=c11", str(ROOT / "examples" / "psk_demo.c"), "-I", str(ROOT / "include"), "-L", str(lib_dir),
                    "-lpsk_hip", "-L", str(rocm_lib), "-lamdhip64", f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{rocm_lib}", "-o", str(exe)],
                   check=True)
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "PSK C ABI OK" in run.stdout
    got, g = _printed(run.stdout), GOLDEN["psk_demo"]
    assert got["table_sha256"] == g["sha256_table"] == "1e9bfcfc3ad261a36f218553a905d142a95295333e2b6ec63d9710543db83d7c"
    assert got["membership_sha256"] == g["sha256_membership_bytes"] and int(got["elements_added"]) == g["elements_added"]
    assert f"false positives {g['false_positives']} of" in run.stdout


def test_plain_c_program_merges_replicas_over_rccl(tmp_path):
    """examples/psk_merge_demo.c: one thread per GPU, psk_merge_or / psk_merge_sum over RCCL communicators the C host made
    itself (ncclCommInitAll); on a one-GPU box the single-rank communicator still drives the whole collective path"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cc = shutil.which("gcc") or shutil.which("cc")
    rocm = Path("/opt/rocm")
    if cc is None or not (rocm / "lib" / "librccl.so").exists() or not (rocm / "include" / "rccl" / "rccl.h").exists():
        pytest.skip("no C compiler / RCCL development files")
    lib_dir = ROOT / "pyprobables_amd" / "csrc"
    exe = tmp_path / "psk_merge_demo"
    subprocess.run([cc, "-O2", "-std=gnu11", "-pthread", "-D__HIP_PLATFORM_AMD__", str(ROOT / "examples" / "psk_merge_demo.c"), "-I", str(ROOT / "include"),
                    "-I", str(rocm / "include"), "-L", str(lib_dir), "-lpsk_hip", "-L", str(rocm / "lib"), "-lamdhip64", "-lrccl",
                    f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{rocm / 'lib'}", "-o", str(exe)], check=True)
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "PSK MERGE OK" in run.stdout


def test_plain_c_program_two_threads_two_handles(tmp_path):
    """examples/psk_threads_demo.c: two pthreads, each with its own handle, per-sketch options and HIP stream on device 0, run a Bloom
    and a CountingBloomFilter workload at the same time; tables byte-identical to a sequential repeat, psk_last_error per thread"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cc = shutil.which("gcc") or shutil.which("cc")
    rocm = Path("/opt/rocm")
    if cc is None or not (rocm / "lib" / "libamdhip64.so").exists() or not (rocm / "include" / "hip" / "hip_runtime_api.h").exists():
        pytest.skip("no C compiler / HIP development files")
    lib_dir = ROOT / "pyprobables_amd" / "csrc"
    exe = tmp_path / "psk_threads_demo"
    subprocess.run([cc, "-O2", "-std=gnu11", "-pthread", "-D__HIP_PLATFORM_AMD__", str(ROOT / "examples" / "psk_threads_demo.c"), "-I", str(ROOT / "include"),
                    "-I", str(rocm / "include"), "-L", str(lib_dir), "-lpsk_hip", "-L", str(rocm / "lib"), "-lamdhip64",
                    f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{rocm / 'lib'}", "-o", str(exe)], check=True)
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "PSK THREADS OK" in run.stdout
    got = _printed(run.stdout)
    assert got["bloom_table_sha256"] == GOLDEN["threads_bloom"]["sha256_table"] and int(got["bloom_elements_added"]) == GOLDEN["threads_bloom"]["elements_added"]
    assert got["cbf_table_sha256"] == GOLDEN["threads_cbf"]["sha256_table"] and int(got["cbf_elements_added"]) == GOLDEN["threads_cbf"]["elements_added"]
    assert got["cbf_mins_sha256"] == GOLDEN["threads_cbf"]["sha256_last_round_mins_u32"]
