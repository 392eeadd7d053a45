"""bench.py's N > 1 path on a one-GPU box: two ranks share cuda:0 over gloo (PSK_BENCH_SINGLE_DEVICE test hook), so the
shard offsets, the two real replicas, allreduce(OR) with the HIP OR-reduce kernel, the lookup split around the merge and
the driver-facing output contract (ONE JSON line, last on stdout) are exercised for real."""

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


LINE_CAP = 8000  # the driver keeps the last ~8 KB of stdout: a longer last line cannot be parsed (BENCH_r05.json: "parsed": null)


def _run(cmd, env, timeout=900, tmp=None):
    """-> (the compact line = the LAST line of stdout, the full object bench.py wrote to its detail file)"""
    detail = Path(tmp or "/tmp") / f"psk_bench_detail_{os.getpid()}.json"
    detail.unlink(missing_ok=True)
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(env, PSK_BENCH_DETAIL=str(detail)), cwd=str(ROOT))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    last = run.stdout.strip().splitlines()[-1]  # the JSON line is the last thing on stdout
    assert len(last) < LINE_CAP, len(last)
    line = json.loads(last)
    full = json.loads(detail.read_text())
    assert full["value"] == pytest.approx(line["value"], rel=1e-4) and line["detail_file"]
    return line, full


@pytest.mark.parametrize("extra", [[], ["--no-overlap"]])
def test_two_ranks_merge_to_the_single_stream_filter(extra):
    """a PLAIN `python bench.py --gpus 2` starts its own ranks (the driver's invocation form)"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, PSK_BENCH_SINGLE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spinup", "0", "--keys-per-rank", "2000000", *extra]
    line, _ = _run(cmd, env)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["cpu_baseline"] is None
    assert line["detail"]["all_inserted_found"] is True
    assert line["detail"]["merged_table_equals_single_stream"] is True
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["achieved"] > 0


def test_eight_ranks_on_one_device():
    """the driver's largest run, 8 ranks, with every rank on cuda:0 (gloo): eight shards, eight bit-range slices per exchange, one merged
    filter equal to the single-stream one -- the rank arithmetic of the N = 8 run is the same code"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, PSK_BENCH_SINGLE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--spinup", "0", "--keys-per-rank", "500000"]
    line, full = _run(cmd, env)
    assert line["n_gpus"] == 8 and line["rc"] == 0 and line["detail"]["all_inserted_found"] is True
    assert line["detail"]["merged_table_equals_single_stream"] is True
    assert line["multi_gpu"]["ranks_seen_by_rccl"] == 8 and line["multi_gpu"]["merge_ms_max"] > 0
    mg = full["multi_gpu"]  # what every rank measured for itself (round 4: the first real 8-GPU run must explain itself); detail file
    assert mg["ranks_seen_by_rccl"] == 8 and [p["rank"] for p in mg["per_rank"]] == list(range(8))
    assert all(p["insert_ms"] > 0 and p["merge_ms"] > 0 and p["check_ms"] > 0 for p in mg["per_rank"])
    assert mg["merge_GBs_per_gpu"] > 0 and mg["table_bytes"] == 2**28 // 8
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--config", "cfg5", "--n-total", "8000003", "--steps", "1", "--warmup", "1", "--spinup", "0"]
    line, full = _run(cmd, env)
    assert line["n_gpus"] == 8 and line["detail"]["all_inserted_found"] is True and line["detail"]["merged_prefix_equals_single_stream"] is True
    assert line["multi_gpu"]["ranks_seen_by_rccl"] == 8 and len(full["multi_gpu"]["per_rank"]) == 8 and full["multi_gpu"]["table_bytes"] == 2**31 // 8


def test_launched_under_torch_distributed_run():
    """the other launch form: python -m torch.distributed.run ... bench.py --gpus 2"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, PSK_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spinup", "0",
           "--keys-per-rank", "1000000"]
    line, _ = _run(cmd, env)
    assert line["n_gpus"] == 2 and line["detail"]["merged_table_equals_single_stream"] is True


def test_cfg5_two_ranks_through_the_real_allreduce_or():
    """cfg5 geometry (m = 2^31, 256 MiB replicas) through all_to_all + psk_or_reduce_slices + all_gather with two ranks:
    the merged prefix must equal the single-stream filter (checked inside bench.py), every inserted key is found"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, PSK_BENCH_SINGLE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--config", "cfg5", "--n-total", "6000001", "--steps", "1", "--warmup", "1", "--spinup", "0"]
    line, _ = _run(cmd, env)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["detail"]["all_inserted_found"] is True and line["detail"]["merged_prefix_equals_single_stream"] is True
    assert line["config"]["m_bits"] == 2**31


@pytest.mark.parametrize("cfg,extra", [("cfg3", ["--keys-per-rank", "1000000"]), ("cfg4", ["--batch", "200000", "--batches", "6"])])
def test_other_configs_emit_a_valid_line(cfg, extra):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--config", cfg, "--steps", "2", "--warmup", "1", "--spinup", "0", "--no-cpu-baseline", *extra]
    line, full = _run(cmd, env)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["achieved"] > 0
    assert cfg in line["config"]["workload"] and len(line["roofline"]["kernel"]) <= 80
    assert full["roofline"]["limiter"]  # the prose lives in the detail file


def test_default_run_line_fits_the_drivers_tail():
    """the driver's own command: the last stdout line must parse, carry roofline + cpu_baseline + the other three configurations, and fit"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    line, full = _run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], env)
    assert line["rc"] == 0 and line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and rf["peak"] == 8000.0 and rf["traffic"] and rf["avg_kernel_ms"] > 0
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] == "port" and cb["sample"]
    assert set(line["configs"]) == {"cfg3", "cfg4", "cfg5"} and all(c["parity_ok"] and c["value"] > 0 and c["frac"] > 0 for c in line["configs"].values())
    # the step's roofline and the wall-clock step agree on the bytes: frac x 8 TB/s x ms_per_step = 117 B x 10 M keys within 5 %
    assert rf["frac"] * 8e12 * line["ms_per_step"] * 1e-3 == pytest.approx(1.17e9, rel=0.05)
    assert set(full["rooflines"]) >= {"bloom_check", "cms_add", "cms_check", "cbf_add", "cbf_check"}
