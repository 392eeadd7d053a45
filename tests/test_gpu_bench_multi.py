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


@pytest.mark.parametrize("extra", [[], ["--no-overlap"]])
def test_two_ranks_merge_to_the_single_stream_filter(extra):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, PSK_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--keys-per-rank", "2000000",
           *extra]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    last = run.stdout.strip().splitlines()[-1]
    line = json.loads(last)  # the JSON line is the last thing on stdout
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["cpu_baseline"] is None
    assert line["detail"]["all_inserted_found"] is True
    assert line["detail"]["merged_table_equals_single_stream"] is True
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["achieved"] > 0
