"""bench.py must never hang silently or die without a result line: a rank that raises, or that hangs
past --timeout, still ends in ONE JSON line with "rc" != 0 and an "error" text -- in both launch forms.  The failures are
injected before any device is touched (PSK_BENCH_INJECT), so these run on the CPU-only box."""

import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(extra_args, inject, timeout=240):
    env = dict(os.environ, PSK_BENCH_INJECT=inject)
    env.pop("WORLD_SIZE", None)
    run = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0", *extra_args], capture_output=True, text=True,
                         timeout=timeout, env=env)
    lines = [x for x in run.stdout.strip().splitlines() if x.strip()]
    assert lines, run.stdout + run.stderr
    return run.returncode, json.loads(lines[-1])


def test_a_failing_single_rank_still_prints_a_json_line():
    rc, line = _run([], "raise")
    assert rc == 1 and line["rc"] == 1 and line["value"] is None
    assert "injected failure" in line["error"]


def test_a_hanging_rank_is_ended_by_the_watchdog_with_a_json_line():
    rc, line = _run(["--timeout", "3"], "hang")
    assert rc == 124 and line["rc"] == 124
    assert "watchdog" in line["error"]


def test_self_launched_ranks_that_fail_end_in_a_json_error_line():
    """python bench.py --gpus 2 (the driver's form for N > 1): rank 1 raises before the rendezvous, rank 0 then waits for it in
    vain -- its watchdog (or the elastic agent tearing the group down) must still leave a JSON line with rc != 0"""
    rc, line = _run(["--gpus", "2", "--timeout", "20", "--init-timeout", "10"], "raise:1", timeout=400)
    assert rc != 0
    assert line["rc"] != 0 and line["value"] is None and line["error"]
    # (round 4) the line names the rank that failed and the phase it was in, whoever ends up printing it
    assert "rank 1 failed in phase" in line["error"] and "injected failure" in line["error"], line["error"]
