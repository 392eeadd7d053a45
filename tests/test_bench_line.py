"""The driver reads bench.py's LAST stdout line out of an ~8 KB tail (BENCH_r05.json could not be parsed: the line had grown to 23 KB).
compact_line() must turn the full result object into a line well under that cap, with the contract's fields intact.  CPU only: the full
objects are the committed round-5 lines under profiles/."""

import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


@pytest.mark.parametrize("name,cfg", [("r05_bench_cfg2_default", "cfg2"), ("r05_bench_cfg2_forced_dist_1rank", "cfg2"), ("r05_bench_cfg3", "cfg3"),
                                      ("r05_bench_cfg4", "cfg4"), ("r05_bench_cfg5", "cfg5")])
def test_compact_line_fits_and_keeps_the_contract(name, cfg):
    import bench

    full = json.loads((ROOT / "profiles" / f"{name}.json").read_text().strip().splitlines()[-1])
    line = bench.compact_line(full, cfg, "gpurun_out/bench_detail.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT < 8000
    back = json.loads(text)
    assert all(k in back for k in CONTRACT)
    assert back["value"] == pytest.approx(full["value"], rel=1e-5) and back["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    rf = back["roofline"]
    assert set(rf) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"} and len(rf["kernel"]) <= 80
    assert rf["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5) and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    assert "workload" in back["config"] and "model" not in back["config"]
    if full.get("cpu_baseline"):
        assert set(back["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    if full.get("configs"):
        assert set(back["configs"]) == set(full["configs"])


def test_an_oversized_object_still_fits():
    import bench

    full = json.loads((ROOT / "profiles" / "r05_bench_cfg2_default.json").read_text().strip().splitlines()[-1])
    full["rooflines"] = {f"op{i}": dict(full["rooflines"]["cms_add"]) for i in range(200)}
    assert len(json.dumps(bench.compact_line(full, "cfg2", None), separators=(",", ":"))) < bench.LINE_LIMIT


def test_the_round_6_detail_object_keeps_the_per_key_latencies_on_the_line():
    """the FULL object of a default run (profiles/r06_bench_cfg2_default_detail.json, as bench.py wrote it on the GPU box) -> the compact line:
    the per-key call latencies travel on it, rounded, and it still fits"""
    import bench

    full = json.loads((ROOT / "profiles" / "r06_bench_cfg2_default_detail.json").read_text())
    line = bench.compact_line(full, "cfg2", "gpurun_out/bench_detail.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT
    pk = json.loads(text)["detail"]["per_key_call_us"]
    assert set(pk) == {"bloom_check", "cms_add", "cms_check", "cbf_add", "cbf_check"} and all(0 < v < 100 for v in pk.values())
