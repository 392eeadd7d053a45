"""pytest configuration: registers the `gpu` marker; puts the repo root on sys.path."""

import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """fixtures produced by the REAL reference (tests/golden/gen_golden.py)"""
    return json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())


@pytest.fixture(scope="session")
def oracle():
    """the plain-C CPU restatement (test infrastructure)"""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as _oracle  # noqa: PLC0415

    _oracle.build()
    return _oracle


@pytest.fixture(scope="session", autouse=True)
def _engine_options_from_env():
    """PSK_PARTITION_MIN_KEYS=1 pytest -m gpu ...  re-runs the whole GPU suite through the partitioned path"""
    import os

    val = os.environ.get("PSK_PARTITION_MIN_KEYS")
    if val is not None:
        from pyprobables_amd import _native as N

        N.set_option("partition_min_keys", int(val))
    # PSK_TEST_OPTS="cbf_lookup_shadow=0,auto_combine=0" pytest -m gpu ...: the suite under other engine defaults (robustness sweeps; the
    # tests that pin a default's own mechanics -- image loads counted, batches left waiting -- are expected to object, parity must not)
    for kv in os.environ.get("PSK_TEST_OPTS", "").split(","):
        if "=" in kv:
            from pyprobables_amd import _native as N

            N.set_option(kv.split("=")[0].strip(), int(kv.split("=")[1], 0))
    yield
