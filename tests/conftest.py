import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle behind the same C ABI (test infrastructure; built by oracle/Makefile)."""
    from sdqlpy_amd import abi
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so"))


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "tpch_golden.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_more():
    """Reference results for queries beyond the configured five (q4, q14): make_golden.py --more."""
    with open(os.path.join(ROOT, "tests", "golden", "tpch_golden_more.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def hip_lib():
    """The product library.  On a box without a GPU only loading / symbol checks are possible."""
    from sdqlpy_amd import engine
    return engine.load_hip_library()


@pytest.fixture(scope="session")
def golden_wide():
    """Reference results for the queries that need the open expression vocabulary (q7, q8, q13, q15, q17, q19,
    q20, q22): make_golden.py --wide."""
    with open(os.path.join(ROOT, "tests", "golden", "tpch_golden_wide.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_sf1():
    """Reference results at SF=1 (6 M lineitem rows: every one of the reference's TPCH queries the interpreter finishes) and with keys
    beyond 2^40 at SF=0.1 / SF=1 — the sizes at which the product's size-dependent paths (device loops over result dictionaries, layout
    choices, walks, delta twins) engage: make_golden.py --sf1."""
    import gzip
    with gzip.open(os.path.join(ROOT, "tests", "golden", "tpch_golden_sf1.json.gz"), "rt") as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_sf10():
    """The reference's own results for the five configured queries at BASELINE.json's size, SF=10 (60 M lineitem rows): make_golden.py --sf10
    (a quarter of an hour of its interpreter per query).  Checked on the GPU box (tests/test_hip_parity.py) and by bench.py itself."""
    import gzip
    with gzip.open(os.path.join(ROOT, "tests", "golden", "tpch_golden_sf10.json.gz"), "rt") as fh:
        return json.load(fh)
