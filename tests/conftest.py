"""pytest configuration: registers the `gpu` marker and puts the repo root on
sys.path.  `-m "not gpu"` runs everywhere; `-m gpu` needs a real MI355X."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_matrices():
    return load_golden("matrices.json")["cases"]


@pytest.fixture(scope="session")
def golden_host():
    return load_golden("host_semantics.json")


@pytest.fixture(scope="session")
def golden_kat():
    return load_golden("kat_device_spmv.json")
