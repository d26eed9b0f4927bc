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


@pytest.fixture(autouse=True)
def _product_library_by_default():
    """Every test starts and ends on the PRODUCT library (libmspmv.so).  A test that forces a code path calls one of the setters of
    include/mspmv_dev.h, which switches merge_spmv_amd to libmspmv_dev.so (same kernels + the per-thread overrides); whatever it
    left there is reset here."""
    yield
    M = sys.modules.get("merge_spmv_amd")
    if M is None or not hasattr(M, "active_library"):
        return
    if M.active_library() != "product" or "dev" in M._libs:
        try:
            M.use_library("dev")
            for vb in (4, 8):
                M.set_tuning(vb); M.set_band_passes(vb, 0); M.set_tdm(vb, 0)
            M.set_record_polls(0); M.set_compact_tiles(0)
        except Exception:  # noqa: BLE001 - (no development library in this checkout: nothing to reset)
            pass
        M.use_library("product")
