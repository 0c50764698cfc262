import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (skipped when absent)")


def pytest_collection_modifyitems(config, items):
    """PyTorch bundles its own HIP runtime; libeasyav1_amd.so uses the system one.  When both live in
    one process the device must be initialised through torch FIRST (the other order leaves
    torch.cuda.is_available() false on this image), so a run that contains GPU tests touches
    torch.cuda before any test gets to load the library."""
    if "not gpu" in (config.getoption("markexpr", "") or ""):
        return
    if any(it.get_closest_marker("gpu") for it in items):
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()


@pytest.fixture(autouse=True)
def _oracle_tables_as_built():
    """the oracle's warp filter table is process-wide state that some tests replace (a caller's own table): every test starts
    with the built-in one (include/ea1_av1_tables.inc), as a fresh library context does"""
    from tests import oracle_lib as ol
    if getattr(ol, "_LIB", None) is not None or os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        try:
            ol.oracle_set_warp_filter(None)
        except Exception:
            pass
    yield
