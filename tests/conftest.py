import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _gpu_available():
    try:
        from spearmint_amd import engine
        return engine.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """GPU-marked tests are skipped on a box without a ROCm device or without libspx.so -- unless the
    run asks for them explicitly (-m gpu), where a missing GPU must fail loudly, not skip."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device / libspx.so (run with -m gpu on an MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
