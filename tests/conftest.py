import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, "anime-illust-image-searcher_amd")
for p in (ROOT, PKG_PARENT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run under gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the product library and the C oracle once per session if they are missing."""
    lib = os.path.join(PKG_PARENT, "libhip_tagsearch.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()
