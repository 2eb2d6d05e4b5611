import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # -m gpu tests must fail loudly on a box without a GPU rather than silently pass; but when the
    # user did not ask for them (-m "not gpu") they are deselected by the marker expression itself.
    pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
