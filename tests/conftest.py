import os
import sys

import pytest

# the oracle's OpenMP loops are tiny per region: on a 256-thread host the fork/join cost dominates (the 500-step drift
# test went from ~1 to ~8 minutes); must be set before libgomp loads
os.environ.setdefault("OMP_NUM_THREADS", "32")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
