import os
import sys

import pytest

# a re-shot eigenray that does not end on the bits of the trial ray the search accepted is an ERROR in the tests (a warning
# and a recorded statistic for users): pygenray_amd/eigenrays.py
os.environ.setdefault("PGR_EIGEN_STRICT", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
