import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference path (test infrastructure, oracle/)."""
    from tests import oracle_lib
    orc = oracle_lib.load()
    # the oracle's OpenMP loops stop scaling around 32 threads (bench.py picks that on a 256-CPU box); the default of one
    # thread per logical CPU makes the many small proofs of the test-suite slower, not faster
    orc.set_threads(min(32, os.cpu_count() or 1))
    return orc


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
