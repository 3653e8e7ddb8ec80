"""pytest configuration: markers and import paths.

``-m "not gpu"`` runs here (no GPU): oracle vs golden fixtures, host logic, C-ABI
export checks, world_size-2 gloo tests.  ``-m gpu`` runs on a real MI355X and calls
the HIP kernels through the C-ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
