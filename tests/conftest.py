import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vc2-reference_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: multi-second CPU oracle runs")


@pytest.fixture(scope="session")
def oracle():
    from vc2lib import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def hip():
    """The product library through its C-ABI; fails loudly if missing or no GPU."""
    from vc2lib import load_hip
    return load_hip()
