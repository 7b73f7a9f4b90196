import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/oracle.py + libmcd_oracle.so).  Test infrastructure only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def mcd():
    import mammo_clip_dissect_amd as m
    return m


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
