import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of host CPU / tens of GB of host memory; opt in with GPSLC_RUN_SLOW=1")


@pytest.fixture(scope="session")
def gp():
    """The product package (HIP path).  Fails loudly if the library is missing."""
    import causalgpslc_jl_amd as gp_
    gp_.load_library()
    return gp_


@pytest.fixture(scope="session")
def oracle():
    import gpslc_oracle
    return gpslc_oracle
