import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


@pytest.fixture(scope="session")
def oracle_lib():
    """TEST-ONLY CPU restatement (oracle/libjslp_oracle.so) behind the product's C ABI."""
    from jslpsolver_amd import _capi
    path = os.path.join(ROOT, "oracle", "libjslp_oracle.so")
    src = os.path.join(ROOT, "oracle", "jslp_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libjslp_oracle.so"])
    return _capi.Library(path)


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests fail (not skip) when it is missing."""
    from jslpsolver_amd import _capi
    return _capi.load_hip()
