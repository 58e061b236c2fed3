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
def hip_hooks_lib():
    """The TEST build of the product library (libjslp_hip_chaos.so: -DJSLP_CHAOS_BUILD, built by __graft_entry__.build()): the same
    sources with the test hooks of the register-resident kernels compiled in -- JSLP_TEST_RESIDENT_ABORT, JSLP_TEST_RESIDENT_LATE_WAVE0 --
    which the shipped library does not carry (they cost the headline kernel 4 % of its pivot rate).  Fails when missing."""
    from jslpsolver_amd import _capi
    path = os.path.join(ROOT, "jslpsolver_amd", "csrc", "libjslp_hip_chaos.so")
    assert os.path.exists(path), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    return _capi.Library(path)


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests fail (not skip) when it is missing."""
    from jslpsolver_amd import _capi
    return _capi.load_hip()
