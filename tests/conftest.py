import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# PyTorch-ROCm wheels bundle their own HIP runtime (same SONAME as /opt/rocm's): whichever is loaded first owns the GPU.
# Tests that load libopal.so directly (ctypes) and tests that use torch.cuda run in one process, so torch goes first --
# whatever the test order or selection (opal_amd/_lib.py does the same for library users).
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build_oracle()
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def golden():
    import golden_io
    return golden_io.load_all()
