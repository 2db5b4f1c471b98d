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


class _OpalOptions:
    """The library's options from inside the process (opalx_set_option, include/opal_hip.h): the library reads the
    environment once, so tests steer the scheduler through this instead of os.environ. Everything set is put back to its
    default when the test ends."""

    def __init__(self):
        import ctypes as C
        from opal_amd import _lib
        self._f = _lib.lib().opalx_set_option
        self._f.restype = C.c_int
        self._f.argtypes = [C.c_char_p, C.c_char_p]
        self._touched = set()

    def set(self, name, value):
        rc = self._f(name.encode(), str(value).encode())
        assert rc == 0, f"opalx_set_option({name}, {value}) -> {rc}"
        self._touched.add(name)

    def unset(self, name):
        assert self._f(name.encode(), None) == 0, name

    def reset(self):
        for name in self._touched:
            self._f(name.encode(), None)
        self._touched.clear()


@pytest.fixture
def opal_opt():
    o = _OpalOptions()
    yield o
    o.reset()
