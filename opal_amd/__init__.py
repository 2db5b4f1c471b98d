"""opal_amd -- MI355X-native inter-sequence database-search aligner behind the opal.h C API.

The product is the C-ABI shared library opal_amd/csrc/libopal.so (include/opal.h,
include/opal_hip.h). This package is the Python mirror of that interface for tests, bench and
scripting; it holds no compute of its own.
"""
from .api import *  # noqa: F401,F403
