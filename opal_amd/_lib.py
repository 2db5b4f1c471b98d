"""Loader for the C-ABI library (opal_amd/csrc/libopal.so). No fallback: if the HIP
library is missing or cannot be loaded the import of anything that computes fails loudly."""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OPAL_LIB_PATH") or os.path.join(_HERE, "csrc", "libopal.so")

_lib = None


class OpalLibraryMissing(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OpalLibraryMissing(
                f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C opal_amd/csrc` (there is no CPU fallback)")
        # PyTorch-ROCm wheels bundle their own HIP runtime (same SONAME as /opt/rocm's). Two HIP
        # runtimes in one process cannot both own the GPU, so when torch is installed it is imported
        # first: libopal.so then binds to the runtime torch already loaded. C/C++ callers are unaffected.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        i32p, u8p = C.POINTER(C.c_int), C.POINTER(C.c_ubyte)
        L.opalx_db_create_flat.restype = C.c_void_p
        L.opalx_db_create_flat.argtypes = [u8p, C.POINTER(C.c_int64), C.c_int]
        L.opalx_db_create_flat_device.restype = C.c_void_p
        L.opalx_db_create_flat_device.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
        L.opalx_release_cached.restype = None
        L.opalx_release_cached.argtypes = []
        L.opalx_db_destroy.restype = None
        L.opalx_db_destroy.argtypes = [C.c_void_p]
        L.opalx_db_num_sequences.restype = C.c_int
        L.opalx_db_num_sequences.argtypes = [C.c_void_p]
        L.opalx_db_num_residues.restype = C.c_int64
        L.opalx_db_num_residues.argtypes = [C.c_void_p]
        L.opalx_search_device.restype = C.c_int
        L.opalx_search_device.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.opalx_search_device_async.restype = C.c_int
        L.opalx_search_device_async.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p]
        L.opalx_db_wait.restype = C.c_int
        L.opalx_db_wait.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.opalx_db_wait2.restype = C.c_int
        L.opalx_db_wait2.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.opalx_search_starts.restype = C.c_int
        L.opalx_search_starts.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, i32p, i32p]
        L.opalx_search_host.restype = C.c_int
        L.opalx_search_host.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, i32p, i32p, i32p]
        L.opalx_multi_create_flat.restype = C.c_void_p
        L.opalx_multi_create_flat.argtypes = [u8p, C.POINTER(C.c_int64), C.c_int, i32p, C.c_int]
        L.opalx_multi_destroy.restype = None
        L.opalx_multi_destroy.argtypes = [C.c_void_p]
        L.opalx_multi_search_device.restype = C.c_int
        L.opalx_multi_search_device.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.opalx_multi_search_device_async.restype = C.c_int
        L.opalx_multi_search_device_async.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, C.c_int,
                                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.opalx_multi_wait.restype = C.c_int
        L.opalx_multi_wait.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.opalx_set_option.restype = C.c_int
        L.opalx_set_option.argtypes = [C.c_char_p, C.c_char_p]
        L.opalx_counters.restype = C.c_int
        L.opalx_counters.argtypes = [C.POINTER(C.c_longlong), C.c_int]
        L.opal_hip_device_count.restype = C.c_int
        L.opal_hip_set_device.restype = C.c_int
        L.opal_hip_set_device.argtypes = [C.c_int]
        L.opal_hip_last_error.restype = C.c_char_p
        L.opalSearchDatabase.restype = C.c_int
        L.opalSearchDatabaseCharSW.restype = C.c_int
        _lib = L
    return _lib
