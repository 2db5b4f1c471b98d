"""ctypes bindings for the CPU checkers under oracle/ (test infrastructure only).

  Oracle   -> oracle/liboracle.so        (opal_oracle.c, the scalar restatement)
  RefLib   -> oracle/_ref/libopal_ref.so (the reference's own SIMD library, symbols renamed ref_*;
                                          built in the build container by oracle/build_ref.sh)
Nothing under opal_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

MODE_NW, MODE_HW, MODE_OV, MODE_SW = 0, 1, 2, 3
SEARCH_SCORE, SEARCH_SCORE_END, SEARCH_ALIGNMENT = 0, 1, 2
OVERFLOW_SIMPLE, OVERFLOW_BUCKETS = 0, 1
MODES = {"NW": MODE_NW, "HW": MODE_HW, "OV": MODE_OV, "SW": MODE_SW}


class OpalSearchResult(C.Structure):
    _fields_ = [("scoreSet", C.c_int), ("score", C.c_int),
                ("endLocationTarget", C.c_int), ("endLocationQuery", C.c_int),
                ("startLocationTarget", C.c_int), ("startLocationQuery", C.c_int),
                ("alignment", C.POINTER(C.c_ubyte)), ("alignmentLength", C.c_int)]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def _as_u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _db_pointers(seqs):
    """list of uint8 arrays -> (keepalive, c_array of pointers, int32 lengths)."""
    seqs = [_as_u8(s) for s in seqs]
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    ptrs = (C.POINTER(C.c_ubyte) * len(seqs))()
    dummy = np.zeros(1, dtype=np.uint8)
    for i, s in enumerate(seqs):
        ptrs[i] = (s if len(s) else dummy).ctypes.data_as(C.POINTER(C.c_ubyte))
    return (seqs, dummy), ptrs, lens


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        self.lib = C.CDLL(path)
        self.lib.oracle_search.restype = C.c_int
        self.lib.oracle_char_sw.restype = C.c_int
        self.lib.oracle_check_alignment.restype = C.c_int

    def search(self, query, seqs, gap_open, gap_ext, matrix, search_type, mode):
        """-> (rc, score, endQ, endT) int32 arrays."""
        q = _as_u8(query)
        keep, ptrs, lens = _db_pointers(seqs)
        m = np.ascontiguousarray(matrix, dtype=np.int32)
        n = len(lens)
        score = np.zeros(n, np.int32); eq = np.full(n, -1, np.int32); et = np.full(n, -1, np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        rc = self.lib.oracle_search(q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), ptrs, n, ip(lens),
                                    int(gap_open), int(gap_ext), ip(m), m.shape[0],
                                    int(search_type), int(mode), ip(score), ip(eq), ip(et))
        return rc, score, eq, et

    def char_sw(self, query, seqs, gap_open, gap_ext, matrix):
        q = _as_u8(query)
        keep, ptrs, lens = _db_pointers(seqs)
        m = np.ascontiguousarray(matrix, dtype=np.int32)
        n = len(lens)
        score = np.zeros(n, np.int32); sset = np.zeros(n, np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        rc = self.lib.oracle_char_sw(q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), ptrs, n, ip(lens),
                                     int(gap_open), int(gap_ext), ip(m), m.shape[0], ip(score), ip(sset))
        return rc, score, sset

    def check_alignment(self, query, target, gap_open, gap_ext, matrix, score, start_q, start_t, end_q, end_t, ops):
        q = _as_u8(query); t = _as_u8(target); o = _as_u8(ops)
        m = np.ascontiguousarray(matrix, dtype=np.int32)
        up = lambda a: a.ctypes.data_as(C.POINTER(C.c_ubyte))
        return self.lib.oracle_check_alignment(up(q), len(q), up(t), len(t), int(gap_open), int(gap_ext),
                                               m.ctypes.data_as(C.POINTER(C.c_int)), m.shape[0], int(score),
                                               int(start_q), int(start_t), int(end_q), int(end_t), up(o), len(o))


class OpalABI:
    """Any library exporting the opal.h C API under `prefix` (""), e.g. ours or the renamed reference ("ref_")."""

    def __init__(self, path, prefix=""):
        self.lib = C.CDLL(path)
        g = lambda n: getattr(self.lib, prefix + n)
        self.f_search = g("opalSearchDatabase"); self.f_search.restype = C.c_int
        self.f_char = g("opalSearchDatabaseCharSW"); self.f_char.restype = C.c_int
        self.f_init = g("opalInitSearchResult"); self.f_init.restype = None
        self.f_empty = g("opalSearchResultIsEmpty"); self.f_empty.restype = C.c_int
        self.f_empty.argtypes = [OpalSearchResult]
        self.f_set = g("opalSearchResultSetScore"); self.f_set.restype = None

    def new_results(self, n):
        arr = (OpalSearchResult * n)()
        ptrs = (C.POINTER(OpalSearchResult) * n)()
        for i in range(n):
            self.f_init(C.byref(arr[i]))
            ptrs[i] = C.pointer(arr[i])
        return arr, ptrs

    def search(self, query, seqs, gap_open, gap_ext, matrix, search_type, mode, overflow=OVERFLOW_BUCKETS,
               results=None):
        """-> (rc, score, endQ, endT, scoreSet) arrays plus the raw result structs."""
        q = _as_u8(query)
        keep, ptrs, lens = _db_pointers(seqs)
        m = np.ascontiguousarray(matrix, dtype=np.int32)
        n = len(lens)
        arr, rptrs = results if results is not None else self.new_results(n)
        rc = self.f_search(q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), ptrs, n,
                           lens.ctypes.data_as(C.POINTER(C.c_int)), int(gap_open), int(gap_ext),
                           m.ctypes.data_as(C.POINTER(C.c_int)), m.shape[0], rptrs,
                           int(search_type), int(mode), int(overflow))
        f = lambda name: np.array([getattr(arr[i], name) for i in range(n)], dtype=np.int32)
        return rc, f("score"), f("endLocationQuery"), f("endLocationTarget"), f("scoreSet"), (arr, rptrs)

    def alignments(self, arr):
        """(startQ, startT, [ops bytes]) from result structs after an OPAL_SEARCH_ALIGNMENT search; frees them."""
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        sq, st, ops = [], [], []
        for r in arr:
            sq.append(r.startLocationQuery); st.append(r.startLocationTarget)
            if r.alignment and r.alignmentLength > 0:
                ops.append(bytes(bytearray(r.alignment[:r.alignmentLength])))
            else:
                ops.append(b"")
            if r.alignment:
                libc.free(C.cast(r.alignment, C.c_void_p))
                r.alignment = None
        return np.array(sq, np.int32), np.array(st, np.int32), ops

    def char_sw(self, query, seqs, gap_open, gap_ext, matrix):
        q = _as_u8(query)
        keep, ptrs, lens = _db_pointers(seqs)
        m = np.ascontiguousarray(matrix, dtype=np.int32)
        n = len(lens)
        arr, rptrs = self.new_results(n)
        rc = self.f_char(q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), ptrs, n,
                         lens.ctypes.data_as(C.POINTER(C.c_int)), int(gap_open), int(gap_ext),
                         m.ctypes.data_as(C.POINTER(C.c_int)), m.shape[0], rptrs)
        f = lambda name: np.array([getattr(arr[i], name) for i in range(n)], dtype=np.int32)
        return rc, f("score"), f("scoreSet")


def ref_lib_path():
    return os.path.join(ORACLE_DIR, "_ref", "libopal_ref.so")


def have_ref():
    return os.path.exists(ref_lib_path())


def RefLib():
    return OpalABI(ref_lib_path(), prefix="ref_")


class RefThreads:
    """The reference's own AVX2 library on every host core (oracle/ref_bench.c: ref_search_run): what the full-size GPU
    tests compare EVERY sequence of a BASELINE-sized database with ("bit-exact ... for every sequence").
    The scalar oracle cannot finish 1e10 cells in seconds; this does in well under one."""

    def __init__(self):
        bench_so = os.path.join(ORACLE_DIR, "_ref", "libref_bench.so")
        self.ok = have_ref() and os.path.exists(bench_so)
        if self.ok:
            C.CDLL(ref_lib_path(), mode=C.RTLD_GLOBAL)
            self.lib = C.CDLL(bench_so)
            self.ok = hasattr(self.lib, "ref_search_run")
        if self.ok:
            self.lib.ref_search_run.restype = C.c_double

    def search(self, query, res, offs, matrix, gap_open, gap_ext, search_type, mode_code, threads=None):
        """-> (rc, score, endQ, endT, seconds) over all len(offs) - 1 sequences"""
        n = len(offs) - 1
        if threads is None:
            threads = max(1, min(os.cpu_count() or 1, 64))
        q = np.ascontiguousarray(query, np.uint8); o = np.ascontiguousarray(offs, np.int64)
        m = np.ascontiguousarray(matrix, np.int32); r = np.ascontiguousarray(res, np.uint8)
        s = np.empty(n, np.int32); eq = np.empty(n, np.int32); et = np.empty(n, np.int32)
        rc = C.c_int(0)
        up = lambda a: a.ctypes.data_as(C.POINTER(C.c_ubyte))
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        sec = self.lib.ref_search_run(up(q), len(q), up(r), o.ctypes.data_as(C.POINTER(C.c_int64)), int(n), int(gap_open),
                                      int(gap_ext), ip(m), int(m.shape[0]), int(mode_code), int(search_type), int(threads),
                                      ip(s), ip(eq), ip(et), C.byref(rc))
        return rc.value, s, eq, et, sec
