"""Python mirror of the reference's interface for the database-search path, on top of the
C-ABI library (include/opal.h, include/opal_hip.h).

Names, argument meaning and error behaviour follow reference src/opal.h:17-165:
``search_database`` is ``opalSearchDatabase`` (same argument order after the arrays),
``search_database_char_sw`` is ``opalSearchDatabaseCharSW``; constants keep their names.
``PreparedDb`` wraps the additive opalx_* API (database packed once, resident in HBM,
results written to device buffers) used by bench.py and by callers with many queries.
"""
import ctypes as C

import numpy as np

from . import _lib

OPAL_ERR_OVERFLOW, OPAL_ERR_NO_SIMD_SUPPORT, OPAL_ERR_INVALID_MODE = 1, 2, 3
OPAL_MODE_NW, OPAL_MODE_HW, OPAL_MODE_OV, OPAL_MODE_SW = 0, 1, 2, 3
OPAL_OVERFLOW_SIMPLE, OPAL_OVERFLOW_BUCKETS = 0, 1
OPAL_SEARCH_SCORE, OPAL_SEARCH_SCORE_END, OPAL_SEARCH_ALIGNMENT = 0, 1, 2
OPAL_ALIGN_MATCH, OPAL_ALIGN_DEL, OPAL_ALIGN_INS, OPAL_ALIGN_MISMATCH = 0, 1, 2, 3
MODES = {"NW": OPAL_MODE_NW, "HW": OPAL_MODE_HW, "OV": OPAL_MODE_OV, "SW": OPAL_MODE_SW}


class OpalSearchResult(C.Structure):
    """struct OpalSearchResult (reference src/opal.h:47-74): 40 bytes, same field order."""
    _fields_ = [("scoreSet", C.c_int), ("score", C.c_int),
                ("endLocationTarget", C.c_int), ("endLocationQuery", C.c_int),
                ("startLocationTarget", C.c_int), ("startLocationQuery", C.c_int),
                ("alignment", C.POINTER(C.c_ubyte)), ("alignmentLength", C.c_int)]


def new_results(n):
    """n initialised results (opalInitSearchResult) and the array of pointers the C API takes."""
    L = _lib.lib()
    arr = (OpalSearchResult * n)()
    ptrs = (C.POINTER(OpalSearchResult) * n)()
    for i in range(n):
        L.opalInitSearchResult(C.byref(arr[i]))
        ptrs[i] = C.pointer(arr[i])
    return arr, ptrs


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _db_arrays(db):
    seqs = [_u8(s) for s in db]
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    ptrs = (C.POINTER(C.c_ubyte) * len(seqs))()
    dummy = np.zeros(1, dtype=np.uint8)
    for i, s in enumerate(seqs):
        ptrs[i] = (s if len(s) else dummy).ctypes.data_as(C.POINTER(C.c_ubyte))
    return (seqs, dummy), ptrs, lens


def set_option(name, value=None):
    """opalx_set_option: change one of the library's options (names of environment variables, e.g. OPAL_HIP_DEVICES; value
    None = back to the default). The library reads the environment only once, at its first use. Raises on an unknown name
    or a value that does not parse."""
    rc = _lib.lib().opalx_set_option(name.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise ValueError(f"opalx_set_option({name!r}, {value!r}) -> {rc} ({'unknown option' if rc == 1 else 'invalid value'})")


def search_database(query, db, gap_open, gap_ext, score_matrix, search_type=OPAL_SEARCH_SCORE,
                    mode=OPAL_MODE_SW, overflow_method=OPAL_OVERFLOW_BUCKETS, results=None):
    """opalSearchDatabase. `db` is a list of uint8 index arrays. Returns (rc, results_array)."""
    L = _lib.lib()
    q = _u8(query)
    keep, ptrs, lens = _db_arrays(db)
    m = np.ascontiguousarray(score_matrix, dtype=np.int32)
    arr, rptrs = results if results is not None else new_results(len(lens))
    rc = L.opalSearchDatabase(q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), ptrs, len(lens),
                              lens.ctypes.data_as(C.POINTER(C.c_int)), int(gap_open), int(gap_ext),
                              m.ctypes.data_as(C.POINTER(C.c_int)), int(m.shape[0]), rptrs,
                              int(search_type), int(mode), int(overflow_method))
    return rc, (arr, rptrs)


def search_database_char_sw(query, db, gap_open, gap_ext, score_matrix, results=None):
    """opalSearchDatabaseCharSW. Returns (rc, results_array)."""
    L = _lib.lib()
    q = _u8(query)
    keep, ptrs, lens = _db_arrays(db)
    m = np.ascontiguousarray(score_matrix, dtype=np.int32)
    arr, rptrs = results if results is not None else new_results(len(lens))
    rc = L.opalSearchDatabaseCharSW(q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), ptrs, len(lens),
                                    lens.ctypes.data_as(C.POINTER(C.c_int)), int(gap_open), int(gap_ext),
                                    m.ctypes.data_as(C.POINTER(C.c_int)), int(m.shape[0]), rptrs)
    return rc, (arr, rptrs)


def result_fields(results, *names):
    arr = results[0]
    return [np.array([getattr(arr[i], n) for i in range(len(arr))], dtype=np.int32) for n in names]


class PreparedDb:
    """A database packed once and kept in device memory (opalx_db_* in include/opal_hip.h)."""

    def __init__(self, residues, offsets, device_ptr=None):
        """`residues`: uint8 host array (flat); or, with `device_ptr` (an int: address of the same flat buffer
        already in device memory, e.g. a torch uint8 tensor's data_ptr()), nothing is uploaded."""
        L = _lib.lib()
        self._off = np.ascontiguousarray(offsets, dtype=np.int64)
        self.num_sequences = len(self._off) - 1
        if device_ptr is not None:
            self._h = L.opalx_db_create_flat_device(C.c_void_p(device_ptr), self._off.ctypes.data_as(C.POINTER(C.c_int64)),
                                                    self.num_sequences)
        else:
            self._res = _u8(residues)
            self._h = L.opalx_db_create_flat(self._res.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                             self._off.ctypes.data_as(C.POINTER(C.c_int64)), self.num_sequences)
        if not self._h:
            raise RuntimeError("opalx_db_create_flat failed: " + (L.opal_hip_last_error() or b"").decode())
        self.num_residues = int(L.opalx_db_num_residues(self._h))

    @classmethod
    def from_sequences(cls, seqs):
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        flat = np.concatenate([_u8(s) for s in seqs]) if len(seqs) else np.zeros(0, np.uint8)
        return cls(flat, offs)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:
            _lib._lib.opalx_db_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def search_device(self, query, gap_open, gap_ext, score_matrix, search_type, mode, d_score, d_end_q=0, d_end_t=0,
                      overflow_method=OPAL_OVERFLOW_BUCKETS):
        """Results into device buffers given as raw pointers (e.g. torch tensor .data_ptr()).
        Returns (rc, kernel_ms)."""
        q = _u8(query)
        m = np.ascontiguousarray(score_matrix, dtype=np.int32)
        ms = C.c_float(0)
        rc = _lib.lib().opalx_search_device(self._h, q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), int(gap_open),
                                            int(gap_ext), m.ctypes.data_as(C.POINTER(C.c_int)), int(m.shape[0]),
                                            int(search_type), int(mode), int(overflow_method),
                                            C.c_void_p(d_score), C.c_void_p(d_end_q or None), C.c_void_p(d_end_t or None),
                                            C.byref(ms))
        return rc, ms.value

    def search_device_async(self, query, gap_open, gap_ext, score_matrix, search_type, mode, d_score, d_end_q=0, d_end_t=0):
        """opalx_search_device_async: enqueue the search and return (results are complete after wait()). Returns rc."""
        q = _u8(query)
        m = np.ascontiguousarray(score_matrix, dtype=np.int32)
        # (the library copies query and matrix into its pinned search header before the call returns: nothing to keep alive)
        return _lib.lib().opalx_search_device_async(self._h, q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), int(gap_open), int(gap_ext),
                                                    m.ctypes.data_as(C.POINTER(C.c_int)), int(m.shape[0]), int(search_type), int(mode),
                                                    C.c_void_p(d_score), C.c_void_p(d_end_q or None), C.c_void_p(d_end_t or None))

    def wait(self):
        """opalx_db_wait: wait for the OLDEST search enqueued by search_device_async and not yet waited for (up to three may be
        queued on a handle). Returns (rc, kernel_ms of that search)."""
        ms = C.c_float(0)
        rc = _lib.lib().opalx_db_wait(self._h, C.byref(ms))
        return rc, ms.value

    def wait2(self):
        """opalx_db_wait2: as wait(), plus the device time between the end of the search waited for before this one and the end
        of this one (queued searches overlap on two streams: what a search cost the queue). Returns (rc, kernel_ms, period_ms)."""
        ms = C.c_float(0); per = C.c_float(0)
        rc = _lib.lib().opalx_db_wait2(self._h, C.byref(ms), C.byref(per))
        return rc, ms.value, per.value

    def search_starts_device(self, query, gap_open, gap_ext, score_matrix, mode, d_score, d_end_q, d_end_t, d_start_q, d_start_t):
        """opalx_search_starts: start cells on the device from the score / end-cell arrays a SCORE_END search of the
        same query left in device memory (raw pointers). Returns (rc, redo) -- `redo` = indices the host must redo."""
        q = _u8(query)
        m = np.ascontiguousarray(score_matrix, dtype=np.int32)
        redo = np.zeros(max(self.num_sequences, 1), np.int32)
        cnt = C.c_int(0)
        rc = _lib.lib().opalx_search_starts(self._h, q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), int(gap_open), int(gap_ext),
                                            m.ctypes.data_as(C.POINTER(C.c_int)), int(m.shape[0]), int(mode),
                                            C.c_void_p(d_score), C.c_void_p(d_end_q), C.c_void_p(d_end_t),
                                            C.c_void_p(d_start_q), C.c_void_p(d_start_t),
                                            redo.ctypes.data_as(C.POINTER(C.c_int)), C.byref(cnt))
        return rc, redo[: cnt.value].copy()

    def search(self, query, gap_open, gap_ext, score_matrix, search_type=OPAL_SEARCH_SCORE, mode=OPAL_MODE_SW,
               overflow_method=OPAL_OVERFLOW_BUCKETS):
        """Host-array convenience: returns (rc, score, endQ, endT) int32 arrays."""
        q = _u8(query)
        m = np.ascontiguousarray(score_matrix, dtype=np.int32)
        n = self.num_sequences
        s = np.zeros(n, np.int32); eq = np.full(n, -1, np.int32); et = np.full(n, -1, np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        rc = _lib.lib().opalx_search_host(self._h, q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), int(gap_open),
                                          int(gap_ext), ip(m), int(m.shape[0]), int(search_type), int(mode),
                                          int(overflow_method), ip(s), ip(eq), ip(et))
        return rc, s, eq, et


class PreparedMultiDb:
    """One database over several GPUs of this process (opalx_multi_* in include/opal_hip.h): residue-balanced shards, one
    search on all GPUs at once, the result blocks gathered on devices[0] by peer DMA and written there in database order."""

    def __init__(self, residues, offsets, devices):
        L = _lib.lib()
        self._res = _u8(residues)
        self._off = np.ascontiguousarray(offsets, dtype=np.int64)
        self.num_sequences = len(self._off) - 1
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        self.devices = [int(d) for d in dev]
        self._h = L.opalx_multi_create_flat(self._res.ctypes.data_as(C.POINTER(C.c_ubyte)), self._off.ctypes.data_as(C.POINTER(C.c_int64)),
                                            self.num_sequences, dev.ctypes.data_as(C.POINTER(C.c_int)), len(dev))
        if not self._h:
            raise RuntimeError("opalx_multi_create_flat failed: " + (L.opal_hip_last_error() or b"").decode())

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:
            _lib._lib.opalx_multi_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def search_device(self, query, gap_open, gap_ext, score_matrix, search_type, mode, d_score, d_end_q=0, d_end_t=0):
        """Results into device buffers on devices[0] (raw pointers), database order. Returns (rc, kernel_ms)."""
        q = _u8(query)
        m = np.ascontiguousarray(score_matrix, dtype=np.int32)
        ms = C.c_float(0)
        rc = _lib.lib().opalx_multi_search_device(self._h, q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), int(gap_open), int(gap_ext),
                                                  m.ctypes.data_as(C.POINTER(C.c_int)), int(m.shape[0]), int(search_type), int(mode),
                                                  C.c_void_p(d_score), C.c_void_p(d_end_q or None), C.c_void_p(d_end_t or None), C.byref(ms))
        return rc, ms.value

    def search_device_async(self, query, gap_open, gap_ext, score_matrix, search_type, mode, d_score, d_end_q=0, d_end_t=0):
        """opalx_multi_search_device_async: enqueue the search on every shard and the gather behind it (results are complete
        after wait()). Up to three may be queued, each into result buffers of its own. Returns rc."""
        q = _u8(query)
        m = np.ascontiguousarray(score_matrix, dtype=np.int32)
        return _lib.lib().opalx_multi_search_device_async(self._h, q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), int(gap_open), int(gap_ext),
                                                          m.ctypes.data_as(C.POINTER(C.c_int)), int(m.shape[0]), int(search_type), int(mode),
                                                          C.c_void_p(d_score), C.c_void_p(d_end_q or None), C.c_void_p(d_end_t or None))

    def wait(self):
        """opalx_multi_wait: the OLDEST queued search (first in, first out). Returns (rc, kernel_ms = the longest over the shards)."""
        ms = C.c_float(0)
        rc = _lib.lib().opalx_multi_wait(self._h, C.byref(ms))
        return rc, ms.value
