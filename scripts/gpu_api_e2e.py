"""End-to-end wall time of the plain opalSearchDatabase entry (pack + H2D + kernels + D2H + scatter) on the C2 workload."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from opal_amd import synth
from opal_amd.score_matrix import blosum50
n = int(os.environ.get("NSEQ", "100000"))
res, offs, lens = synth.database(n, seed=42)
q = synth.query(300, seed=7); m = np.ascontiguousarray(blosum50()[1], np.int32)
abi = oracle_lib.OpalABI(os.path.join(ROOT, "opal_amd", "csrc", "libopal.so"))
base = res.ctypes.data
ptrs = (C.POINTER(C.c_ubyte) * n)()
for i in range(n):
    ptrs[i] = C.cast(base + int(offs[i]), C.POINTER(C.c_ubyte))
ln = np.ascontiguousarray(lens, np.int32)
cells = 300.0 * float(offs[-1])
for st in (0, 1):
    for it in range(3):
        arr, rptrs = abi.new_results(n)
        t0 = time.perf_counter()
        rc = abi.f_search(q.ctypes.data_as(C.POINTER(C.c_ubyte)), 300, ptrs, n, ln.ctypes.data_as(C.POINTER(C.c_int)), 11, 1,
                          m.ctypes.data_as(C.POINTER(C.c_int)), 24, rptrs, st, 3, 1)
        dt = time.perf_counter() - t0
        print(f"opalSearchDatabase searchType={st} rc={rc}: {dt*1e3:.1f} ms -> {cells/dt/1e9:.1f} GCUPS end to end (score[0]={arr[0].score})", flush=True)
