"""Thread scaling of the host traceback (opalx_find_alignments, no GPU involved): OPAL_HOST_THREADS=k python scripts/host_traceback_scaling.py"""
import sys, time, ctypes as C, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, oracle_lib
from opal_amd import synth
from opal_amd.score_matrix import blosum50
from oracle_lib import OpalSearchResult
LIB=os.path.join(ROOT,'opal_amd','csrc','libopal.so')
lib = C.CDLL(LIB)
n=4000
res, offs, lens = synth.database(n, seed=42, first=50000)
db=[res[offs[i]:offs[i+1]] for i in range(n)]
q=synth.query(300, seed=7); m=blosum50()[1]
orc=oracle_lib.Oracle()
rc,s,eq,et=orc.search(q,db,3,1,m,1,3)
keep, dptrs, lens32 = oracle_lib._db_pointers(db)
qq=np.ascontiguousarray(q,np.uint8); mm=np.ascontiguousarray(m,np.int32)
libc=C.CDLL(None); libc.free.argtypes=[C.c_void_p]
for rep in range(3):
    arr=(OpalSearchResult*n)(); ptrs=(C.POINTER(OpalSearchResult)*n)()
    for i in range(n):
        arr[i].scoreSet=1; arr[i].score=int(s[i]); arr[i].endLocationQuery=int(eq[i]); arr[i].endLocationTarget=int(et[i]); ptrs[i]=C.pointer(arr[i])
    t0=time.perf_counter()
    rc=lib.opalx_find_alignments(qq.ctypes.data_as(C.POINTER(C.c_ubyte)), len(qq), dptrs, n, lens32.ctypes.data_as(C.POINTER(C.c_int)), 3, 1, mm.ctypes.data_as(C.POINTER(C.c_int)), 24, ptrs, 3)
    dt=time.perf_counter()-t0
    cells=float(((eq+1)*(et+1)).sum())
    print(f"threads={os.environ.get('OPAL_HOST_THREADS')} rep{rep}: {dt*1e3:.1f} ms, {cells/dt/1e9:.2f} Gcells/s")
    for r in arr:
        if r.alignment: libc.free(C.cast(r.alignment, C.c_void_p))
