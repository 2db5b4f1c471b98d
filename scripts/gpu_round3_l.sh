#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/ord.py <<'PY'
import ctypes as C, numpy as np, sys, time, os
sys.path.insert(0,'.')
from opal_amd import synth
L = C.CDLL('opal_amd/csrc/libopal.so')
L.opalx_debug_order.restype = C.c_int
for n in (100000, 625000):
    lens = synth.lengths(42,0,n)
    cap=n+2
    seq=np.full(cap,-7,np.int32); ln=np.full(cap,-7,np.int32); off=np.zeros(cap,np.uint64)
    args=(lens.ctypes.data_as(C.POINTER(C.c_int)), n, seq.ctypes.data_as(C.POINTER(C.c_int)), ln.ctypes.data_as(C.POINTER(C.c_int)), off.ctypes.data_as(C.POINTER(C.c_ulonglong)), cap)
    ts=[]
    for i in range(60):
        t0=time.perf_counter(); L.opalx_debug_order(*args); ts.append((time.perf_counter()-t0)*1e3)
    print('threads', os.environ.get('OPAL_ORDER_THREADS'), 'n', n, 'median %.3f ms min %.3f' % (np.median(ts), min(ts)))
PY
for t in 1 2 4 8 16 32; do OPAL_ORDER_THREADS=$t python3 /tmp/ord.py; done
