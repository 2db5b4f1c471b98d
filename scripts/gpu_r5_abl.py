"""SW + end locations: where the time goes (measurement only). Kernel ms of a few searches under the library named by
OPAL_LIB_PATH (build_abl/libopal_abl<n>.so: parts of the end-cell code switched off, results wrong on purpose)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opal_amd
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
import ctypes as C
from opal_amd import _lib
CNAMES = ["stream_sw", "stream_gl", "one_launch", "class", "chunk", "w32", "dtrace", "htrace", "reused", "swapped", "paired"]
def counters():
    buf = (C.c_longlong * 11)()
    _lib.lib().opalx_counters(buf, 11)
    return np.array([int(v) for v in buf])
dev = torch.device("cuda", 0)
m = blosum50()[1]
tag = os.environ.get("TAG", "base")
CASES = [("c2", 625000, 513, 11, {}), ("c2", 625000, 513, 11, {"OPAL_HIP_STREAM": "0"}), ("c2", 625000, 513, 3, {}), ("c2", 625000, 513, 3, {"OPAL_HIP_STREAM": "0"}),
         ("sprot", 541762, 513, 11, {}), ("sprot", 541762, 513, 11, {"OPAL_HIP_STREAM": "0"}), ("sprot", 541762, 300, 3, {}), ("sprot", 541762, 300, 3, {"OPAL_HIP_STREAM": "0"}),
         ("sprot", 541762, 513, 3, {"OPAL_HIP_STREAM": "0"}), ("sprot", 541762, 513, 3, {"OPAL_HIP_STREAM": "1"}), ("sprot", 541762, 513, 3, {}),
         ("c2", 100000, 2000, 11, {}), ("c2", 100000, 300, 11, {}), ("c2", 100000, 300, 3, {}), ("c2", 100000, 2000, 3, {})]
if os.environ.get("CASES"):
    CASES = [CASES[int(os.environ["CASES"])]]
for shape, n, Q, go, o in CASES:
    lens = synth.lengths(42, 0, n, "sprot" if shape == "sprot" else "c2")
    d_res, offs = synth.residues_device(np.arange(n, dtype=np.int64), lens, 42, dev, stride=synth.SPROT_STRIDE if shape == "sprot" else 5000)
    torch.cuda.synchronize()
    pdb = PreparedDb(None, offs, device_ptr=d_res.data_ptr())
    out = torch.zeros(3 * n, dtype=torch.int32, device=dev)
    q = synth.query(Q, seed=7)
    cells = float(offs[-1]) * Q
    for k, v in o.items(): opal_amd.set_option(k, v)
    line = f"{tag:5s} {shape:5s} n={n:6d} Q={Q:4d} gaps {go:2d}/1 {str(o):28s}"
    for st in (0, 1):
        ks = []
        c0 = counters()
        for rep in range(6):
            rc, k = pdb.search_device(q, go, 1, m, st, opal_amd.MODES["SW"], out.data_ptr(), out[n:].data_ptr() if st else 0, out[2 * n:].data_ptr() if st else 0)
            if rep: ks.append(k)
        dc = counters() - c0
        line += f" | st={st} {np.median(ks):8.3f} ms {cells / np.median(ks) / 1e6 / 8738:.3f} [" + "+".join(f"{CNAMES[i]}*{int(dc[i]) // 6}" for i in range(11) if dc[i] > 0) + "]"
    print(line, flush=True)
    for k in o: opal_amd.set_option(k, None)
    del pdb, d_res, out
    torch.cuda.empty_cache()
