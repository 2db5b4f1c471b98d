"""Where the time of one C2 search goes: per-workgroup start/end ticks of the DP launch (timeline probe of the library),
summarised as: launch span, when the workgroups of each resident round start, how long a workgroup lives, how many are
alive over time."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from opal_amd import _lib, synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
n = int(os.environ.get("NSEQ", "100000")); qlen = int(os.environ.get("QLEN", "300"))
mode = int(os.environ.get("MODE", "3")); stype = int(os.environ.get("ST", "0"))
res, offs, lens = synth.database(n, seed=42); q = synth.query(qlen, seed=7); m = blosum50()[1]
if os.environ.get("UNIFORM"):
    L0 = int(os.environ["UNIFORM"]); lens = np.full(n, L0, np.int32); offs = np.arange(n + 1, dtype=np.int64) * L0; res = res[: n * L0]
pdb = PreparedDb(res, offs); out = torch.zeros(3 * n, dtype=torch.int32, device="cuda")
L = _lib.lib()
L.opalx_db_clock_probe.argtypes = [C.c_void_p, C.c_int]
L.opalx_db_last_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; L.opalx_db_last_timeline.restype = C.c_int
L.opal_hip_wall_clock_hz.restype = C.c_double
args = (q, 11, 1, m, stype, mode, out.data_ptr(), out[n:].data_ptr() if stype else 0, out[2 * n:].data_ptr() if stype else 0)
for _ in range(20):
    rc, kms = pdb.search_device(*args)
L.opalx_db_clock_probe(pdb._h, 2)
for rep in range(2):
    rc, kms = pdb.search_device(*args)
    buf = np.zeros(2 * 65536, np.uint64)
    nb = L.opalx_db_last_timeline(pdb._h, buf.ctypes.data, len(buf))
    hz = L.opal_hip_wall_clock_hz()
    t = buf[: 2 * nb].reshape(nb, 2).astype(np.float64)
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / hz * 1e6; end = (t[:, 1] - t0) / hz * 1e6
    life = end - start
    print(f"--- search {rep}: kernel_ms (events) {kms:.4f}; {nb} workgroups, first start -> last end {end.max():.1f} us")
    print(f"workgroup life: min {life.min():.1f} median {np.median(life):.1f} max {life.max():.1f} us")
    order = np.argsort(start)
    for lo, hi in ((0, 256), (256, 1024), (1024, 2048), (2048, nb)):
        if lo >= nb: break
        sel = order[lo:min(hi, nb)]
        print(f"  workgroups {lo:5d}-{min(hi, nb):5d} by start: start {start[sel].min():8.1f} .. {start[sel].max():8.1f} us, life median {np.median(life[sel]):7.1f} us, end median {np.median(end[sel]):8.1f}")
    edges = np.linspace(0, end.max(), 21)
    alive = [(int(((start <= x) & (end > x)).sum())) for x in edges[:-1] + np.diff(edges) / 2]
    print("alive workgroups at 20 points of the span:", alive)
    byidx = [f"{np.median(life[i:i+128]):.0f}" for i in range(0, nb, max(128, nb // 24 // 128 * 128))]
    print("median life by grid index (blocks of 128):", " ".join(byidx))
