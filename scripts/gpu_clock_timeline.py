"""The shader clock over the life of a launch (C2's 1.5 ms launch reads 2.13 GHz on the clock probe, the same kernel
on 1 M sequences 2.38): every workgroup of the DP launch stamps the constant-rate counter AND the shader cycle counter at its
start and end (timeline probe); clock seen by a workgroup = d(cycles) / d(ticks) x tick rate. Printed: the clock by time bin
within one launch, for a lone launch (the device idle before it), for the launch at the end of a back-to-back queue of 2 / 3
searches, and for the long launch of a 1 M-sequence database."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from opal_amd import _lib, synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50

L = _lib.lib()
L.opalx_db_clock_probe.argtypes = [C.c_void_p, C.c_int]
for f in (L.opalx_db_last_timeline, L.opalx_db_last_timeline_cycles):
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; f.restype = C.c_int
L.opal_hip_wall_clock_hz.restype = C.c_double
L.opalx_db_last_clock_ghz.restype = C.c_double; L.opalx_db_last_clock_ghz.argtypes = [C.c_void_p]
hz = L.opal_hip_wall_clock_hz()
m = blosum50()[1]
q = synth.query(300, seed=7)


def timeline(pdb):
    t = np.zeros(2 * 65536, np.uint64); c = np.zeros(2 * 65536, np.uint64)
    nb = L.opalx_db_last_timeline(pdb._h, t.ctypes.data, len(t))
    nc = L.opalx_db_last_timeline_cycles(pdb._h, c.ctypes.data, len(c))
    nb = min(nb, nc)
    t = t[: 2 * nb].reshape(nb, 2).astype(np.float64); c = c[: 2 * nb].reshape(nb, 2).astype(np.float64)
    return t, c


def report(tag, pdb, kms):
    t, c = timeline(pdb)
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / hz * 1e6; end = (t[:, 1] - t0) / hz * 1e6
    ghz = (c[:, 1] - c[:, 0]) / np.maximum(t[:, 1] - t[:, 0], 1) * hz * 1e-9
    span = end.max()
    print(f"--- {tag}: kernel_ms {kms:.4f}, {len(start)} workgroups, span {span:.0f} us, probe (whole launch) {L.opalx_db_last_clock_ghz(pdb._h):.3f} GHz; "
          f"clock by workgroup: median {np.median(ghz):.3f}, p5 {np.percentile(ghz, 5):.3f}, p95 {np.percentile(ghz, 95):.3f}")
    # the clock at a time x = the cycle rate of the workgroups alive then, weighted by nothing: median over them
    edges = np.linspace(0, span, 13)
    row = []
    for a, b in zip(edges[:-1], edges[1:]):
        mid = 0.5 * (a + b)
        alive = (start <= mid) & (end > mid)
        # workgroups that live entirely inside a narrow window say most about that window
        inside = (start >= a - (b - a)) & (end <= b + (b - a))
        sel = inside if inside.sum() >= 8 else alive
        row.append(f"{mid:6.0f}us:{np.median(ghz[sel]) if sel.sum() else float('nan'):.3f}({int(alive.sum())})")
    print("    clock of the workgroups around each twelfth of the span (alive):", " ".join(row))
    # short-lived workgroups only (life < 1/6 of the span): local readings
    life = end - start
    short = life < span / 6
    if short.sum() >= 32:
        o = np.argsort(start[short])
        g = ghz[short][o]; s_ = start[short][o]
        parts = np.array_split(np.arange(len(g)), 8)
        print("    short-lived workgroups by start time:", " ".join(f"{s_[p].mean():6.0f}us:{np.median(g[p]):.3f}" for p in parts if len(p)))


def run(n, label):
    res, offs, lens = synth.database(n, seed=42)
    pdb = PreparedDb(res, offs)
    outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(4)]
    for _ in range(20):
        pdb.search_device(q, 11, 1, m, 0, 3, outs[0].data_ptr())
    L.opalx_db_clock_probe(pdb._h, 2)
    # (a) a lone launch: the device idle for 20 ms before it
    for idle_ms in (20, 2, 0):
        torch.cuda.synchronize(); time.sleep(idle_ms * 1e-3)
        rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, outs[0].data_ptr())
        report(f"{label}: one search after {idle_ms} ms of idle", pdb, kms)
    # (b) the LAST of a queue of searches submitted back to back (the timeline buffer is zeroed before each: it holds the last)
    for depth in (2, 3):
        torch.cuda.synchronize()
        for d in range(depth):
            rc = pdb.search_device_async(q, 11, 1, m, 0, 3, outs[d].data_ptr())
        kms = 0.0
        for d in range(depth):
            rc, kms = pdb.wait()
        report(f"{label}: the last of {depth} searches queued back to back", pdb, kms)
    L.opalx_db_clock_probe(pdb._h, 0)


run(100000, "C2 (100 000 sequences)")
run(int(os.environ.get("BIG", "1000000")), "1 000 000 sequences")
