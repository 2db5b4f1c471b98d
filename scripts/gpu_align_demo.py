"""OPAL_SEARCH_ALIGNMENT end to end: start cells on the device + box traceback against the host's full reverse pass
(OPAL_HIP_HOST_STARTS=1), same results, wall time of the opalSearchDatabase call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (libopal binds to torch's HIP runtime)
import oracle_lib
from opal_amd import synth
from opal_amd.score_matrix import blosum50

LIB = os.path.join(ROOT, "opal_amd", "csrc", "libopal.so")
ours = oracle_lib.OpalABI(LIB)
n = int(os.environ.get("NSEQ", "4000")); qlen = int(os.environ.get("QLEN", "300"))
res, offs, lens = synth.database(n, seed=42, first=50000)
db = [res[offs[i]:offs[i + 1]] for i in range(n)]
q = synth.query(qlen, seed=7); m = blosum50()[1]
for mode in ("SW", "OV", "HW", "NW"):
    out = {}
    for tag, env in (("device starts + box traceback", None), ("host full reverse pass", "1")):
        if env: os.environ["OPAL_HIP_HOST_STARTS"] = env
        else: os.environ.pop("OPAL_HIP_HOST_STARTS", None)
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            rc, s, eq, et, sset, (arr, _) = ours.search(q, db, 3, 1, m, 2, oracle_lib.MODES[mode])
            dt = time.perf_counter() - t0
            sq, st, ops = ours.alignments(arr)
            best = dt if best is None else min(best, dt)
        out[tag] = (best, s, sq, st, ops)
        print(f"{mode} {tag:32s}: {best * 1e3:8.1f} ms for {n} alignments (incl. the Python-side marshalling of both runs)", flush=True)
    a, b = out["device starts + box traceback"], out["host full reverse pass"]
    assert (a[1] == b[1]).all() and (a[2] == b[2]).all() and (a[3] == b[3]).all() and a[4] == b[4], mode
print("identical start cells and operation strings")
