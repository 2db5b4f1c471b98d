"""opalSearchDatabase wall time on the C2 workload when one call is split into K concurrent sub-searches on the same
GPU (OPAL_HIP_DEVICES=0,0,...: pack/upload of one part overlaps the search of another)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from opal_amd import synth
from opal_amd.score_matrix import blosum50
n = int(os.environ.get("NSEQ", "100000"))
res, offs, lens = synth.database(n, seed=42)
q = synth.query(300, seed=7); m = blosum50()[1]
import torch  # libopal binds to torch's HIP runtime
from opal_amd.api import PreparedDb
rc, want, _, _ = PreparedDb(res, offs).search(q, 11, 1, m, 0, 3)
for devs in ("", "0,0", "0,0,0", "0,0,0,0"):
    if devs: os.environ["OPAL_HIP_DEVICES"] = devs
    else: os.environ.pop("OPAL_HIP_DEVICES", None)
    for st in (0, 1):
        ms, ok = bench.e2e_call(q, res, offs, m, 11, 1, st, 3, want)
        print(f"devices={devs or 'current':8s} search_type={st}: {ms:.3f} ms  scores_match={ok}", flush=True)
