"""Tail-scheduling exploration on the C2 database: sweep OPAL_HIP_TAIL_FRACTION."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
n = int(os.environ.get("NSEQ", "100000")); qlen = int(os.environ.get("QLEN", "300"))
res, offs, lens = synth.database(n, seed=42)
q = synth.query(qlen, seed=7); m = blosum50()[1]
score = torch.zeros(n, dtype=torch.int32, device="cuda")
pdb = PreparedDb(res, offs)
cells = float(qlen) * float(offs[-1])
for frac in os.environ.get("FRACS", "10,0.5,0.35,0.25,0.15,0.1").split(","):
    os.environ["OPAL_HIP_TAIL_FRACTION"] = frac
    os.environ["OPAL_HIP_DEBUG"] = "1"
    ks = []
    for it in range(4):
        rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr())
        os.environ.pop("OPAL_HIP_DEBUG", None)
        ks.append(kms)
    print(f"tail_fraction={frac}: rc={rc} kernel_ms={min(ks):.3f} GCUPS={cells / min(ks) / 1e6:.1f}", flush=True)
