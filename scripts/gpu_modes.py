"""Kernel time of every mode on the C2 database (score+end), with and without the skewed NW/HW/OV kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from opal_amd import synth
from opal_amd.api import PreparedDb, MODES
from opal_amd.score_matrix import blosum50
n = 100000; qlen = int(os.environ.get("QLEN", "300"))
res, offs, lens = synth.database(n, seed=42)
q = synth.query(qlen, seed=7); m = blosum50()[1]
s = torch.zeros(n, dtype=torch.int32, device="cuda"); a = torch.zeros_like(s); b = torch.zeros_like(s)
pdb = PreparedDb(res, offs); cells = float(qlen) * float(offs[-1])
for noskew in ("", "1"):
    if noskew: os.environ["OPAL_HIP_NO_SKEW"] = "1"
    else: os.environ.pop("OPAL_HIP_NO_SKEW", None)
    for mode in ("SW", "NW", "HW", "OV"):
        for st in (0, 1):
            ks = []
            for it in range(6):
                rc, k = pdb.search_device(q, 11, 1, m, st, MODES[mode], s.data_ptr(), a.data_ptr(), b.data_ptr()); ks.append(k)
            print(f"skew={'off' if noskew else 'on '} {mode} searchType={st}: {np.median(ks):.3f} ms -> {cells/np.median(ks)/1e6:.0f} GCUPS", flush=True)
