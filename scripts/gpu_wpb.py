import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
n = 100000; qlen = 300
res, offs, lens = synth.database(n, seed=42)
q = synth.query(qlen, seed=7); m = blosum50()[1]
score = torch.zeros(n, dtype=torch.int32, device="cuda")
pdb = PreparedDb(res, offs); cells = float(qlen) * float(offs[-1])
for rnd in range(2):
  for wpb in os.environ.get("WPBS", "1,2,4,8").split(","):
    os.environ["OPAL_HIP_WAVES_PER_BLOCK"] = wpb
    for frac in os.environ.get("FRACS", "0.35,0.2").split(","):
        os.environ["OPAL_HIP_TAIL_FRACTION"] = frac
        ks = []
        for it in range(8):
            rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr()); ks.append(kms)
        print(f"round {rnd} wpb={wpb} frac={frac}: min {min(ks):.3f} ms median {sorted(ks)[4]:.3f} ms -> {cells/min(ks)/1e6:.0f} / {cells/sorted(ks)[4]/1e6:.0f} GCUPS", flush=True)
