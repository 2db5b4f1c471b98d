"""A/B of the narrow rung (packed int16 vs plain 16-bit) on the C2 database and a uniform one."""
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
L = 347
dbs = {"c2": (PreparedDb(res, offs), float(qlen) * float(offs[-1])),
       "uniform": (PreparedDb(res[: n * L], np.arange(n + 1, dtype=np.int64) * L), float(qlen) * n * L)}
ref = {}
for narrow in os.environ.get("NARROWS", "16,17").split(","):
    os.environ["OPAL_HIP_NARROW"] = narrow
    for name, (pdb, cells) in dbs.items():
        for g in os.environ.get("GLIST", "auto").split(","):
            if g == "auto": os.environ.pop("OPAL_HIP_LANES_PER_SEQ", None)
            else: os.environ["OPAL_HIP_LANES_PER_SEQ"] = g
            for frac in os.environ.get("FRACS", "0.35").split(","):
                os.environ["OPAL_HIP_TAIL_FRACTION"] = frac
                ks = []
                for it in range(5):
                    rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr())
                    ks.append(kms)
                sc = score.cpu().numpy().copy()
                key = name
                if key in ref: assert (ref[key] == sc).all(), "results differ between variants!"
                else: ref[key] = sc
                print(f"narrow={narrow} {name} G={g} frac={frac}: rc={rc} kernel_ms={min(ks):.3f} GCUPS={cells / min(ks) / 1e6:.1f}", flush=True)
