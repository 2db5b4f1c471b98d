"""What do LDS bank conflicts cost? The C2 database against the same lengths with ONE letter everywhere (every lane of
a slot then reads the same letter's profile: no two lanes of a ds_read_b128 group collide on a bank)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50

n = int(os.environ.get("NSEQ", "100000"))
qlen = int(os.environ.get("QLEN", "300"))
res, offs, lens = synth.database(n, seed=42)
q = synth.query(qlen, seed=7); m = blosum50()[1]
out = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(3)]
cells = float(qlen) * float(offs[-1])

def run(pdb, tag):
    for mode, st, name in ((3, 0, "SW score"), (3, 1, "SW end"), (0, 1, "NW"), (1, 1, "HW"), (2, 1, "OV")):
        ks = []
        for it in range(int(os.environ.get('ITERS','40'))):
            rc, kms = pdb.search_device(q, 11, 1, m, st, mode, out[0].data_ptr(), out[1].data_ptr() if st else 0, out[2].data_ptr() if st else 0)
            ks.append(kms)
        print(f"{tag:12s} {name:9s}: rc={rc} kernel_ms min {min(ks):.3f} median {sorted(ks)[len(ks)//2]:.3f} GCUPS={cells / min(ks) / 1e6:.1f}", flush=True)

run(PreparedDb(res, offs), "random")
one = np.full_like(res, 11)  # K everywhere
run(PreparedDb(one, offs), "one letter")
two = res.copy(); two[:] = np.where((np.arange(len(res)) & 1) == 0, 0, 1).astype(res.dtype)  # A/R alternating: adjacent letters, 2 distinct
run(PreparedDb(two, offs), "two letters")
