"""C2 database without its long sequences (what the bulk class of a search sees): streaming against the class kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import opal_amd
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50

n = int(os.environ.get("NSEQ", "100000"))
res, offs, lens = synth.database(n, seed=42)
m = blosum50()[1]
q = synth.query(int(os.environ.get("QLEN", "300")), seed=7)
out = torch.zeros(n, dtype=torch.int32, device="cuda")
for cut in [int(c) for c in os.environ.get("CUTS", "700,1000,1500,6000").split(",")]:
    keep = np.nonzero(lens < cut)[0]
    pdb = PreparedDb.from_sequences([res[offs[i]:offs[i + 1]] for i in keep])
    cells = float(len(q)) * float(lens[keep].sum())
    for s in ("0", "1"):
        opal_amd.set_option("OPAL_HIP_STREAM", s)
        ks = []
        for _ in range(30):
            rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, out.data_ptr(), 0, 0)
            assert rc == 0
            ks.append(kms)
        print(f"L<{cut} ({len(keep)} seqs) stream={s}: min {min(ks):.3f} ms median {sorted(ks)[15]:.3f}  {cells / min(ks) / 1e6:.0f} GCUPS", flush=True)
