import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, opal_amd
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
n = synth.SPROT_SEQS; dev = torch.device("cuda", 0)
lens = synth.lengths(42, 0, n, "sprot")
d_res, offs = synth.residues_device(np.arange(n, dtype=np.int64), lens, 42, dev, stride=synth.SPROT_STRIDE)
torch.cuda.synchronize()
pdb = PreparedDb(None, offs, device_ptr=d_res.data_ptr()); m = blosum50()[1]
out = torch.zeros(3 * n, dtype=torch.int32, device=dev)
for Q in (300, 1000, 2000):
    q = synth.query(Q, seed=7)
    for mode in ("HW", "NW", "OV"):
        ks = []
        for r in range(4):
            rc, k = pdb.search_device(q, 3, 1, m, 1, opal_amd.MODES[mode], out.data_ptr(), out[n:].data_ptr(), out[2 * n:].data_ptr()); ks.append(round(k, 2))
        cells = float(offs[-1]) * Q
        print(f"sprot 541762 Q={Q} {mode}+end: kernel ms {ks} -> {cells / (min(ks[1:]) * 1e-3) / 1e9:.0f} GCUPS, frac of the 6-instr bound {cells / (min(ks[1:]) * 1e-3) / 1e9 / 13107:.3f}, crc {int(out[:n].to(torch.int64).sum().item())} rc {rc}", flush=True)
