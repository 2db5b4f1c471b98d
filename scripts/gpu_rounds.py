"""Wave-quantisation check: uniform-length databases whose wave count is an exact multiple of the 4096 resident waves
(1024 SIMDs x 4) against ones just above it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
q = synth.query(300, seed=7); m = blosum50()[1]
res, offs, lens = synth.database(150000, seed=42)
score = torch.zeros(150000, dtype=torch.int32, device="cuda")
L = 347
for n in (32768, 65536, 98304, 100000, 102400, 114688, 131072):
    pdb = PreparedDb(res[: n * L], np.arange(n + 1, dtype=np.int64) * L)
    ks = []
    for it in range(14):
        rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr())
        if it >= 3: ks.append(kms)
    cells = 300.0 * n * L
    print(f"uniform n={n:7d} waves={n // 8:6d} rounds={n / 8 / 4096:5.2f} kernel_ms {np.median(ks):.4f}  GCUPS {cells / np.median(ks) / 1e6:6.0f}  ms/round {np.median(ks) / (n / 8 / 4096):.4f}", flush=True)
