"""Fixed cost of one search: uniform databases of exactly one resident round (32768 sequences = 4096 waves) with
growing sequence length; kernel time = a + b * L."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
q = synth.query(300, seed=7); m = blosum50()[1]
res, offs, lens = synth.database(60000, seed=42)
n = int(os.environ.get("NSEQ", "32768"))
score = torch.zeros(n, dtype=torch.int32, device="cuda")
import time
for L in (30, 60, 120, 240, 347, 480):
    pdb = PreparedDb(res[: n * L], np.arange(n + 1, dtype=np.int64) * L)
    ks, ws = [], []
    for it in range(14):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr())
        w = (time.perf_counter() - t0) * 1e3
        if it >= 3: ks.append(kms); ws.append(w)
    print(f"L={L:4d} n={n}: kernel_ms {np.median(ks):.4f}  wall_ms {np.median(ws):.4f}  per step-column {np.median(ks) * 1e3 / (L + 7):.3f} us", flush=True)
