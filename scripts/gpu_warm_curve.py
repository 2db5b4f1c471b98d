"""Kernel time of the C2 search against time since the process started using the GPU (fresh box: DPM ramp?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
res, offs, lens = synth.database(100000, seed=42); q = synth.query(300, seed=7); m = blosum50()[1]
pdb = PreparedDb(res, offs); score = torch.zeros(100000, dtype=torch.int32, device="cuda")
t0 = time.perf_counter(); ks = []
for it in range(2500):
    rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr())
    ks.append(kms)
    if (it + 1) % 250 == 0:
        print(f"t={time.perf_counter() - t0:6.2f} s  searches {it - 248:4d}-{it + 1:4d}: kernel_ms median {np.median(ks[-250:]):.4f} min {min(ks[-250:]):.4f}", flush=True)
