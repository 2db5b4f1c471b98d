"""Is the C2 search (1.6 ms of kernels per call) running at a lower clock than long kernels? Compare the per-call kernel
time (a) as bench.py does it, (b) with a single sleeping wave kept resident on another stream for the whole series,
(c) right after a long warm-up kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50

n = 100000
res, offs, lens = synth.database(n, seed=42)
q = synth.query(300, seed=7); m = blosum50()[1]
pdb = PreparedDb(res, offs)
score = torch.zeros(n, dtype=torch.int32, device="cuda")
cells = 300.0 * float(offs[-1])


def series(tag, pre=None, iters=30):
    ks = []
    for it in range(iters):
        if pre: pre()
        rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr())
        if it >= 5: ks.append(kms)
    print(f"{tag:46s} kernel_ms median {np.median(ks):.4f} min {min(ks):.4f} GCUPS {cells / np.median(ks) / 1e6:.0f}", flush=True)


series("plain series")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    torch.cuda._sleep(int(2.0e9))  # ~1 s of one spinning thread on a side stream
series("with a spinning thread resident on a side stream")
torch.cuda.synchronize()
big = torch.randn(8192, 8192, device="cuda")
def warm():
    for _ in range(3): (big @ big)
    torch.cuda.synchronize()
series("each call right after 3 big GEMMs", pre=warm, iters=15)
def idle():
    time.sleep(0.02)
series("each call after 20 ms of idle", pre=idle, iters=15)
