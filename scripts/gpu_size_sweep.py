"""Kernel GCUPS against database size and length distribution (is the C2 workload losing to start-up, tail or clock?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50

q = synth.query(300, seed=7); m = blosum50()[1]
res, offs, lens = synth.database(400000, seed=42)
score = torch.zeros(400000, dtype=torch.int32, device="cuda")


def run(tag, pdb, cells, iters=12):
    ks = []
    for it in range(iters):
        rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr())
        assert rc == 0
        if it >= 2: ks.append(kms)
    print(f"{tag:40s} kernel_ms median {np.median(ks):8.4f} min {min(ks):8.4f}  GCUPS {cells / np.median(ks) / 1e6:7.0f}", flush=True)


for n in (25000, 50000, 100000, 200000, 400000):
    run(f"c2-like n={n}", PreparedDb(res[: offs[n]], offs[: n + 1]), 300.0 * float(offs[n]))
L = 347
for n in (100000, 400000):
    run(f"uniform L={L} n={n}", PreparedDb(res[: n * L], np.arange(n + 1, dtype=np.int64) * L), 300.0 * n * L)
# the c2 database without its long tail (sequences < 1000) and only the tail
keep = np.nonzero(lens[:100000] < 1000)[0]
seqs = [res[offs[i]:offs[i + 1]] for i in keep]
run(f"c2 n=100000 only L<1000 ({len(keep)})", PreparedDb.from_sequences(seqs), 300.0 * float(lens[keep].sum()))
# back-to-back repeats of the 100k search with no host work in between other than the call itself
pdb = PreparedDb(res[: offs[100000]], offs[:100001])
import time
for rep in (1, 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(rep):
        pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / rep
    print(f"wall per call over {rep} calls: {dt * 1e3:.4f} ms", flush=True)
